"""C-ABI behaviours on the GPU beyond plain stepping: K-steps-per-launch equals K launches, tape replay through a
captured hipGraph, masked reset, external stream adoption, error reporting.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest

from golden_util import load_trace, params_for
from fleetrl_amd import _capi

pytestmark = pytest.mark.gpu


def _mk(name="ct5_both_rainflow", E=37, seed=5, **kw):
    from fleetrl_amd.batch import FleetBatch

    g = load_trace(name)
    p = params_for(g, num_envs=E, **kw)
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, g.tables.T - g.ep_steps - 60, size=(4, E)).astype(np.int32)
    b = FleetBatch(p, g.tables, g.time_feat)
    b.set_start_schedule(starts)
    return g, b, rng


def test_step_many_and_graph_tape_equal_single_steps():
    import torch

    dev = torch.device("cuda", 0)
    K = 230
    g, a, rng = _mk()
    _, b, _ = _mk()
    _, c, _ = _mk()
    acts = rng.uniform(-1, 1, size=(K, a.E, g.N)).astype(np.float32)
    a.reset(); b.reset(); c.reset()
    want_r = np.zeros(a.E); want_d = np.zeros(a.E, dtype=np.int32)
    for k in range(K):
        o, r, d, _t = a.step(acts[k])
        want_r += r; want_d += d
    tape = torch.from_numpy(acts).to(dev)
    obs = torch.empty((a.E, a.obs_dim), device=dev); rs = torch.empty(a.E, device=dev, dtype=torch.float64)
    dc = torch.empty(a.E, device=dev, dtype=torch.int32); dn = torch.empty(a.E, device=dev, dtype=torch.uint8)
    b.step_many_dev(K, tape.data_ptr(), obs.data_ptr(), rs.data_ptr(), dc.data_ptr())
    b.synchronize()
    np.testing.assert_array_equal(dc.cpu().numpy(), want_d)
    np.testing.assert_allclose(rs.cpu().numpy(), want_r, rtol=1e-12, atol=1e-9)
    np.testing.assert_array_equal(obs.cpu().numpy(), o)
    for f in ("soc", "soh", "hours_left", "time_idx", "rf_len", "fd_cyc", "episodes", "ep_return"):
        np.testing.assert_array_equal(b.get(f), a.get(f), err_msg=f)
    # single steps AFTER a K-step launch: the K-step kernel leaves the carried schedule records and the row flags of the env
    # head behind for the single-step kernel (which reads no table row for them), across schedule events and an episode end
    more = rng.uniform(-1, 1, size=(200, a.E, g.N)).astype(np.float32)
    for k in range(200):
        oa, ra, da, _ = a.step(more[k])
        ob, rb, db, _ = b.step(more[k])
        np.testing.assert_array_equal(ob, oa, err_msg=f"obs, single step {k} after the K-step launch")
        np.testing.assert_array_equal(rb, ra)
        np.testing.assert_array_equal(db, da)
    for f in ("soc", "soh", "hours_left", "time_idx", "rf_len", "fd_cyc", "episodes", "ep_return"):
        np.testing.assert_array_equal(b.get(f), a.get(f), err_msg=f)
    # the same tape, one launch per step, replayed through a captured hipGraph (tape_len | K not required)
    c.run_tape_dev(K, tape.data_ptr(), 23, obs.data_ptr(), rs.data_ptr(), dn.data_ptr(), use_graph=True)
    c.synchronize()
    _, d2, _ = _mk()
    d2.reset()
    for k in range(K):
        o2, r2, dd2, _t = d2.step(acts[k % 23])
    np.testing.assert_array_equal(obs.cpu().numpy(), o2)
    np.testing.assert_array_equal(rs.cpu().numpy(), r2)  # run_tape writes the LAST step's reward/done
    np.testing.assert_array_equal(dn.cpu().numpy(), dd2)
    for f in ("soc", "soh", "time_idx", "episodes"):
        np.testing.assert_array_equal(c.get(f), d2.get(f), err_msg=f)
    per = c.time_steps_dev(5, tape.data_ptr(), 23, obs.data_ptr(), rs.data_ptr(), dn.data_ptr())
    assert per.shape == (5,) and (per > 0).all() and (per < 5.0).all()
    for x in (a, b, c, d2):
        x.check_errors(); x.close()


def test_masked_reset_and_abandoned_episode_counting():
    g, b, rng = _mk(E=9)
    obs0 = b.reset()
    for _ in range(5):
        b.step(rng.uniform(-1, 1, size=(b.E, g.N)).astype(np.float32))
    t_before, ep_before = b.get("time_idx"), b.get("episodes")
    mask = np.zeros(b.E, dtype=np.uint8); mask[[1, 4]] = 1
    keep = np.full((b.E, b.obs_dim), 7.0, dtype=np.float32)
    out = b.reset(mask=mask, out=keep.copy())
    assert np.all(out[[0, 2, 3]] == 7.0) and not np.any(out[[1, 4]] == 7.0)   # unmasked rows untouched
    t_after, ep_after = b.get("time_idx"), b.get("episodes")
    assert np.array_equal(t_after[[0, 2, 3, 5]], t_before[[0, 2, 3, 5]])
    assert np.array_equal(ep_after[[1, 4]], ep_before[[1, 4]] + 1) and np.array_equal(ep_after[[0, 2]], ep_before[[0, 2]])
    assert np.array_equal(b.get("ep_len")[[1, 4]], [0, 0]) and b.get("ep_len")[0] == 5
    b.close()


def test_adopts_an_external_stream():
    import torch

    g, b, rng = _mk(E=5)
    s = torch.cuda.Stream()
    b.set_stream(s.cuda_stream)
    dev = torch.device("cuda", 0)
    obs = torch.empty((b.E, b.obs_dim), device=dev); rew = torch.empty(b.E, device=dev, dtype=torch.float64)
    done = torch.empty(b.E, device=dev, dtype=torch.uint8)
    with torch.cuda.stream(s):
        b.reset_dev(obs.data_ptr())
        a = torch.rand((b.E, g.N), device=dev) * 2 - 1
        b.step_dev(a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    s.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    b.close()


def test_error_reporting():
    from fleetrl_amd.batch import FleetBatch, FleetHipError

    g = load_trace("ct2_pv_nodeg")
    p = params_for(g)
    p.abi_version = 99
    with pytest.raises(FleetHipError) as ei:
        FleetBatch(p, g.tables, g.time_feat)
    assert ei.value.status == _capi.ERR_INVALID and "abi" in str(ei.value)
    p = params_for(g)
    p.deg_mode, p.init_soh = _capi.DEG_RAINFLOW, 0.95
    with pytest.raises(FleetHipError):
        FleetBatch(p, g.tables, g.time_feat)
    p = params_for(g)
    b = FleetBatch(p, g.tables, g.time_feat)
    with pytest.raises(FleetHipError):
        b.set_start_schedule(np.full((1, b.E), g.tables.T + 5, dtype=np.int32))
    with pytest.raises(ValueError):
        b.step(np.zeros((b.E + 1, g.N), dtype=np.float32))
    with pytest.raises(KeyError):
        b.get("no_such_field")
    # an episode that runs past the table raises the device error word instead of reading out of bounds -- at the step that leaves
    # the table, like the reference's table lookup (a reset near the end of the table is legal)
    b.set_start_schedule(np.full((1, b.E), g.tables.T - 3, dtype=np.int32))
    b.reset()
    b.check_errors()
    act = np.zeros((b.E, g.N), dtype=np.float32)
    b.step(act)
    b.step(act)
    with pytest.raises(IndexError):
        b.step(act)
    with pytest.raises(IndexError) as ei:   # (the reference: a KeyError of its table lookup; here the env and the row are named)
        b.check_errors()
    assert ei.value.status == _capi.ERR_STATE and ei.value.error_bits & _capi.DEVERR_TABLE_END and "table row" in str(ei.value)
    assert (b.get("error_bits") & _capi.DEVERR_TABLE_END).all()
    b.close()


def test_device_errors_are_raised_by_the_step_that_causes_them():
    """The reference raises inside step() (fleet_environment.py:610, rainflow_sei_degradation.py:164-167,179-180,209-210, a
    table lookup past the last row).  Here the kernels set per-env error bits, the host-pointer step brings their OR back in the
    block that carries rewards and dones (no extra launch), fleet_step_host returns FLEET_ERR_STATE from that very call and
    FleetBatch.step raises the reference's exception type: (a) an episode that runs off the table -> IndexError naming env and row,
    on the step that leaves the table, not earlier; (b) a SOC series whose first half cycle is deeper than 5 -> TypeError("DoD too
    large.") on the daily row that evaluates it, and the CPU oracle flags the same bit on the same step."""
    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.config import resolve_config
    from fleetrl_amd.params import make_params, time_features
    from fleetrl_amd.synth import synth_tables
    from oracle.fleet_oracle import OracleBatch
    from test_rainflow_adversarial_gpu import _always_there_tables, _cfg

    # (a) table end: one env starts 5 rows before the end of the table, the others far from it
    g = load_trace("lmd1_price_linear")
    p = params_for(g, num_envs=4)
    b = FleetBatch(p, g.tables, g.time_feat)
    starts = np.full((1, 4), 100, dtype=np.int32)
    starts[0, 2] = g.tables.T - 6
    b.set_start_schedule(starts)
    b.reset()
    a = np.zeros((4, g.N), dtype=np.float32)
    for s in range(5):       # rows T-5 .. T-1: still inside the table
        b.step(a)
        assert b.last_step_error_bits() == 0
    with pytest.raises(IndexError) as ei:   # the sixth step would read row T
        b.step(a)
    assert b.last_step_error_bits() & _capi.DEVERR_TABLE_END
    assert ei.value.env == 2 and ei.value.status == _capi.ERR_STATE and "env 2" in str(ei.value)
    assert list(np.flatnonzero(b.get("error_bits"))) == [2]
    b.close()

    # (b) DoD > 5: every EV comes back with SOC_on_return = 6 (a hand-made table), is charged "down" to the target by the
    # first step (the reference's min(need / eta, demand) with a negative need), discharged for a while and charged again:
    # reversal points [6.0, ~0.6, last] -> the first half cycle has range 5.4 and lies in the evaluated slice of the first
    # daily row (rainflow_length = 1)
    N, E = 4, 3
    tb = _always_there_tables(N)
    tb.soc_on_return[:] = 6.0
    rc = resolve_config(_cfg(48))
    p = make_params(rc, tb, E, seed=5)
    tf = time_features(tb)
    hip, cpu = FleetBatch(p, tb, tf), OracleBatch(p, tb, tf)
    daily = int(np.flatnonzero((tb.hour == 14) & (tb.minute == 45))[3])
    st = np.full((1, E), daily - 12, dtype=np.int32)
    for eng in (hip, cpu):
        eng.set_start_schedule(st)
    np.testing.assert_array_equal(hip.reset(), cpu.reset())
    raised_at = None
    for s in range(14):
        act = np.full((E, N), 1.0 if s == 0 or s >= 6 else -0.5, dtype=np.float32)
        cpu.step(act)
        try:
            hip.step(act)
            assert not cpu.get("error_bits").any(), f"the oracle flags an error at step {s}, the HIP step did not raise"
        except TypeError as exc:
            assert str(exc) == "DoD too large." and exc.error_bits & _capi.DEVERR_DOD_RANGE
            raised_at = s
            break
    assert raised_at == 11, raised_at      # the step that advances to the daily row
    assert (cpu.get("error_bits") & _capi.DEVERR_DOD_RANGE).all()
    hip.close()
    cpu.close()


def test_get_dev_matches_get():
    """fleet_get_dev unpacks a field into a device buffer; same values as the host-side fleet_get."""
    import torch

    from fleetrl_amd.batch import FleetBatch

    g = load_trace("ct5_both_rainflow")
    E = 9
    hip = FleetBatch(params_for(g, num_envs=E), g.tables, g.time_feat)
    hip.set_start_schedule(np.full((1, E), 7, dtype=np.int32))
    hip.reset()
    rng = np.random.default_rng(0)
    for _ in range(g.ep_steps + 3):
        hip.step(rng.uniform(-1, 1, size=(E, g.N)).astype(np.float32))
    dev = torch.device("cuda", 0)
    for name, dt in (("last_ep_return", torch.float64), ("last_ep_len", torch.int32), ("soc", torch.float64), ("hours_left", torch.float32)):
        shape = (E, g.N) if _capi.FIELDS[name][2] else (E,)
        buf = torch.zeros(shape, device=dev, dtype=dt)
        hip.get_dev(name, buf.data_ptr())
        hip.synchronize()
        np.testing.assert_array_equal(buf.cpu().numpy(), hip.get(name))
    assert hip.get("last_ep_len").min() == g.ep_steps


def test_c_abi_from_plain_c(tmp_path):
    """examples/capi_demo.c: the library driven from a plain C program (no Python / PyTorch in that process), against the
    same run through the Python front end."""
    import os
    import subprocess

    from fleetrl_amd import build
    from fleetrl_amd.batch import FleetBatch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "capi_demo")
    subprocess.run(["gcc", "-O1", "-DFLEET_DEMO_DEVICE_TAPE", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                    "-I", os.path.join(root, "include"), os.path.join(root, "examples", "capi_demo.c"), "-o", exe,
                    "-L", lib_dir, "-l:libfleet_hip.so", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib",
                    "-lm"], check=True)
    g = load_trace("ct5_both_rainflow")
    E, steps = 11, 230
    p = params_for(g, num_envs=E)
    p.picker_mode = _capi.PICK_RANDOM
    p.start_lo, p.start_hi = 0, g.tables.T - g.ep_steps - 60
    tb = g.tables
    (tmp_path / "params.bin").write_bytes(bytes(p))
    cols = [tb.there.astype(np.uint8), tb.time_left.astype(np.float32), tb.soc_on_return.astype(np.float64)] + \
           [np.asarray(getattr(tb, k), dtype=np.float64) for k in ("delu", "tariff", "prc", "trc", "load", "pv")] + \
           [np.asarray(getattr(tb, k), dtype=np.uint8) for k in ("hour", "minute", "month", "weekday")]
    (tmp_path / "tables.bin").write_bytes(b"".join(np.ascontiguousarray(c).tobytes() for c in cols))
    out = subprocess.run([exe, str(tmp_path / "params.bin"), str(tmp_path / "tables.bin"), str(steps)], check=True,
                         capture_output=True, text=True).stdout.split()
    c_reward, c_episodes, c_soc, c_dim = float(out[0]), int(out[1]), float(out[2]), int(out[3])
    # ... and, in that process (the system's HIP and HSA runtimes, no PyTorch), the same action tape through HIP launches and through
    # the library's own queue: the same final state
    assert out[4:] == ["direct_queue_matches", "1", "queues", "1"], out[4:]

    hip = FleetBatch(p, tb, None)  # time features computed by the library, as in the C program
    hip.reset()
    reward_sum, episodes = 0.0, 0
    idx = np.arange(E * g.N, dtype=np.int64)
    for s in range(steps):
        a = (((idx * 7 + s * 13) % 21 - 8) / 12.0).astype(np.float32).reshape(E, g.N)
        _o, r, d, _t = hip.step(a)
        for e in range(E):
            reward_sum += float(r[e])
        episodes += int(d.sum())
    assert c_dim == hip.obs_dim and c_episodes == episodes and episodes >= E
    assert c_reward == reward_sum
    soc_sum = 0.0
    for v in hip.get("soc").reshape(-1).tolist():  # same left-to-right float64 sum as the C loop
        soc_sum += v
    assert c_soc == soc_sum


def test_cycle_stress_stays_within_its_stated_accuracy():
    """the stress of a closed cycle uses a float32 hardware logarithm and polynomial exponentials inside a float64
    path.  On 2^27 samples of the reachable (depth of discharge, mean SOC, weight) domain the kernel's form stays within 1e-8 relative of
    the library's pow / exp (the docs say <= 4e-9 from the logarithm alone); SoH moves by 1e-5 per day, so that is 1e-13 of SoH."""
    import ctypes as C

    lib = _capi.load_library()
    worst = C.c_double(-1.0)
    for seed in (1, 99):
        assert lib.fleet_selftest_stress(0, 1 << 27, seed, C.byref(worst)) == _capi.OK
        assert 0.0 <= worst.value < 1e-8, worst.value


def test_rccl_gather_of_episode_stats_through_the_c_abi():
    """SURVEY.md section 8e / north star: "a single RCCL gather of episode returns for logging".  The C ABI does it without
    PyTorch's collectives: fleet_rccl_unique_id / fleet_rccl_comm_create (ncclCommInitRank) / fleet_gather_episode_stats_rccl (one
    ncclAllGather on the handle's stream).  The pool's boxes have one GPU, so the communicator has one rank here; the gathered
    block must equal the handle's own last_ep_return / last_ep_len."""
    import torch

    from fleetrl_amd.batch import FleetBatch

    g = load_trace("lmd1_price_linear")
    E = 6
    b = FleetBatch(params_for(g, num_envs=E), g.tables, g.time_feat)
    b.set_start_schedule(np.repeat(g.starts[:, :1], E, axis=1))
    b.reset()
    rng = np.random.default_rng(0)
    for _ in range(g.ep_steps + 3):        # one finished episode per env
        b.step(rng.uniform(-1, 1, size=(E, g.N)).astype(np.float32))
    lib = b.lib
    uid = (C.c_char * 128)()
    assert lib.fleet_rccl_unique_id(uid) == _capi.OK, lib.fleet_last_error(None)
    comm = C.c_void_p()
    assert lib.fleet_rccl_comm_create(0, 1, 0, uid, C.byref(comm)) == _capi.OK, lib.fleet_last_error(None)
    out = torch.zeros((1, 2, E), device="cuda:0", dtype=torch.float64)
    assert lib.fleet_gather_episode_stats_rccl(b.h, comm, 1, C.c_void_p(out.data_ptr())) == _capi.OK, lib.fleet_last_error(b.h)
    b.synchronize()
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[0, 0], b.get("last_ep_return"))
    np.testing.assert_array_equal(got[0, 1], b.get("last_ep_len").astype(np.float64))
    assert (got[0, 1] == g.ep_steps).all()
    assert lib.fleet_rccl_comm_destroy(comm) == _capi.OK
    b.close()


def test_division_by_reciprocal_is_bit_exact():
    """The charge arithmetic's two divisions (ev_charger.py:114 `need / eta_c`, :128 / :189 `energy / cap`) are formed from a
    reciprocal with a residual correction (fleet_kernels.hip div_rcp).  2^30 pseudo-random operand pairs of the charge
    arithmetic's ranges -- incl. +-0 and a 7e-18-sized residue -- against the IEEE division sequence on the device: not one
    quotient may differ in any bit."""
    lib = _capi.load_library()
    bad = (C.c_uint64 * 2)()
    assert lib.fleet_selftest_division(0, 1 << 30, 12345, bad) == 0
    assert (bad[0], bad[1]) == (0, 0), f"quotients that differ from the IEEE division: need/eta {bad[0]}, energy/cap {bad[1]}"
