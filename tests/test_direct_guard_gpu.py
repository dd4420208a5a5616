"""The safety net of the library's own launch queue (fleetrl_amd/csrc/fleet_direct.hip).

Launches on that queue carry no release fence, so an env's newest state lives in ONE die's L2 between two steps: that is only correct
while workgroup w of every launch of a run runs on the die that ran workgroup w of the previous one.  The platform deals workgroups
that way but does not promise it (and the die a queue deals from moves whenever a queue is created or destroyed in the process), so
  * the queue is probed when it is opened (`fleet_direct_placement`),
  * every run records the queue's dies in its launches' argument blocks and every launch checks itself against the record
    (FLEET_DEVERR_PLACEMENT),
  * the negative tests here force a mismatch and must see the error, the positive ones must never see it;
and the launch path bench.py times is held against the CPU oracle DIRECTLY at the headline size (round 5 only compared it with the
stream launches, which are what meets the oracle).  Needs an MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_util import load_trace, params_for
from fleetrl_amd import _capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STATE = ("soc", "soh", "hours_left", "time_idx", "rf_len", "fd_cyc", "episodes", "ep_return", "last_ep_return", "last_ep_len", "error_bits")


def _batch(name="ct5_both_rainflow", E=640, seed=3, n=1):
    from fleetrl_amd.batch import FleetBatch

    g = load_trace(name)
    p = params_for(g, num_envs=E)
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, g.tables.T - g.ep_steps - 60, size=(5, E)).astype(np.int32)
    out = []
    for _ in range(n):
        b = FleetBatch(p, g.tables, g.time_feat)
        b.set_start_schedule(starts)
        out.append(b)
    return g, out, rng


def _bufs(b):
    import torch

    dev = torch.device("cuda", 0)
    o = (torch.zeros((b.E, b.obs_dim), device=dev), torch.zeros(b.E, device=dev, dtype=torch.float64), torch.zeros(b.E, device=dev, dtype=torch.uint8))
    b.reset_dev(o[0].data_ptr())
    return o


def test_placement_probe_reports_a_rotation_over_the_dies():
    g, (b,), _ = _batch(E=64)
    m, num_xcc, any_grid = b.direct_placement()
    assert num_xcc >= 1 and all(0 <= x < 8 for x in m)
    if num_xcc == 8:  # the whole part as one device: eight consecutive workgroups sit on eight different dies, in rotation
        assert sorted(m) == list(range(8))
        assert all((m[k + 1] - m[k]) % 8 == 1 for k in range(7))
    assert any_grid  # (as measured on this platform: a grid that is not a multiple of 8 workgroups does not move the next one)
    b.close()


@pytest.mark.parametrize("kind", [1, 2])
def test_placement_guard_trips_when_a_launch_lands_elsewhere(kind):
    """kind 1: the run's placement record is one workgroup off (= the queue's first die moved in the middle of a run); kind 2: one
    launch of the run has its workgroups shifted by one.  Either way the launches find themselves on another die than recorded and
    say so; a healthy run before and a fresh handle after stay clean."""
    import torch

    g, (b, c), rng = _batch(E=640, n=2)
    m, num_xcc, _ = b.direct_placement()
    if num_xcc < 2:
        pytest.skip("a single-die device has nothing to misplace")
    acts = rng.uniform(-1, 1, size=(8, 640, g.N)).astype(np.float32)
    tape = torch.from_numpy(acts).to("cuda:0")
    for x in (b, c):
        o = _bufs(x)
        x.run_tape_dev(40, tape.data_ptr(), 8, *(t.data_ptr() for t in o), use_graph=_capi.LAUNCH_DIRECT)
        x.synchronize()
        x.check_errors()  # healthy
        x._o = o
    b.debug_direct_fault(kind, 3)
    b.run_tape_dev(16, tape.data_ptr(), 8, *(t.data_ptr() for t in b._o), use_graph=_capi.LAUNCH_DIRECT)
    b.synchronize()
    bits = b.get("error_bits")
    assert (bits & _capi.DEVERR_PLACEMENT).any()
    with pytest.raises(_capi.FleetHipError, match="PLACEMENT"):
        b.check_errors()
    c.run_tape_dev(16, tape.data_ptr(), 8, *(t.data_ptr() for t in c._o), use_graph=_capi.LAUNCH_DIRECT)  # the other handle: untouched
    c.synchronize()
    c.check_errors()
    b.close(); c.close()


def test_handle_mutation_between_two_runs_on_the_same_buffers():
    """The prepared argument blocks of a run are keyed by the buffers AND the handle's generation: a start schedule set between two
    runs with the same tape / outputs must be seen by the second run (the key was pointer identity only)."""
    import torch

    g, (a, b), rng = _batch(E=300, n=2)
    acts = rng.uniform(-1, 1, size=(6, 300, g.N)).astype(np.float32)
    tape = torch.from_numpy(acts).to("cuda:0")
    oa, ob = _bufs(a), _bufs(b)
    starts2 = rng.integers(0, g.tables.T - g.ep_steps - 60, size=(3, 300)).astype(np.int32)
    for k in range(2):
        a.run_tape_dev(70, tape.data_ptr(), 6, *(t.data_ptr() for t in oa), use_graph=_capi.LAUNCH_EAGER)
        b.run_tape_dev(70, tape.data_ptr(), 6, *(t.data_ptr() for t in ob), use_graph=_capi.LAUNCH_DIRECT)
        for x in (a, b):
            x.synchronize()
        for f in STATE:
            np.testing.assert_array_equal(b.get(f), a.get(f), err_msg=f"{f} after run {k}")
        for x, o in ((a, oa), (b, ob)):  # between the runs: another schedule, a reset -- same tape, same output buffers
            x.set_start_schedule(starts2)
            x.reset_dev(o[0].data_ptr())
    np.testing.assert_array_equal(ob[0].cpu().numpy(), oa[0].cpu().numpy())
    a.close(); b.close()


def test_direct_queue_is_refused_on_a_borrowed_stream():
    import torch

    g, (b,), rng = _batch(E=64)
    o = _bufs(b)
    tape = torch.zeros((2, 64, g.N), device="cuda:0")
    b.use_torch_stream()
    with pytest.raises(_capi.FleetHipError, match="borrowed"):
        b.run_tape_dev(4, tape.data_ptr(), 2, *(t.data_ptr() for t in o), use_graph=_capi.LAUNCH_DIRECT)
    b.use_own_stream()
    b.run_tape_dev(4, tape.data_ptr(), 2, *(t.data_ptr() for t in o), use_graph=_capi.LAUNCH_DIRECT)
    b.synchronize()
    b.check_errors()
    b.close()


def test_agent_is_matched_by_pci_address_under_visible_devices():
    """HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES re-number the devices: the queue must be opened on the agent with the HIP device's PCI
    address (there is no by-ordinal fallback any more).  A child process with the variables set runs a direct run against the stream
    launches."""
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "from golden_util import load_trace, params_for\n"
        "from fleetrl_amd import _capi\n"
        "from fleetrl_amd.batch import FleetBatch\n"
        "g = load_trace('ct5_both_rainflow'); p = params_for(g, num_envs=200)\n"
        "tape = (torch.rand((5, 200, g.N), device='cuda:0') * 2 - 1)\n"
        "res = []\n"
        "for mode in (_capi.LAUNCH_EAGER, _capi.LAUNCH_DIRECT):\n"
        "    b = FleetBatch(p, g.tables, g.time_feat)\n"
        "    o = torch.zeros((200, b.obs_dim), device='cuda:0'); r = torch.zeros(200, device='cuda:0', dtype=torch.float64); d = torch.zeros(200, device='cuda:0', dtype=torch.uint8)\n"
        "    b.reset_dev(o.data_ptr()); b.run_tape_dev(60, tape.data_ptr(), 5, o.data_ptr(), r.data_ptr(), d.data_ptr(), use_graph=mode); b.synchronize(); b.check_errors()\n"
        "    res.append((o.cpu().numpy(), b.get('soc'))); b.close()\n"
        "assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])\n"
        "print('ok')\n")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0", ROCR_VISIBLE_DEVICES="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_direct_run_at_the_headline_shape_against_the_oracle():
    """4096 envs x 50 EVs, 400 launches through the library's own queue over two episode ends, from a 32-row action tape: the final
    observations / rewards / done flags and the whole state against the CPU oracle stepping the same tape -- the direct check of the
    launch path bench.py times, at full size (it was only transitively oracle-checked)."""
    import torch

    from bench import bench_config
    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.config import resolve_config
    from fleetrl_amd.params import make_params, time_features
    from fleetrl_amd.synth import synth_tables
    from oracle.fleet_oracle import OracleBatch

    E, N, L, steps = 4096, 50, 32, 400
    tb = synth_tables("ct", N)
    p = make_params(resolve_config(bench_config(E, N, "ct")), tb, E, auto_reset=True, seed=0)
    tf = time_features(tb)
    rng = np.random.default_rng(21)
    acts = rng.uniform(-1, 1, size=(L, E, N)).astype(np.float32)
    acts[rng.random(acts.shape) < 0.15] = 0.0
    b = FleetBatch(p, tb, tf)
    cpu = OracleBatch(p, tb, tf, threads=min(8, os.cpu_count() or 1))
    o = _bufs(b)
    oc = cpu.reset()
    np.testing.assert_allclose(o[0].cpu().numpy(), oc, rtol=1e-6, atol=1e-7)
    tape = torch.from_numpy(acts).to("cuda:0")
    done_steps = 0
    for chunk in (1, 63, 200, 136):  # several runs: each starts with its own placement record and ends with the release
        b.run_tape_dev(chunk, tape.data_ptr(), L, *(t.data_ptr() for t in o), use_graph=_capi.LAUNCH_DIRECT)
        # (the tape is replayed from row 0 by every run)
        for k in range(chunk):
            oc, rc, dc, _ = cpu.step(acts[k % L])
        b.synchronize()
        done_steps += chunk
        assert np.array_equal(o[2].cpu().numpy(), dc), f"done after {done_steps} launches"
        np.testing.assert_allclose(o[0].cpu().numpy(), oc, rtol=1e-5, atol=1e-6, err_msg=f"obs after {done_steps} launches")
        np.testing.assert_allclose(o[1].cpu().numpy(), rc, rtol=1e-9, atol=1e-12, err_msg=f"reward after {done_steps} launches")
    np.testing.assert_allclose(b.get("soc"), cpu.get("soc"), rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(b.get("soh"), cpu.get("soh"), rtol=1e-9)
    # The cycle bookkeeping: the reference's reversal extraction compares SOC samples EXACTLY, so an EV whose SOC saturates (this tape
    # drives batteries into their limits, like bench.py's) can count one cycle more or less in one engine than in the other once the
    # two SoH values differ in their last bits ("What that implies over very long horizons"): at most a handful
    # of the 204 800 EVs, and never anything but the cycle bookkeeping of those EVs.
    fd_h, fd_c, len_h, len_c = b.get("fd_cyc"), cpu.get("fd_cyc"), b.get("rf_len"), cpu.get("rf_len")
    off = ~np.isclose(fd_h, fd_c, rtol=1e-8, atol=1e-18) | (len_h != len_c)
    if off.any():
        import warnings

        e, c = np.argwhere(off)[0]
        warnings.warn(f"{int(off.sum())} of {off.size} EVs differ in their cycle bookkeeping; first: env {e} EV {c}: fd_cyc {fd_h[e, c]!r} / "
                      f"{fd_c[e, c]!r}, rainflow_length {len_h[e, c]} / {len_c[e, c]}, soh {b.get('soh')[e, c]!r} / {cpu.get('soh')[e, c]!r}, "
                      f"sei_l {b.get('sei_l')[e, c]!r} / {cpu.get('sei_l')[e, c]!r}")
    assert off.sum() <= 4, f"{int(off.sum())} EVs differ in their cycle bookkeeping"
    for f in ("time_idx", "episodes", "hours_left"):
        np.testing.assert_array_equal(b.get(f), cpu.get(f), err_msg=f)
    assert b.get("episodes").min() >= 2
    b.check_errors()
    b.close(); cpu.close()


def test_rainflow_cycle_and_stack_getters_count_what_a_plain_rainflow_counts():
    """FLEET_F_RF_CYCLES / FLEET_F_RF_STACK (bench.py's workload invariants: the share of EV-steps that push a reversal point / close a
    cycle) against a plain three-point rainflow over the logged SOC samples of every EV, inside one episode."""
    g, (b,), rng = _batch(E=6)
    b.set_rainflow_count_all(True)  # (by default the count stops at the episode's last degradation row: tests/test_rf_tail_gpu.py)
    obs = b.reset()
    series = [b.get("soc_deg").copy()]
    ep0 = b.get("episodes").copy()
    for k in range(150):
        b.step(rng.uniform(-1, 1, size=(6, g.N)).astype(np.float32))
        series.append(b.get("soc_deg").copy())
    assert np.array_equal(b.get("episodes"), ep0)  # still the first episode
    s = np.stack(series)  # [steps + 1, E, N]
    want_c, want_s = np.zeros((6, g.N), np.int32), np.zeros((6, g.N), np.int32)
    for e in range(6):
        for c in range(g.N):
            stack, nc, sgn, prev = [s[0, e, c]], 0, 0, s[0, e, c]
            for x in s[1:, e, c]:
                if x == prev:
                    continue
                sg = 1 if x > prev else 2
                if sgn and sg != sgn:  # `prev` was a reversal point
                    stack.append(prev)
                    while len(stack) >= 3 and not abs(stack[-1] - stack[-2]) < abs(stack[-2] - stack[-3]):
                        nc += 1
                        if len(stack) == 3:
                            stack.pop(0)
                        else:
                            last = stack.pop(); stack.pop(); stack.pop(); stack.append(last)
                sgn, prev = sg, x
            want_c[e, c], want_s[e, c] = nc, len(stack)
    np.testing.assert_array_equal(b.get("rf_cycles"), want_c)
    np.testing.assert_array_equal(b.get("rf_stack"), want_s)
    assert want_c.sum() > 20 and want_s.max() > 3
    b.close()
