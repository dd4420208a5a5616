"""The safety net of the library's own launch queue (fleetrl_amd/csrc/fleet_direct.hip) and the closed-loop step on it.

Launches on that queue carry no release fence, so an env's newest state lives in ONE die's L2 between two steps: that is only correct
while workgroup w of every launch of a chain runs on the die that ran workgroup w of the previous one.  The platform deals workgroups
that way but does not promise it (and the die a queue deals from moves when queues are created), so
  * the queue is probed when it is opened (`fleet_direct_placement`),
  * every chain records the queue's dies on the device and every launch checks itself against the record (FLEET_DEVERR_PLACEMENT),
  * the negative tests here force a mismatch and must see the error, the positive ones must never see it;
and the closed-loop entry (`fleet_step_direct_dev` / `fleet_wait_step`: outputs written through, state left in the L2s) must give a
policy that reads EVERY step's observation exactly what the stream launches give it -- and what the CPU oracle computes.
Reference contract: FleetEnv.step returns the observation of every step (fleet_environment.py:436,702).  Needs an MI355X."""
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
    """kind 1: the chain's placement record is one workgroup off (= the queue's first die moved in the middle of a chain); kind 2: one
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


# ---- the closed loop ----------------------------------------------------------------------------------------------------------
def _policy(obs, N):
    """A policy that needs EVERY step's observation: charge against the state of charge and the time left it has just been shown
    (float32 arithmetic, the same on the device and in NumPy)."""
    soc, hl = obs[:, :N], obs[:, N:2 * N]
    a = 0.9 - 1.7 * soc + 0.02 * hl
    return a.clip(-1.0, 1.0) if isinstance(a, np.ndarray) else a.clamp(-1.0, 1.0)


@pytest.mark.parametrize("name,E,steps", [("ct5_both_rainflow", 777, 420), ("lmd1_price_linear", 300, 250), ("ut3_both_norm_rainflow", 640, 300)])
def test_closed_loop_steps_equal_stream_steps_golden_shapes(name, E, steps):
    """A torch policy on torch's stream reads every step's observation and writes the next action; the steps go through the library's
    own queue (no release fence, outputs written through) on one handle and through fleet_step_dev on the other: bit-identical
    observations, rewards and done flags at EVERY step, and the same state afterwards."""
    import torch

    g, (a, b), rng = _batch(name, E, n=2)
    dev = torch.device("cuda", 0)
    oa, ob = _bufs(a), _bufs(b)
    act_a = torch.zeros((E, g.N), device=dev)
    act_b = torch.zeros((E, g.N), device=dev)
    term = torch.zeros((E, a.obs_dim), device=dev)
    term_b = torch.zeros((E, a.obs_dim), device=dev)
    for k in range(steps):
        act_a.copy_(_policy(oa[0], g.N))
        act_b.copy_(_policy(ob[0], g.N))
        torch.cuda.synchronize()
        a.step_dev(act_a.data_ptr(), *(t.data_ptr() for t in oa), terminal_ptr=term.data_ptr())
        a.synchronize()
        b.step_direct_dev(act_b.data_ptr(), *(t.data_ptr() for t in ob), terminal_ptr=term_b.data_ptr())
        b.wait_step()  # outputs visible; the state is NOT written back
        for x, y, what in zip(ob, oa, ("obs", "reward", "done")):
            assert torch.equal(x, y), f"{what} at step {k}"
        d = oa[2].bool()
        assert torch.equal(term_b[d], term[d]), f"terminal observations at step {k}"
    for f in STATE:
        np.testing.assert_array_equal(b.get(f), a.get(f), err_msg=f)  # (the get writes the state back first)
    b.check_errors()
    a.close(); b.close()


def test_closed_loop_at_the_headline_shape_against_the_oracle():
    """4096 envs x 50 EVs, 400 closed-loop steps over two episode ends: every step's observation feeds the next action.  The library's
    own queue against the stream launches bit for bit at every step, and against the CPU oracle (driven with the same policy on ITS
    observations) within the parity tolerances -- the direct check of this launch path at full size."""
    import torch

    from bench import bench_config
    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.config import resolve_config
    from fleetrl_amd.params import make_params, time_features
    from fleetrl_amd.synth import synth_tables
    from oracle.fleet_oracle import OracleBatch

    E, N, steps = 4096, 50, 400
    tb = synth_tables("ct", N)
    p = make_params(resolve_config(bench_config(E, N, "ct")), tb, E, auto_reset=True, seed=0)
    tf = time_features(tb)
    dev = torch.device("cuda", 0)
    a, b = FleetBatch(p, tb, tf), FleetBatch(p, tb, tf)
    cpu = OracleBatch(p, tb, tf, threads=min(8, os.cpu_count() or 1))
    oa, ob = _bufs(a), _bufs(b)
    oc = cpu.reset()
    np.testing.assert_allclose(ob[0].cpu().numpy(), oc, rtol=1e-6, atol=1e-7)
    act_a, act_b = torch.zeros((E, N), device=dev), torch.zeros((E, N), device=dev)
    worst = 0.0
    for k in range(steps):
        act_a.copy_(_policy(oa[0], N))
        act_b.copy_(_policy(ob[0], N))
        torch.cuda.synchronize()
        a.step_dev(act_a.data_ptr(), *(t.data_ptr() for t in oa))
        a.synchronize()
        b.step_direct_dev(act_b.data_ptr(), *(t.data_ptr() for t in ob))
        b.wait_step()
        for x, y, what in zip(ob, oa, ("obs", "reward", "done")):
            assert torch.equal(x, y), f"{what} at step {k}"
        oc, rc, dc, _ = cpu.step(_policy(oc, N).astype(np.float32))
        if k % 8 == 0 or k > steps - 4:
            oh = ob[0].cpu().numpy()
            assert np.array_equal(ob[2].cpu().numpy(), dc), f"done at step {k}"
            np.testing.assert_allclose(oh, oc, rtol=1e-5, atol=1e-6, err_msg=f"obs vs the oracle at step {k}")
            np.testing.assert_allclose(ob[1].cpu().numpy(), rc, rtol=1e-7, atol=1e-9, err_msg=f"reward vs the oracle at step {k}")
            worst = max(worst, float(np.max(np.abs(oh - oc) / (np.abs(oc) + 1e-6))))
    for f in STATE:
        np.testing.assert_array_equal(b.get(f), a.get(f), err_msg=f)
    np.testing.assert_allclose(b.get("soc"), cpu.get("soc"), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b.get("soh"), cpu.get("soh"), rtol=1e-9)
    assert np.array_equal(b.get("time_idx"), cpu.get("time_idx")) and np.array_equal(b.get("rf_len"), cpu.get("rf_len"))
    assert b.get("episodes").min() >= 2
    b.check_errors()
    a.close(); b.close(); cpu.close()


@pytest.mark.parametrize("E,N", [(333, 50), (512, 100), (300, 200), (96, 300), (1500, 5)])
def test_publishing_launches_equal_stream_launches_every_geometry(E, N):
    """FLEET_LAUNCH_DIRECT_PUBLISH (the closed-loop launch fed from a tape) for one wavefront per env with a partly filled last
    workgroup, two and four wavefronts per env, several EVs per lane and 8-lane groups: the write-through stores of every geometry."""
    import torch

    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.config import resolve_config
    from fleetrl_amd.params import make_params, time_features
    from test_hip_shapes import _cfg, _tables

    tb = _tables("ct", N)
    p = make_params(resolve_config(_cfg("ct", "rainflow", False, episode_length=24)), tb, E, seed=7)
    tf = time_features(tb)
    rng = np.random.default_rng(E + N)
    acts = rng.uniform(-1, 1, size=(11, E, N)).astype(np.float32)
    tape = torch.from_numpy(acts).to("cuda:0")
    a, b = FleetBatch(p, tb, tf), FleetBatch(p, tb, tf)
    oa, ob = _bufs(a), _bufs(b)
    for steps in (3, 150, 64):
        a.run_tape_dev(steps, tape.data_ptr(), 11, *(t.data_ptr() for t in oa), use_graph=_capi.LAUNCH_EAGER)
        b.run_tape_dev(steps, tape.data_ptr(), 11, *(t.data_ptr() for t in ob), use_graph=_capi.LAUNCH_DIRECT_PUBLISH)
        a.synchronize(); b.synchronize()
        for x, y, what in zip(ob, oa, ("obs", "reward", "done")):
            assert torch.equal(x, y), f"{what} after {steps} launches"
        for f in STATE:
            np.testing.assert_array_equal(b.get(f), a.get(f), err_msg=f)
    b.check_errors()
    a.close(); b.close()


def test_closed_loop_steps_mix_with_every_other_entry():
    """Steps through the queue, then -- without any explicit synchronisation -- host steps, a masked reset, a K-step launch, a tape run:
    every entry point writes the state back first."""
    import torch

    g, (a, b), rng = _batch(E=200, n=2)
    dev = torch.device("cuda", 0)
    oa, ob = _bufs(a), _bufs(b)
    acts = rng.uniform(-1, 1, size=(40, 200, g.N)).astype(np.float32)
    tape = torch.from_numpy(acts).to(dev)
    for k in range(25):
        a.step_dev(tape[k].data_ptr(), *(t.data_ptr() for t in oa))
        b.step_direct_dev(tape[k].data_ptr(), *(t.data_ptr() for t in ob))  # another action pointer every step: re-prepared each time
    for k in range(10):  # host steps straight after
        xa, ra, da, _ = a.step(acts[k])
        xb, rb, db, _ = b.step(acts[k])
        np.testing.assert_array_equal(xb, xa); np.testing.assert_array_equal(rb, ra)
    mask = (np.arange(200) % 3 == 0).astype(np.uint8)
    for k in range(5):
        b.step_direct_dev(tape[k].data_ptr(), *(t.data_ptr() for t in ob))
        a.step_dev(tape[k].data_ptr(), *(t.data_ptr() for t in oa))
    np.testing.assert_array_equal(b.reset(mask), a.reset(mask))
    rs_a, rs_b = torch.zeros(200, device=dev, dtype=torch.float64), torch.zeros(200, device=dev, dtype=torch.float64)
    b.step_direct_dev(tape[7].data_ptr(), *(t.data_ptr() for t in ob))
    a.step_dev(tape[7].data_ptr(), *(t.data_ptr() for t in oa))
    a.step_many_dev(16, tape.data_ptr(), oa[0].data_ptr(), rs_a.data_ptr())
    b.step_many_dev(16, tape.data_ptr(), ob[0].data_ptr(), rs_b.data_ptr())
    b.step_direct_dev(tape[9].data_ptr(), *(t.data_ptr() for t in ob))
    a.step_dev(tape[9].data_ptr(), *(t.data_ptr() for t in oa))
    a.run_tape_dev(30, tape.data_ptr(), 40, *(t.data_ptr() for t in oa), use_graph=_capi.LAUNCH_GRAPH)
    b.run_tape_dev(30, tape.data_ptr(), 40, *(t.data_ptr() for t in ob), use_graph=_capi.LAUNCH_DIRECT)
    for f in STATE:
        np.testing.assert_array_equal(b.get(f), a.get(f), err_msg=f)
    assert torch.equal(ob[0], oa[0]) and torch.equal(rs_b, rs_a)
    b.check_errors()
    a.close(); b.close()
