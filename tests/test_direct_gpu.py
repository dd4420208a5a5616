"""Direct AQL submission (FLEET_LAUNCH_DIRECT, fleetrl_amd/csrc/fleet_direct.hip): a run of single-step launches whose kernel
boundaries carry no L2 write-back must leave exactly the state, observations and rewards the HIP-stream launches leave -- bit for
bit, over episode ends (auto-reset inside the run), daily degradation rows and every lane geometry of the single-step kernel.
The hazards this guards: a stale line in a CU's vector cache or in another die's L2 would show as a lost update.  Needs an MI355X."""
import numpy as np
import pytest

from golden_util import load_trace, params_for
from fleetrl_amd import _capi

pytestmark = pytest.mark.gpu

STATE = ("soc", "soh", "hours_left", "time_idx", "rf_len", "fd_cyc", "episodes", "ep_return", "last_ep_return", "last_ep_len",
         "error_bits")


def _pair(name, E, seed=11, **kw):
    from fleetrl_amd.batch import FleetBatch

    g = load_trace(name)
    p = params_for(g, num_envs=E, **kw)
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, g.tables.T - g.ep_steps - 60, size=(5, E)).astype(np.int32)
    out = []
    for _ in range(2):
        b = FleetBatch(p, g.tables, g.time_feat)
        b.set_start_schedule(starts)
        out.append(b)
    return g, out[0], out[1], rng


def _run_both(a, b, acts, tape_len, launches):
    import torch

    dev = torch.device("cuda", 0)
    tape = torch.from_numpy(acts[:tape_len]).to(dev)
    bufs = []
    for x in (a, b):
        bufs.append((torch.zeros((x.E, x.obs_dim), device=dev), torch.zeros(x.E, device=dev, dtype=torch.float64),
                     torch.zeros(x.E, device=dev, dtype=torch.uint8)))
        x.reset_dev(bufs[-1][0].data_ptr())
    for steps in launches:  # several runs back to back: the second starts from state the first left in the L2s and wrote back
        a.run_tape_dev(steps, tape.data_ptr(), tape_len, *(t.data_ptr() for t in bufs[0]), use_graph=_capi.LAUNCH_EAGER)
        b.run_tape_dev(steps, tape.data_ptr(), tape_len, *(t.data_ptr() for t in bufs[1]), use_graph=_capi.LAUNCH_DIRECT)
        a.synchronize()
        b.synchronize()
        for k, what in enumerate(("obs", "reward", "done")):
            np.testing.assert_array_equal(bufs[1][k].cpu().numpy(), bufs[0][k].cpu().numpy(), err_msg=f"{what} after {steps} launches")
        for f in STATE:
            np.testing.assert_array_equal(b.get(f), a.get(f), err_msg=f"{f} after {steps} launches")
    return bufs


@pytest.mark.parametrize("name,E,kw", [
    ("ct5_both_rainflow", 1500, {}),                       # 8-lane groups, rainflow: 32 envs per workgroup
    ("lmd1_price_linear", 777, {}),                        # one EV per env
    ("ut3_both_norm_rainflow", 640, {}),
])
def test_direct_run_equals_stream_launches_golden_shapes(name, E, kw):
    g, a, b, rng = _pair(name, E, **kw)
    acts = rng.uniform(-1, 1, size=(29, E, g.N)).astype(np.float32)
    _run_both(a, b, acts, 29, (1, 7, 300, 120))
    a.close(); b.close()


@pytest.mark.parametrize("E,N", [(1024, 50), (333, 50), (512, 100), (300, 200), (96, 300)])
def test_direct_run_equals_stream_launches_synthetic(E, N):
    """One wavefront per env (N = 50: env records of different workgroups share cache lines), two and four wavefronts per env, and
    several EVs per lane; two episodes long, so every env resets inside a run."""
    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.config import resolve_config
    from fleetrl_amd.params import make_params, time_features
    from test_hip_shapes import _cfg, _tables

    tb = _tables("ct", N)
    p = make_params(resolve_config(_cfg("ct", "rainflow", False, episode_length=24)), tb, E, seed=7)
    tf = time_features(tb)
    rng = np.random.default_rng(E + N)
    acts = rng.uniform(-1, 1, size=(17, E, N)).astype(np.float32)
    a, b = FleetBatch(p, tb, tf), FleetBatch(p, tb, tf)
    _run_both(a, b, acts, 17, (3, 200, 64))
    a.check_errors(); b.check_errors()
    a.close(); b.close()


def test_direct_run_then_ordinary_calls_and_back():
    """Steps, K-step launches and resets after a direct run see its results; a direct run after them sees theirs."""
    g, a, b, rng = _pair("ct5_both_rainflow", 200)
    acts = rng.uniform(-1, 1, size=(40, 200, g.N)).astype(np.float32)
    bufs = _run_both(a, b, acts, 40, (50,))
    for k in range(30):
        oa, ra, da, _ = a.step(acts[k])
        ob, rb, db, _ = b.step(acts[k])
        np.testing.assert_array_equal(ob, oa)
        np.testing.assert_array_equal(rb, ra)
    mask = (np.arange(200) % 3 == 0).astype(np.uint8)
    np.testing.assert_array_equal(b.reset(mask), a.reset(mask))
    import torch

    tape = torch.from_numpy(acts).to("cuda:0")
    a.run_tape_dev(90, tape.data_ptr(), 40, *(t.data_ptr() for t in bufs[0]), use_graph=_capi.LAUNCH_GRAPH)
    b.run_tape_dev(90, tape.data_ptr(), 40, *(t.data_ptr() for t in bufs[1]), use_graph=_capi.LAUNCH_DIRECT)
    # no synchronize: the get() itself must wait for the run
    for f in STATE:
        np.testing.assert_array_equal(b.get(f), a.get(f), err_msg=f)
    np.testing.assert_array_equal(bufs[1][0].cpu().numpy(), bufs[0][0].cpu().numpy())
    a.close(); b.close()


def test_direct_is_refused_where_it_does_not_apply():
    from fleetrl_amd.batch import FleetBatch
    import torch

    g = load_trace("ct5_both_rainflow")
    p = params_for(g, num_envs=8)
    p.log_data, p.log_capacity = 1, 16
    b = FleetBatch(p, g.tables, g.time_feat)
    tape = torch.zeros((4, 8, g.N), device="cuda:0")
    obs = torch.zeros((8, b.obs_dim), device="cuda:0"); r = torch.zeros(8, device="cuda:0", dtype=torch.float64)
    d = torch.zeros(8, device="cuda:0", dtype=torch.uint8)
    b.reset_dev(obs.data_ptr())
    with pytest.raises(Exception, match="single-step"):
        b.run_tape_dev(4, tape.data_ptr(), 4, obs.data_ptr(), r.data_ptr(), d.data_ptr(), use_graph=_capi.LAUNCH_DIRECT)
    b.close()


def test_direct_float64_tape_and_two_handles_in_flight():
    """The float64-action instance through the queue, and two handles (two queues) with runs in flight at the same time."""
    import torch

    g, a, b, rng = _pair("ct5_both_rainflow", 400)
    _, c, d, _ = _pair("lmd1_price_linear", 300, seed=12)
    acts = rng.uniform(-1, 1, size=(13, 400, g.N))
    acts2 = rng.uniform(-1, 1, size=(13, 300, c.N)).astype(np.float32)
    dev = torch.device("cuda", 0)
    t64, t32 = torch.from_numpy(acts).to(dev), torch.from_numpy(acts2).to(dev)

    def bufs(x):
        o = (torch.zeros((x.E, x.obs_dim), device=dev), torch.zeros(x.E, device=dev, dtype=torch.float64), torch.zeros(x.E, device=dev, dtype=torch.uint8))
        x.reset_dev(o[0].data_ptr())
        return o

    ba, bb, bc, bd = bufs(a), bufs(b), bufs(c), bufs(d)
    a.run_tape_dev(150, t64.data_ptr(), 13, *(t.data_ptr() for t in ba), use_graph=_capi.LAUNCH_EAGER, act_dtype=_capi.ACT_F64)
    c.run_tape_dev(220, t32.data_ptr(), 13, *(t.data_ptr() for t in bc), use_graph=_capi.LAUNCH_EAGER)
    b.run_tape_dev(150, t64.data_ptr(), 13, *(t.data_ptr() for t in bb), use_graph=_capi.LAUNCH_DIRECT, act_dtype=_capi.ACT_F64)
    d.run_tape_dev(220, t32.data_ptr(), 13, *(t.data_ptr() for t in bd), use_graph=_capi.LAUNCH_DIRECT)  # while b's run is in flight
    for x in (a, b, c, d):
        x.synchronize()
    for want, got, x, y in ((ba, bb, a, b), (bc, bd, c, d)):
        for k in range(3):
            np.testing.assert_array_equal(got[k].cpu().numpy(), want[k].cpu().numpy())
        for f in STATE:
            np.testing.assert_array_equal(y.get(f), x.get(f), err_msg=f)
    for x in (a, b, c, d):
        x.close()


def test_large_batch_runs_on_two_queues_bit_identically():
    """6400 envs x 50 EVs = 6400 wavefronts per launch: FLEET_LAUNCH_DIRECT covers the grid with two ranges of workgroups on two queues
    (the second range's workgroups continue the first one's numbering); the result is that of the stream launches and of one queue."""
    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.config import resolve_config
    from fleetrl_amd.params import make_params, time_features
    from test_hip_shapes import _cfg, _tables
    import torch

    E, N = 6400, 50
    tb = _tables("ct", N)
    p = make_params(resolve_config(_cfg("ct", "rainflow", False, episode_length=24)), tb, E, seed=3)
    tf = time_features(tb)
    rng = np.random.default_rng(1)
    acts = rng.uniform(-1, 1, size=(9, E, N)).astype(np.float32)
    dev = torch.device("cuda", 0)
    tape = torch.from_numpy(acts).to(dev)
    runs = []
    for mode in (_capi.LAUNCH_EAGER, _capi.LAUNCH_DIRECT, _capi.LAUNCH_DIRECT_ONE_QUEUE):
        b = FleetBatch(p, tb, tf)
        o = (torch.zeros((E, b.obs_dim), device=dev), torch.zeros(E, device=dev, dtype=torch.float64), torch.zeros(E, device=dev, dtype=torch.uint8))
        b.reset_dev(o[0].data_ptr())
        for steps in (5, 130):
            b.run_tape_dev(steps, tape.data_ptr(), 9, *(t.data_ptr() for t in o), use_graph=mode)
        b.synchronize()
        runs.append((b, o))
    assert runs[1][0].direct_queues() == 2 and runs[2][0].direct_queues() == 1
    for b, o in runs[1:]:
        for k in range(3):
            np.testing.assert_array_equal(o[k].cpu().numpy(), runs[0][1][k].cpu().numpy())
        for f in STATE:
            np.testing.assert_array_equal(b.get(f), runs[0][0].get(f), err_msg=f)
        b.check_errors()
    for b, _ in runs:
        b.close()


@pytest.mark.parametrize("E,steps,queues", [(4096, 2500, 1), (16384, 700, 2), (4096, 100000, 1)])
def test_direct_long_run_at_the_bench_shapes(E, steps, queues):
    """The headline batch (4096 x 50, one queue) over 13 episodes and the 16384 x 50 batch (two queues) over 3: the state the library's
    own launches leave is the stream launches' state, word for word -- every env, every EV, the rainflow counts and the SoH included.
    The third case is a soak: 100 000 launches in ONE run (520 episodes per env, every launch checked by the placement guard)."""
    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.config import resolve_config
    from fleetrl_amd.params import make_params, time_features
    from fleetrl_amd.synth import synth_tables
    from bench import bench_config
    import torch

    N = 50
    tb = synth_tables("ct", N)
    p = make_params(resolve_config(bench_config(E, N, "ct")), tb, E, seed=0)
    tf = time_features(tb)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    tape = torch.rand((8, E, N), device=dev, generator=gen) * 2 - 1
    tape[torch.rand((8, E, N), device=dev, generator=gen) < 0.15] = 0.0
    out = []
    for mode in (_capi.LAUNCH_GRAPH, _capi.LAUNCH_DIRECT):
        b = FleetBatch(p, tb, tf)
        o = (torch.zeros((E, b.obs_dim), device=dev), torch.zeros(E, device=dev, dtype=torch.float64), torch.zeros(E, device=dev, dtype=torch.uint8))
        b.reset_dev(o[0].data_ptr())
        b.run_tape_dev(steps, tape.data_ptr(), 8, *(t.data_ptr() for t in o), use_graph=mode)
        b.synchronize()
        out.append((b, o))
    assert out[1][0].direct_queues() == queues
    for k in range(3):
        np.testing.assert_array_equal(out[1][1][k].cpu().numpy(), out[0][1][k].cpu().numpy())
    for f in STATE + ("sei_l", "cashflow"):
        np.testing.assert_array_equal(out[1][0].get(f), out[0][0].get(f), err_msg=f)
    assert out[0][0].get("episodes").min() >= steps // 192
    for b, _ in out:
        b.check_errors()
        b.close()
