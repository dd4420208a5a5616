"""The drop-in Python surface on the GPU: return types/shapes/dtypes of the reference's FleetEnv, the SB3 VecEnv
contract (auto-reset, terminal_observation, env_method) and the gymnasium vector signature.  Needs an MI355X."""
import numpy as np
import pytest

from golden_util import load_trace, params_for

pytestmark = pytest.mark.gpu


def test_single_env_signature_and_values_match_the_reference():
    """FleetEnv.reset()/step() return exactly what the reference returns (fleet_environment.py:434, :702)."""
    from fleetrl_amd import FleetEnv

    g = load_trace("lmd1_price_linear")
    env = FleetEnv(g.cfg, tables=g.tables, start_rows=g.starts[:, :1], extrema=g.extrema, start_range=(0, 0))
    assert env.observation_space.shape == (int(g.sc_obs_dim),) and env.observation_space.dtype == np.float32
    assert env.action_space.shape == (g.N,) and float(env.action_space.low.min()) == -1 and float(env.action_space.high.max()) == 1
    assert np.isinf(env.observation_space.low).all()
    obs, info = env.reset()
    assert isinstance(obs, np.ndarray) and obs.dtype == np.float32 and obs.shape == (int(g.sc_obs_dim),) and info == {}
    np.testing.assert_array_equal(obs, g.reset_obs[0, 0])
    for k in range(g.ep_steps):
        out = env.step(g.actions[0, k])
        assert len(out) == 5
        o, r, d, tr, inf = out
        assert isinstance(r, float) and isinstance(d, bool) and tr is False and inf == {}
        assert o.dtype == np.float32
        np.testing.assert_allclose(o, g.obs[0, k], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r, g.reward[0, k], rtol=1e-9, atol=1e-12)
        assert d == bool(g.done[0, k])
        assert env.is_done() == d
    assert d is True
    # gymnasium.Env semantics: no auto-reset, the last observation is the terminal one
    np.testing.assert_allclose(o, g.terminal_obs[0, 0], rtol=1e-5, atol=1e-6)
    obs, _ = env.reset()
    np.testing.assert_array_equal(obs, g.reset_obs[0, 1])
    np.testing.assert_allclose(env.get_dist_factor(), g.dist_factor[0, 1], rtol=1e-12)
    assert str(env.get_time()) == str(env.get_start_time())
    env.close()


def test_sb3_vec_env_contract_against_a_sequential_oracle_loop():
    from fleetrl_amd import FleetVecEnv
    from oracle.fleet_oracle import OracleBatch

    g = load_trace("ct5_both_rainflow")
    E = 6
    rng = np.random.default_rng(3)
    starts = rng.integers(0, g.tables.T - g.ep_steps - 60, size=(2, E)).astype(np.int32)
    venv = FleetVecEnv(g.cfg, E, tables=g.tables, start_rows=starts, extrema=g.extrema, start_range=(0, 0))
    ora = OracleBatch(venv.core.params, g.tables, g.time_feat)
    ora.set_start_schedule(starts)
    assert venv.num_envs == E and venv.env_is_wrapped(object) == [False] * E
    obs = venv.reset()
    np.testing.assert_array_equal(obs, ora.reset())
    n_done = 0
    for s in range(g.ep_steps + 5):
        a = rng.uniform(-0.3, 1, size=(E, g.N)).astype(np.float32)
        venv.step_async(a)
        obs, rew, dones, infos = venv.step_wait()
        o2, r2, d2, t2 = ora.step(a)
        assert obs.dtype == np.float32 and rew.dtype == np.float32 and dones.dtype == bool and len(infos) == E
        np.testing.assert_array_equal(dones, d2.astype(bool))
        np.testing.assert_allclose(obs, o2, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rew, r2.astype(np.float32), rtol=1e-6, atol=1e-6)
        for i in range(E):
            if dones[i]:
                n_done += 1
                np.testing.assert_allclose(infos[i]["terminal_observation"], t2[i], rtol=1e-5, atol=1e-6)
                assert infos[i]["episode"]["l"] == g.ep_steps and infos[i]["TimeLimit.truncated"] is False
            else:
                assert infos[i] == {}
    assert n_done == E
    # env_method fan-out (fleet_environment.py:741-799)
    assert venv.env_method("is_done") == [False] * E
    t = venv.env_method("get_time")
    assert len(t) == E and str(t[0]) == str(g.tables.dates[ora.get("time_idx")[0]]).replace("T", " ")
    np.testing.assert_allclose(np.array(venv.env_method("get_dist_factor", indices=[1, 3])), ora.dist_factor()[[1, 3]], rtol=1e-12)
    venv.env_method("set_start_time", "2020-03-01 00:00", indices=0)
    assert venv.env_method("get_start_time", indices=0) == ["2020-03-01 00:00"]
    assert venv.env_method("get_log")[0] is None
    with pytest.raises(AttributeError):
        venv.env_method("no_such_method")
    venv.close()


def test_gymnasium_vector_signature_and_torch_path():
    import torch

    from fleetrl_amd import FleetVecEnv, FleetVectorEnv

    g = load_trace("lmd5_price_linear")
    E = 9
    starts = np.full((1, E), 10, dtype=np.int32)
    v1 = FleetVectorEnv(g.cfg, E, tables=g.tables, start_rows=starts, extrema=g.extrema, start_range=(0, 0))
    v2 = FleetVecEnv(g.cfg, E, tables=g.tables, start_rows=starts, extrema=g.extrema, start_range=(0, 0))
    o1, info = v1.reset(seed=123)
    o2 = v2.reset()
    assert info == {} and np.array_equal(o1, o2)
    assert v1.observation_space.shape == (E, v1.core.obs_dim) and v1.action_space.shape == (E, g.N)
    rng = np.random.default_rng(0)
    dev = torch.device("cuda", 0)
    for s in range(g.ep_steps):
        a = rng.uniform(-1, 1, size=(E, g.N)).astype(np.float32)
        obs, rew, term, trunc, infos = v1.step(a)
        ot, rt, dt = v2.step_torch(torch.from_numpy(a).to(dev))
        v2.core.batch.synchronize()
        assert term.dtype == bool and not trunc.any() and rew.dtype == np.float64
        np.testing.assert_array_equal(obs, ot.cpu().numpy())
        np.testing.assert_array_equal(rew, rt.cpu().numpy())
        np.testing.assert_array_equal(term, dt.cpu().numpy().astype(bool))
    assert term.all() and infos["_final_observation"].all() and infos["final_observation"][0].shape == (v1.core.obs_dim,)
    v1.close()
    v2.close()


def test_data_log_matches_the_reference_logger():
    """`get_log()` after two episodes against the DataLogger DataFrame of the reference (log_data=True), column by
    column, including the non-obvious ones: Penalties = reward - cashflow*price_multiplier (quirk Q13), Charging energy
    with its cross-car carry-over (quirk Q8), Degradation only on the 14:45 row, no row for the final step of an episode."""
    from fleetrl_amd import FleetVecEnv

    g = load_trace("custom3_both_overload_log")
    venv = FleetVecEnv(g.cfg, g.E, tables=g.tables, start_rows=g.starts, extrema=g.extrema, start_range=(0, 0))
    venv.reset()
    for k in range(g.total):
        venv.step(g.actions[:, k])
    # the reference never logs the reset that follows the last episode; our vec env has auto-reset into a third one
    _assert_log_equals_reference(g, venv.env_method("get_log"), extra_rows=1)
    assert (g.log_grid > 0).any() and (g.log_socv > 0).any() and (np.abs(g.log_deg) > 0).any()
    venv.close()


def _assert_log_equals_reference(g, logs, extra_rows):
    for e in range(g.E):
        lg = logs[e].reset_index(drop=True)
        rows = g.log_reward.shape[1]
        assert len(lg) == rows + extra_rows
        lg = lg.iloc[:rows]
        np.testing.assert_array_equal(lg["Episode"].values.astype(int), g.log_episode[e])
        np.testing.assert_array_equal(lg["Time"].values.astype("datetime64[s]").astype(np.int64), g.log_time[e])
        np.testing.assert_allclose(lg["Reward"].values.astype(float), g.log_reward[e], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(lg["Cashflow"].values.astype(float), g.log_cashflow[e], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(lg["Penalties"].values.astype(float), g.log_penalty[e], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(lg["Grid overloading"].values.astype(float), g.log_grid[e], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(lg["SOC violation"].values.astype(float), g.log_socv[e], rtol=1e-9, atol=1e-12)
        for k in range(rows):
            np.testing.assert_allclose(np.broadcast_to(lg["Degradation"].iloc[k], (g.N,)), g.log_deg[e, k], rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(lg["Charging energy"].iloc[k], g.log_charge[e, k], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(lg["SOH"].iloc[k], g.log_soh[e, k], rtol=1e-9)
            np.testing.assert_allclose(lg["Observation"].iloc[k], g.log_obs[e, k], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(np.asarray(lg["Action"].iloc[k], dtype=np.float64), g.log_action[e, k])


def test_data_log_is_written_by_k_step_launches_too():
    """The reference's harnesses read the log after their policy loop (benchmarking/uncontrolled_charging.py:56,
    night_charging.py:100).  The log is a device-side ring written by the kernels, so open-loop K-step launches
    (`fleet_step_many_dev`) leave exactly the rows single steps would: the golden trace's actions replayed as a tape in
    chunks of 37 steps reproduce the reference's DataLogger frame."""
    import torch

    from fleetrl_amd import FleetVecEnv

    g = load_trace("custom3_both_overload_log")
    venv = FleetVecEnv(g.cfg, g.E, tables=g.tables, start_rows=g.starts, extrema=g.extrema, start_range=(0, 0))
    venv.reset()
    b = venv.core.batch
    dev = torch.device("cuda", 0)
    tape = torch.as_tensor(np.ascontiguousarray(g.actions.transpose(1, 0, 2)), device=dev)  # [steps, E, N] float32
    obs = torch.zeros((g.E, b.obs_dim), device=dev)
    rsum = torch.zeros(g.E, device=dev, dtype=torch.float64)
    dcnt = torch.zeros(g.E, device=dev, dtype=torch.int32)
    torch.cuda.synchronize()
    k = 0
    while k < g.total:
        n = min(37, g.total - k)
        b.step_many_dev(n, tape[k:].data_ptr(), obs.data_ptr(), rsum.data_ptr(), dcnt.data_ptr())
        k += n
    b.synchronize()
    b.check_errors()
    _assert_log_equals_reference(g, venv.env_method("get_log"), extra_rows=1)
    # the ring keeps the newest rows: a capacity of 50 rows per env holds the tail of the same log
    cfg = dict(g.cfg)
    cfg["log_capacity"] = 50
    small = FleetVecEnv(cfg, g.E, tables=g.tables, start_rows=g.starts, extrema=g.extrema, start_range=(0, 0))
    small.reset()
    for j in range(g.total):
        small.step(g.actions[:, j])
    full = venv.env_method("get_log")
    assert venv.core.batch.log_dropped() == 0
    # a ring that is too small loses rows the reference's unbounded DataLogger would keep: counted, and get_log() says so
    assert small.core.batch.log_dropped() == sum(max(len(f) - 50, 0) for f in full) > 0
    with pytest.warns(RuntimeWarning, match="overwritten"):
        tail = small.env_method("get_log")
    for e in range(g.E):
        assert len(tail[e]) == 50
        ref = full[e].iloc[-50:].reset_index(drop=True)
        for col in ("Episode", "Reward", "Cashflow", "Penalties", "Grid overloading", "SOC violation"):
            np.testing.assert_array_equal(tail[e][col].values.astype(float), ref[col].values.astype(float), err_msg=col)
        assert (tail[e]["Time"].values == ref["Time"].values).all()
        for col in ("Charging energy", "SOH", "Observation", "Action"):
            for a, r in zip(tail[e][col], ref[col]):
                np.testing.assert_array_equal(a, r)
    small.core.clear_log()
    assert all(len(x) == 0 for x in small.env_method("get_log"))
    venv.close()
    small.close()


def test_data_log_after_a_device_side_policy_rollout():
    """`fleet_rollout_policy_dev` (uncontrolled charging evaluated on the device, 150 steps in two launches) logs what the
    same policy stepped from the host logs -- every column of the reference's frame, incl. the action the rule chose."""
    from fleetrl_amd import FleetVecEnv
    from fleetrl_amd.policies import run_policy

    g = load_trace("custom3_both_overload_log")
    kw = dict(tables=g.tables, start_rows=g.starts, extrema=g.extrema, start_range=(0, 0))
    dev_env, host_env = FleetVecEnv(g.cfg, g.E, **kw), FleetVecEnv(g.cfg, g.E, **kw)
    dev_env.reset()
    host_env.reset()
    run_policy(dev_env.core.batch, "uncontrolled", 150, chunk=96)
    for _ in range(150):
        host_env.step(np.ones((g.E, g.N), dtype=np.float32))
    for a, b in zip(dev_env.env_method("get_log"), host_env.env_method("get_log")):
        assert len(a) == len(b) > 140 and list(a.columns) == ["Episode", "Time", "Observation", "Action", "Reward", "Cashflow", "Penalties",
                                                             "Grid overloading", "SOC violation", "Degradation", "Charging energy", "SOH"]
        for col in ("Episode", "Reward", "Cashflow", "Penalties", "Grid overloading", "SOC violation"):
            np.testing.assert_allclose(a[col].values.astype(float), b[col].values.astype(float), rtol=1e-12, atol=1e-12, err_msg=col)
        assert (a["Time"].values == b["Time"].values).all()
        for col in ("Charging energy", "SOH", "Observation", "Action", "Degradation"):
            for x, y in zip(a[col], b[col]):
                np.testing.assert_allclose(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), rtol=1e-12, atol=1e-15)
    dev_env.close()
    host_env.close()


def test_mixed_fleet_vec_env_equals_its_groups_stepped_separately():
    """BASELINE.json's largest configuration mixes fleet types: three groups (lmd / ct / ut tables and parameters) behind one
    vector env, every group a handle with its own stream writing into slices of shared device buffers.  Must equal the
    same three groups stepped as separate FleetVecEnvs (same env-id offsets = same start-row streams)."""
    from fleetrl_amd import FleetMixedVecEnv, FleetVecEnv
    from fleetrl_amd.synth import synth_tables
    from test_hip_shapes import _cfg

    N = 12
    specs = [("lmd", 9), ("ct", 17), ("ut", 6)]
    groups = [(_cfg(uc, "rainflow", False), n, dict(tables=synth_tables(uc, N, seed=40 + k), seed=3)) for k, (uc, n) in enumerate(specs)]
    mixed = FleetMixedVecEnv(groups)
    assert mixed.num_envs == 32 and mixed.action_space.shape == (N,)
    singles, off = [], 0
    for cfg, n, kw in groups:
        singles.append(FleetVecEnv(cfg, n, env_id_offset=off, **kw))
        off += n
    obs = mixed.reset()
    np.testing.assert_array_equal(obs, np.concatenate([s.reset() for s in singles]))
    rng = np.random.default_rng(0)
    n_done = 0
    for k in range(110):  # 24 h episodes: one auto-reset per env
        a = rng.uniform(-1, 1, size=(32, N)).astype(np.float32)
        o, r, d, infos = mixed.step(a)
        parts, lo = [], 0
        for s in singles:
            parts.append(s.step(a[lo:lo + s.num_envs]))
            lo += s.num_envs
        np.testing.assert_array_equal(o, np.concatenate([p[0] for p in parts]))
        np.testing.assert_array_equal(r, np.concatenate([p[1] for p in parts]))
        np.testing.assert_array_equal(d, np.concatenate([p[2] for p in parts]))
        want_infos = [x for p in parts for x in p[3]]
        for got, want in zip(infos, want_infos):
            assert set(got) == set(want)
            if "terminal_observation" in want:
                np.testing.assert_array_equal(got["terminal_observation"], want["terminal_observation"])
                assert got["episode"] == want["episode"]
                n_done += 1
    assert n_done == 32
    assert len(mixed.env_method("get_time")) == 32 and mixed.env_method("is_done", indices=[0, 31]) == [False, False]
    mixed.close()
    for s in singles:
        s.close()


def test_drop_in_classes_under_installed_gymnasium_and_sb3_base_classes():
    """With gymnasium / stable-baselines3 importable (stand-ins with the real base classes' constructors, tests/
    test_base_classes.py) the three classes construct through their base classes' __init__ -- SB3's VecEnv asks
    get_attr("render_mode") in there -- pass the isinstance tests SB3 and gymnasium apply, and step as before."""
    import importlib
    import sys

    from test_base_classes import fake_modules

    fakes = fake_modules()
    saved = {k: sys.modules.get(k) for k in list(fakes) + ["fleetrl_amd.spaces", "fleetrl_amd.vec_env"]}
    sys.modules.update(fakes)
    try:
        for name in ("fleetrl_amd.spaces", "fleetrl_amd.vec_env"):
            sys.modules.pop(name, None)
        ve = importlib.import_module("fleetrl_amd.vec_env")
        gym, vec = fakes["gymnasium"], fakes["stable_baselines3.common.vec_env"]
        g = load_trace("lmd1_price_linear")
        kw = dict(tables=g.tables, extrema=g.extrema, start_range=(0, 0))
        venv = ve.FleetVecEnv(g.cfg, 3, start_rows=np.repeat(g.starts[:, :1], 3, axis=1), **kw)
        assert isinstance(venv, vec.VecEnv) and venv.num_envs == 3 and venv.render_mode is None and len(venv.reset_infos) == 3
        assert isinstance(venv.observation_space, gym.spaces.Box) and isinstance(venv.action_space, gym.spaces.Box)
        obs = venv.reset()
        np.testing.assert_array_equal(obs[1], g.reset_obs[0, 0])
        o, r, d, infos = venv.step(np.repeat(g.actions[0, :1], 3, axis=0))   # VecEnv.step of the base class: step_async + step_wait
        np.testing.assert_allclose(o[2], g.obs[0, 0], rtol=1e-5, atol=1e-6)
        assert r.dtype == np.float32 and d.dtype == bool and len(infos) == 3
        venv.close()
        env = ve.FleetEnv(g.cfg, start_rows=g.starts[:, :1], **kw)
        assert isinstance(env, gym.Env)
        ob, info = env.reset()
        np.testing.assert_array_equal(ob, g.reset_obs[0, 0])
        env.close()
        vv = ve.FleetVectorEnv(g.cfg, 2, start_rows=np.repeat(g.starts[:, :1], 2, axis=1), **kw)
        assert isinstance(vv, gym.vector.VectorEnv) and vv.num_envs == 2
        ob, _ = vv.reset()
        o, r, term, trunc, inf = vv.step(np.repeat(g.actions[0, :1], 2, axis=0))
        assert term.dtype == bool and not trunc.any()
        vv.close()
        assert vv.closed
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        import fleetrl_amd

        for attr in ("spaces", "vec_env"):
            mod = sys.modules.get(f"fleetrl_amd.{attr}")
            if mod is not None:
                setattr(fleetrl_amd, attr, mod)


def test_fresh_observation_arrays_equal_the_pinned_ring_buffers():
    """`copy_obs=True` (default; a fresh pageable array per step, like the reference's env) takes the pipelined route of
    fleet_step_host -- pieces on the link through a pinned landing buffer, copied out by worker threads -- and must deliver the
    very bytes the direct transfer into a pinned buffer (`copy_obs=False`) does; sized above the 256 KiB below which the
    transfer is not split, with an odd row count so that the pieces do not end on row boundaries."""
    from fleetrl_amd import FleetVecEnv
    from fleetrl_amd.synth import synth_tables
    from test_hip_shapes import _cfg

    N, E = 50, 333
    tb = synth_tables("ct", N, seed=77)
    fresh = FleetVecEnv(_cfg("ct", "rainflow", False), E, tables=tb, seed=5)
    ring = FleetVecEnv(_cfg("ct", "rainflow", False), E, tables=tb, seed=5, copy_obs=False)
    assert E * fresh.observation_space.shape[0] * 4 > (1 << 18)
    np.testing.assert_array_equal(fresh.reset(), ring.reset())
    rng = np.random.default_rng(1)
    kept = []
    for k in range(120):  # 24 h episodes: across an auto-reset
        a = rng.uniform(-1, 1, size=(E, N)).astype(np.float32)
        o1, r1, d1, i1 = fresh.step(a)
        o2, r2, d2, i2 = ring.step(a)
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(r1, r2)
        np.testing.assert_array_equal(d1, d2)
        kept.append((o1, o2.copy()))
    for o1, o2 in kept[-8:]:  # the fresh arrays are private: later steps did not touch them
        np.testing.assert_array_equal(o1, o2)
    fresh.close()
    ring.close()


def test_two_vec_envs_stepped_from_two_threads_share_the_copy_workers():
    """The host-pointer path's copy workers are one pool per process: two handles stepped concurrently from two Python threads
    (ctypes releases the GIL inside fleet_step_host) must each get their own observations, equal to the same envs stepped alone."""
    import threading

    from fleetrl_amd import FleetVecEnv
    from fleetrl_amd.synth import synth_tables
    from test_hip_shapes import _cfg

    N, E, steps = 50, 300, 40
    tbs = [synth_tables("ct", N, seed=81), synth_tables("ut", N, seed=82)]
    cfgs = [_cfg("ct", "rainflow", False), _cfg("ut", "rainflow", True)]
    acts = [np.random.default_rng(3 + k).uniform(-1, 1, size=(steps, E, N)).astype(np.float32) for k in range(2)]

    def rollout(k, out):
        env = FleetVecEnv(cfgs[k], E, tables=tbs[k], seed=9)
        obs = [env.reset()]
        for a in acts[k]:
            obs.append(env.step(a)[0])
        env.close()
        out[k] = np.stack(obs)

    alone, together = {}, {}
    for k in range(2):
        rollout(k, alone)
    th = [threading.Thread(target=rollout, args=(k, together)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        np.testing.assert_array_equal(together[k], alone[k])
