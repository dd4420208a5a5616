"""HIP product vs the CPU oracle on seeded synthetic inputs across group geometries (N = 1 ... 200 EVs: one lane per
env up to several EVs per lane), ragged batch sizes, all degradation modes and normalisation, with auto-reset over
several episodes.  Needs an MI355X.  Bar: flags/indices bit-exact, float32 obs <= 1e-5 rel, float64 state <= 1e-9 rel."""
import numpy as np
import pytest

from fleetrl_amd.config import resolve_config
from fleetrl_amd.params import make_params, time_features
from fleetrl_amd.synth import synth_tables

pytestmark = pytest.mark.gpu

_TABLES = {}


def _tables(uc, n):
    if (uc, n) not in _TABLES:
        _TABLES[(uc, n)] = synth_tables(uc, n, seed=100 + n)
    return _TABLES[(uc, n)]


def _cfg(uc, deg, norm, aux=True, building=True, pv=True, episode_length=24, real_time=False):
    return {
        "data_path": "<synthetic>", "use_case": uc, "building_name": None, "price_name": None, "tariff_name": None,
        "schedule_name": None, "pv_name": None, "seed": 0, "include_building": building, "include_pv": pv,
        "include_price": True, "time_picker": "random", "max_batt_cap_in_all_use_cases": 60, "init_soh": 1.0,
        "log_data": False, "deg_emp": deg == "linear", "calculate_degradation": deg != "none", "verbose": 0,
        "normalize_in_env": norm, "aux": aux, "ignore_price_reward": False, "ignore_overloading_penalty": False,
        "ignore_invalid_penalty": False, "ignore_overcharging_penalty": False, "gen_schedule": False,
        "gen_start_date": None, "gen_end_date": None, "gen_name": None, "gen_n_evs": 1, "spot_markup": None,
        "spot_mul": None, "feed_in_ded": None, "real_time": real_time, "episode_length": episode_length, "target_soc": 0.85,
    }


def _compare(uc, n_evs, num_envs, deg, norm, steps, aux=True, building=True, pv=True, seed=0, real_time=False, episode_length=24):
    from fleetrl_amd.batch import FleetBatch
    from oracle.fleet_oracle import OracleBatch

    tb = _tables(uc, n_evs)
    rc = resolve_config(_cfg(uc, deg, norm, aux, building, pv, real_time=real_time, episode_length=episode_length))
    p = make_params(rc, tb, num_envs, seed=seed + 1)
    tf = time_features(tb)
    hip, cpu = FleetBatch(p, tb, tf), OracleBatch(p, tb, tf, threads=min(32, max(4, num_envs // 64)))
    rng = np.random.default_rng(seed)
    np.testing.assert_array_equal(hip.reset(), cpu.reset())
    np.testing.assert_array_equal(hip.get("start_idx"), cpu.get("start_idx"))  # same Philox start rows
    n_done = 0
    for s in range(steps):
        mode = (s // 40) % 3
        a = rng.uniform(-1, 1, size=(num_envs, n_evs)) if mode == 0 else rng.uniform(-0.2, 1, size=(num_envs, n_evs)) \
            if mode == 1 else np.full((num_envs, n_evs), 1.0)
        a[rng.random(a.shape) < 0.15] = 0.0
        if real_time:  # quiet envs so that rows are really skipped: half of the envs idle in two steps out of three
            quiet = (np.arange(num_envs) % 2 == 0) & (s % 3 != 0)
            a[quiet] = 0.0
        a = a.astype(np.float32)
        oh, rh, dh, th = hip.step(a)
        oc, rc_, dc, tc = cpu.step(a)
        np.testing.assert_array_equal(dh, dc, err_msg=f"done, step {s}")
        np.testing.assert_allclose(oh, oc, rtol=1e-5, atol=1e-6, err_msg=f"obs, step {s}")
        np.testing.assert_allclose(rh, rc_, rtol=1e-9, atol=1e-9, err_msg=f"reward, step {s}")
        if dh.any():
            n_done += int(dh.sum())
            np.testing.assert_allclose(th[dh.astype(bool)], tc[dc.astype(bool)], rtol=1e-5, atol=1e-6)
        if s % 16 == 0 or s == steps - 1:
            np.testing.assert_array_equal(hip.get("time_idx"), cpu.get("time_idx"))
            np.testing.assert_array_equal(hip.get("hours_left"), cpu.get("hours_left"))
            np.testing.assert_allclose(hip.get("soc"), cpu.get("soc"), rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(hip.get("soh"), cpu.get("soh"), rtol=1e-9)
            np.testing.assert_allclose(hip.get("cashflow"), cpu.get("cashflow"), rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(hip.get("ep_return"), cpu.get("ep_return"), rtol=1e-9, atol=1e-8)
            if deg == "rainflow":
                # (same bounds as the replay of the reference's golden traces, tests/golden_util.py: the cycle stress uses a
                # hardware float32 logarithm inside x^-0.501, <= 4e-9 relative, see pow_m0501 in fleet_kernels.hip)
                np.testing.assert_array_equal(hip.get("rf_len"), cpu.get("rf_len"))
                np.testing.assert_allclose(hip.get("fd_cyc"), cpu.get("fd_cyc"), rtol=1e-8, atol=1e-18)
                np.testing.assert_allclose(hip.get("sei_l"), cpu.get("sei_l"), rtol=1e-9, atol=1e-18)
    hip.check_errors()
    assert not cpu.get("error_bits").any()
    np.testing.assert_array_equal(hip.get("episodes"), cpu.get("episodes"))
    np.testing.assert_allclose(hip.get("last_ep_return"), cpu.get("last_ep_return"), rtol=1e-9, atol=1e-8)
    assert n_done >= num_envs  # every env went through at least one auto-reset
    hip.close()
    cpu.close()


# (N > 64: two wavefronts per env up to 128 EVs -- 100 with an odd number of envs: a partly filled last workgroup behind its barrier --,
# four up to 256, and beyond that one wavefront whose lanes walk several EVs each)
@pytest.mark.parametrize("n_evs,num_envs", [(1, 130), (2, 67), (3, 50), (7, 41), (16, 33), (31, 9), (64, 6), (70, 5), (100, 7), (128, 4),
                                            (130, 5), (200, 3), (256, 3), (257, 2)])
def test_group_geometries_rainflow(n_evs, num_envs):
    _compare("lmd", n_evs, num_envs, "rainflow", False, steps=200)


@pytest.mark.parametrize("deg,norm", [("none", True), ("linear", False), ("linear", True), ("rainflow", True)])
def test_degradation_and_normalisation_modes(deg, norm):
    _compare("ut", 12, 37, deg, norm, steps=200)


@pytest.mark.parametrize("aux,building,pv,norm", [(False, False, False, False), (True, True, False, True), (True, False, True, False),
                                                  (False, True, True, True)])
def test_observer_variants(aux, building, pv, norm):
    _compare("ct", 5, 19, "rainflow", norm, steps=120, aux=aux, building=building, pv=pv)


def test_headline_shape_slice():
    """50 EVs per env (one wavefront per env), caretaker fleet, load+pv, rainflow: the bench workload at 96 envs."""
    _compare("ct", 50, 96, "rainflow", False, steps=220)


def test_baseline_config1_at_its_nominal_size():
    """BASELINE.json configs[1] as it is written: 256 envs x 5 EVs, last-mile delivery, price-only observations, linear
    degradation, 48 h episodes with auto-reset (`bench.py --config c2`), against the oracle over two episodes and into the third
    (the golden trace of this family, trace_lmd5_price_linear, pins 3 envs against the reference itself)."""
    _compare("lmd", 5, 256, "linear", False, steps=400, building=False, pv=False, episode_length=48, seed=5)


def test_long_soak_many_episodes():
    """1000 steps (ten 24 h episodes per env) at the headline geometry: accumulated degradation bookkeeping
    (rainflow_length / fd_cyc / l carried across episodes, quirk Q6) must stay in lock-step with the oracle."""
    _compare("ct", 50, 24, "rainflow", False, steps=1000, seed=3)


def test_float64_actions_and_static_eval_pickers():
    from fleetrl_amd.batch import FleetBatch
    from oracle.fleet_oracle import OracleBatch

    tb = _tables("lmd", 7)
    for picker in ("static", "eval"):
        cfg = _cfg("lmd", "linear", False)
        cfg["time_picker"] = picker
        rc = resolve_config(cfg)
        p = make_params(rc, tb, 11, seed=9)
        tf = time_features(tb)
        hip, cpu = FleetBatch(p, tb, tf), OracleBatch(p, tb, tf)
        np.testing.assert_array_equal(hip.reset(), cpu.reset())
        st = hip.get("start_idx")
        if picker == "static":
            assert (st == 172).all()  # "01/02/2021 19:00" re-based to the table's year
        else:
            assert st.min() >= p.start_lo and st.max() <= p.start_hi and len(set(st.tolist())) > 1
        rng = np.random.default_rng(2)
        for s in range(110):
            a = rng.uniform(-1, 1, size=(11, 7))  # float64, not float32-representable
            oh, rh, dh, _ = hip.step(a)
            oc, rcpu, dc, _ = cpu.step(a)
            np.testing.assert_array_equal(dh, dc)
            np.testing.assert_allclose(oh, oc, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(rh, rcpu, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(hip.get("soc"), cpu.get("soc"), rtol=1e-12, atol=1e-15)
        np.testing.assert_array_equal(hip.get("start_idx"), cpu.get("start_idx"))
        hip.close()
        cpu.close()


@pytest.mark.parametrize("uc,n_evs,num_envs,deg,norm", [
    ("ct", 50, 37, "rainflow", False),   # one wavefront per env: wave-uniform event test
    ("lmd", 5, 67, "linear", True),      # 8 lanes per env: envs of one wavefront leave the skipping loop at different rows
    ("ut", 1, 130, "none", False),       # one lane per env
    ("ct", 130, 9, "rainflow", False),   # several EVs per lane
])
def test_real_time_event_skipping_matches_oracle(uc, n_evs, num_envs, deg, norm):
    """real_time=True: one launch repeats the step with the same action until a relevant event (fleet_environment.py:453,
    692-699); envs advance by different numbers of rows per step."""
    _compare(uc, n_evs, num_envs, deg, norm, steps=260 if n_evs < 100 else 150, seed=11, real_time=True)


def test_full_headline_batch_equals_its_eight_shards():
    """The multi-GPU layout at BASELINE's full size, on one GPU: the 4096 x 50 batch against the eight 512-env shards an 8-GPU
    run would hold (contiguous env ranges, env_id_offset = first global env id, same seed), 48 h episodes stepped over an episode
    end -- observations, rewards, dones and the degradation state bit-identical, i.e. no result depends on which envs share a
    launch (weak scaling needs no data-path collective)."""
    from fleetrl_amd.batch import FleetBatch

    tb = _tables("ct", 50)
    rc = resolve_config(_cfg("ct", "rainflow", False, episode_length=48))
    tf = time_features(tb)
    E, S = 4096, 8
    whole = FleetBatch(make_params(rc, tb, E, seed=5), tb, tf)
    shards = [FleetBatch(make_params(rc, tb, E // S, seed=5, env_id_offset=k * (E // S)), tb, tf) for k in range(S)]
    np.testing.assert_array_equal(whole.reset(), np.concatenate([b.reset() for b in shards]))
    rng = np.random.default_rng(10)
    for s in range(200):
        a = rng.uniform(-1, 1, size=(E, 50)).astype(np.float32)
        a[rng.random((E, 50)) < 0.15] = 0.0
        ow, rw, dw, _ = whole.step(a)
        parts = [b.step(a[k * (E // S):(k + 1) * (E // S)]) for k, b in enumerate(shards)]
        np.testing.assert_array_equal(ow, np.concatenate([p[0] for p in parts]), err_msg=f"obs, step {s}")
        np.testing.assert_array_equal(rw, np.concatenate([p[1] for p in parts]), err_msg=f"reward, step {s}")
        np.testing.assert_array_equal(dw, np.concatenate([p[2] for p in parts]), err_msg=f"done, step {s}")
    for name in ("soc", "soh", "fd_cyc", "rf_len", "time_idx", "last_ep_return", "episodes"):
        np.testing.assert_array_equal(whole.get(name), np.concatenate([b.get(name) for b in shards]), err_msg=name)
    assert whole.get("episodes").min() >= 1
    for b in [whole] + shards:
        b.close()


def test_k_steps_per_launch_equal_single_steps_at_full_size():
    """Size-independent property at BASELINE's full size: K steps in one launch (`fleet_step_many_dev`, the kernel that carries
    the rainflow row head in registers) leave the 4096 x 50 batch in exactly the state K single-step launches do -- state fields
    bit-identical, reward sums to the last place of a re-associated sum -- over episode ends and daily degradation rows, and a
    device-side policy rollout equals the same policy's actions fed step by step."""
    import torch

    from fleetrl_amd import _capi
    from fleetrl_amd.batch import FleetBatch

    tb = _tables("ct", 50)
    rc = resolve_config(_cfg("ct", "rainflow", False, episode_length=48))
    tf = time_features(tb)
    E, N = 4096, 50
    many, single = FleetBatch(make_params(rc, tb, E, seed=7), tb, tf), FleetBatch(make_params(rc, tb, E, seed=7), tb, tf)
    np.testing.assert_array_equal(many.reset(), single.reset())
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(3)
    obs = torch.empty((E, many.obs_dim), device=dev)
    obs1 = torch.empty_like(obs)
    rs = torch.empty(E, device=dev, dtype=torch.float64)
    r1 = torch.empty(E, device=dev, dtype=torch.float64)
    d1 = torch.empty(E, device=dev, dtype=torch.uint8)
    dc = torch.empty(E, device=dev, dtype=torch.int32)
    fields = ("soc", "soh", "hours_left", "fd_cyc", "sei_l", "rf_len", "time_idx", "episodes", "ep_return", "last_ep_return")
    for launch, K in enumerate((64, 96, 61)):  # 221 steps: past the end of the 192-step episodes
        tape = torch.rand((K, E, N), device=dev, generator=gen) * 2 - 1
        tape[torch.rand((K, E, N), device=dev, generator=gen) < 0.15] = 0.0
        want_r = torch.zeros(E, device=dev, dtype=torch.float64)
        want_d = torch.zeros(E, device=dev, dtype=torch.int32)
        torch.cuda.synchronize()  # the handles launch on streams of their own: torch's work on the tape must be done
        many.step_many_dev(K, tape.data_ptr(), obs.data_ptr(), rs.data_ptr(), dc.data_ptr())
        for k in range(K):
            single.step_dev(tape[k].data_ptr(), obs1.data_ptr(), r1.data_ptr(), d1.data_ptr())
            single.synchronize()
            want_r += r1
            want_d += d1.to(torch.int32)
        many.synchronize()
        assert torch.equal(dc, want_d), f"episode ends, launch {launch}"
        torch.testing.assert_close(rs, want_r, rtol=1e-12, atol=1e-9)
        assert torch.equal(obs, obs1), f"last observation, launch {launch}"
        for name in fields:
            np.testing.assert_array_equal(many.get(name), single.get(name), err_msg=f"{name} after launch {launch}")
    assert many.get("episodes").min() >= 1
    # device-side policy ("distributed": clip(get_dist_factor(), 0, 1)) against its own actions fed one step at a time
    K = 48
    act = torch.empty((E, N), device=dev, dtype=torch.float64)
    many.rollout_policy_dev(_capi.POLICY_DISTRIBUTED, K, obs.data_ptr(), rs.data_ptr(), dc.data_ptr())
    for k in range(K):
        act.copy_(torch.from_numpy(np.clip(single.dist_factor(), 0.0, 1.0)))
        torch.cuda.synchronize()
        single.step_dev(act.data_ptr(), obs1.data_ptr(), r1.data_ptr(), d1.data_ptr(), act_dtype=_capi.ACT_F64)
        single.synchronize()
    many.synchronize()
    assert torch.equal(obs, obs1), "last observation of the policy rollout"
    for name in fields:
        np.testing.assert_array_equal(many.get(name), single.get(name), err_msg=f"{name} after the policy rollout")
    many.check_errors()
    single.check_errors()
    many.close()
    single.close()


def test_headline_size_full_batch():
    """BASELINE.json configs[2] at its FULL size and as bench.py runs it -- 4096 envs x 50 EVs, caretaker fleet, load+pv,
    rainflow, 48 h episodes -- HIP against the oracle over two whole episodes and into the third (quirk Q6: rainflow_length /
    fd_cyc / l carried across resets at the headline geometry and length; the oracle runs OpenMP over envs)."""
    _compare("ct", 50, 4096, "rainflow", False, steps=2 * 192 + 12, seed=21, episode_length=48)


def test_c5_shard_size_full_batch():
    """BASELINE.json configs[4], one GPU's shard at FULL size: 8192 envs x 200 EVs (several EVs per lane), rainflow."""
    _compare("lmd", 200, 8192, "rainflow", False, steps=100, seed=22)


def test_c5_mixed_shard_full_batch():
    """BASELINE.json configs[4] as it is written -- one GPU's shard of 8192 envs x 200 EVs with one third each of last-mile,
    caretaker and utility fleets (own tables / parameters per type, three handles on three streams behind
    `FleetMixedVecEnv`), spot_2021-like prices and a fixed feed-in tariff that is NOT the spot price -- against three CPU
    oracles with the same env-id offsets, over an episode end (24 h episodes, 110 steps)."""
    from fleetrl_amd import FleetMixedVecEnv
    from fleetrl_amd.distributed import shard_range
    from oracle.fleet_oracle import OracleBatch

    E, N, steps = 8192, 200, 110
    groups = []
    for k, uc in enumerate(("lmd", "ct", "ut")):
        lo, hi = shard_range(E, 3, k)
        tb = synth_tables(uc, N, seed=300 + k, price_year="2021", feed_in="fixed")
        assert not np.array_equal(tb.tariff, tb.delu)
        groups.append((_cfg(uc, "rainflow", False), hi - lo, dict(tables=tb, seed=7)))
    mixed = FleetMixedVecEnv(groups)
    assert mixed.num_envs == E and mixed.obs_dim == 7 * N + 38
    cpus = [OracleBatch(c.params, c.tables, time_features(c.tables), threads=32) for c in mixed.cores]
    assert [int(c.params.env_id_offset) for c in mixed.cores] == [0, 2731, 5462]
    obs = mixed.reset()
    np.testing.assert_array_equal(obs, np.concatenate([c.reset() for c in cpus]))
    rng = np.random.default_rng(31)
    n_done = 0
    for s in range(steps):
        a = rng.uniform(-1, 1, size=(E, N)).astype(np.float32)
        a[rng.random(a.shape) < 0.15] = 0.0
        o, r, d, infos = mixed.step(a)
        lo = 0
        for c in cpus:
            oc, rc_, dc, tc = c.step(a[lo:lo + c.E])
            np.testing.assert_array_equal(d[lo:lo + c.E], dc.astype(bool), err_msg=f"done, step {s}")
            np.testing.assert_allclose(o[lo:lo + c.E], oc, rtol=1e-5, atol=1e-6, err_msg=f"obs, step {s}")
            np.testing.assert_allclose(r[lo:lo + c.E], rc_, rtol=1e-5, atol=1e-4, err_msg=f"reward (float32 in the VecEnv), step {s}")
            for i in np.nonzero(dc)[0][:8]:
                np.testing.assert_allclose(infos[lo + i]["terminal_observation"], tc[i], rtol=1e-5, atol=1e-6)
            lo += c.E
        n_done += int(d.sum())
    assert n_done >= E  # every env went through an auto-reset
    for core, c in zip(mixed.cores, cpus):
        core.batch.check_errors()
        np.testing.assert_array_equal(core.batch.get("time_idx"), c.get("time_idx"))
        np.testing.assert_allclose(core.batch.get("soc"), c.get("soc"), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(core.batch.get("soh"), c.get("soh"), rtol=1e-9)
        np.testing.assert_array_equal(core.batch.get("rf_len"), c.get("rf_len"))
        np.testing.assert_allclose(core.batch.get("cashflow"), c.get("cashflow"), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(core.batch.get("last_ep_return"), c.get("last_ep_return"), rtol=1e-9, atol=1e-8)
        c.close()
    mixed.close()


def test_c4_shard_size_full_batch():
    """BASELINE.json configs[3], one GPU's shard at full size: 2048 envs x 50 EVs, utility fleet, normalised observations."""
    _compare("ut", 50, 2048, "rainflow", True, steps=200, seed=23)


def test_sharding_invariance_and_determinism():
    """Size-independent properties at the headline geometry: (1) a batch of 2E envs equals two batches of E envs with
    env_id_offset 0 and E (what the multi-GPU layout relies on: start-row streams are keyed by the global env id);
    (2) two identical runs are bit-identical (no atomics / order-dependent sums on the path)."""
    from fleetrl_amd.batch import FleetBatch

    tb = _tables("ct", 50)
    rc = resolve_config(_cfg("ct", "rainflow", False))
    tf = time_features(tb)
    E = 96
    whole = FleetBatch(make_params(rc, tb, 2 * E, seed=5), tb, tf)
    again = FleetBatch(make_params(rc, tb, 2 * E, seed=5), tb, tf)
    lo = FleetBatch(make_params(rc, tb, E, seed=5, env_id_offset=0), tb, tf)
    hi = FleetBatch(make_params(rc, tb, E, seed=5, env_id_offset=E), tb, tf)
    o = whole.reset()
    np.testing.assert_array_equal(o, again.reset())
    np.testing.assert_array_equal(o, np.concatenate([lo.reset(), hi.reset()]))
    rng = np.random.default_rng(9)
    for s in range(130):
        a = rng.uniform(-1, 1, size=(2 * E, 50)).astype(np.float32)
        ow, rw, dw, tw = whole.step(a)
        oa, ra, da, ta = again.step(a)
        ol, rl, dl, tl = lo.step(a[:E])
        oh, rh, dh, th = hi.step(a[E:])
        for x, y in ((ow, oa), (rw, ra), (dw, da)):
            np.testing.assert_array_equal(x, y, err_msg=f"run-to-run, step {s}")
        np.testing.assert_array_equal(ow, np.concatenate([ol, oh]), err_msg=f"sharded obs, step {s}")
        np.testing.assert_array_equal(rw, np.concatenate([rl, rh]))
        np.testing.assert_array_equal(dw, np.concatenate([dl, dh]))
    for name in ("soc", "soh", "fd_cyc", "rf_len", "time_idx", "last_ep_return"):
        np.testing.assert_array_equal(whole.get(name), np.concatenate([lo.get(name), hi.get(name)]), err_msg=name)
        np.testing.assert_array_equal(whole.get(name), again.get(name), err_msg=name)
    assert whole.get("episodes").min() >= 1
    for b in (whole, again, lo, hi):
        b.close()


def _fuzz_cases(n, seed=2024, sizes=(1, 2, 3, 5, 9, 17, 33, 50, 65, 90)):
    """Seeded random corners of the supported matrix: fleet type x EVs per env x observer flags x
    normalisation x degradation model x episode length x real_time."""
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        uc = ["lmd", "ct", "ut"][int(rng.integers(3))]
        n_evs = int(rng.choice(list(sizes)))
        building, pv = bool(rng.integers(2)), bool(rng.integers(2))
        norm = bool(rng.integers(2))
        if norm and pv and not building:
            continue  # crashes in the reference (Q4)
        deg = ["none", "linear", "rainflow"][int(rng.integers(3))]
        aux = bool(rng.integers(4) > 0)
        rt = bool(rng.integers(5) == 0)
        if rt and not (building and pv) and False:
            continue
        ep = int(rng.choice([12, 24, 36]))
        envs = int(rng.integers(3, 40))
        cases.append((uc, n_evs, envs, deg, norm, aux, building, pv, ep, rt))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(24) + _fuzz_cases(10, seed=77, sizes=(100, 128, 130, 160, 200, 256)),
                         ids=lambda c: "-".join(str(x) for x in c))
def test_seeded_random_configurations(case):
    """HIP against the oracle on 34 seeded random configurations (the last ten: envs of two / four wavefronts), 130 steps each with
    auto-reset."""
    uc, n_evs, envs, deg, norm, aux, building, pv, ep, rt = case
    from fleetrl_amd.batch import FleetBatch
    from oracle.fleet_oracle import OracleBatch

    tb = synth_tables(uc, n_evs, seed=900 + n_evs, include_building=building, include_pv=pv)
    cfg = _cfg(uc, deg, norm, aux, building, pv, episode_length=ep, real_time=rt)
    rc = resolve_config(cfg)
    p = make_params(rc, tb, envs, seed=17)
    tf = time_features(tb)
    hip, cpu = FleetBatch(p, tb, tf), OracleBatch(p, tb, tf, threads=8)
    rng = np.random.default_rng(envs * 131 + n_evs)
    np.testing.assert_array_equal(hip.reset(), cpu.reset())
    for s in range(130):
        a = rng.uniform(-1, 1, size=(envs, n_evs))
        a[rng.random(a.shape) < 0.3] = 0.0
        if s % 7 == 3:
            a[:] = 1.0
        a = a.astype(np.float32)
        oh, rh, dh, th = hip.step(a)
        oc, rc_, dc, tc = cpu.step(a)
        np.testing.assert_array_equal(dh, dc, err_msg=f"done, step {s}")
        np.testing.assert_allclose(oh, oc, rtol=1e-5, atol=1e-6, err_msg=f"obs, step {s}")
        np.testing.assert_allclose(rh, rc_, rtol=1e-9, atol=1e-9, err_msg=f"reward, step {s}")
        if dh.any():
            np.testing.assert_allclose(th[dh.astype(bool)], tc[dc.astype(bool)], rtol=1e-5, atol=1e-6)
    for name in ("time_idx", "hours_left", "episodes", "start_idx"):
        np.testing.assert_array_equal(hip.get(name), cpu.get(name), err_msg=name)
    for name in ("soc", "soh", "cashflow", "ep_return", "penalty_record"):
        np.testing.assert_allclose(hip.get(name), cpu.get(name), rtol=1e-9, atol=1e-9, err_msg=name)
    if deg == "rainflow":
        np.testing.assert_array_equal(hip.get("rf_len"), cpu.get("rf_len"))
    hip.check_errors()
    hip.close()
    cpu.close()


@pytest.mark.parametrize("n_evs,num_envs", [(50, 70), (130, 21), (100, 9), (8, 45)])
def test_k_steps_per_launch_match_the_oracle_step_by_step(n_evs, num_envs):
    """`fleet_step_many_dev` at the geometries of the BASELINE shapes -- one env per wavefront with one EV per lane (N = 50: the
    kernel that carries the head of each EV's rainflow row in registers over the K steps), several EVs per lane (N = 130) and
    several envs per wavefront (N = 8) -- against the oracle stepped one row at a time with the same action tape: launches of
    61 steps over 24 h episodes with auto-reset, i.e. episode ends, in-launch resets and daily degradation rows fall INSIDE
    launches, and single steps follow the last launch on the same handle."""
    import torch

    from fleetrl_amd.batch import FleetBatch
    from oracle.fleet_oracle import OracleBatch

    tb = _tables("ct", n_evs)
    rc = resolve_config(_cfg("ct", "rainflow", False))
    p = make_params(rc, tb, num_envs, seed=11)
    tf = time_features(tb)
    hip, cpu = FleetBatch(p, tb, tf), OracleBatch(p, tb, tf, threads=8)
    np.testing.assert_array_equal(hip.reset(), cpu.reset())
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    K, launches = 61, 5
    obs = torch.empty((num_envs, hip.obs_dim), device=dev)
    rs = torch.empty(num_envs, device=dev, dtype=torch.float64)
    dc = torch.empty(num_envs, device=dev, dtype=torch.int32)

    def check(where):
        np.testing.assert_array_equal(hip.get("time_idx"), cpu.get("time_idx"), err_msg=where)
        np.testing.assert_array_equal(hip.get("episodes"), cpu.get("episodes"), err_msg=where)
        np.testing.assert_array_equal(hip.get("hours_left"), cpu.get("hours_left"), err_msg=where)
        np.testing.assert_array_equal(hip.get("rf_len"), cpu.get("rf_len"), err_msg=where)
        np.testing.assert_allclose(hip.get("soc"), cpu.get("soc"), rtol=1e-9, atol=1e-12, err_msg=where)
        np.testing.assert_allclose(hip.get("soh"), cpu.get("soh"), rtol=1e-9, err_msg=where)
        np.testing.assert_allclose(hip.get("fd_cyc"), cpu.get("fd_cyc"), rtol=1e-8, atol=1e-18, err_msg=where)
        np.testing.assert_allclose(hip.get("sei_l"), cpu.get("sei_l"), rtol=1e-9, atol=1e-18, err_msg=where)
        np.testing.assert_allclose(hip.get("ep_return"), cpu.get("ep_return"), rtol=1e-9, atol=1e-8, err_msg=where)

    for l in range(launches):
        acts = rng.uniform(-1, 1, size=(K, num_envs, n_evs)).astype(np.float32)
        acts[rng.random(acts.shape) < 0.15] = 0.0
        tape = torch.from_numpy(acts).to(dev)
        hip.step_many_dev(K, tape.data_ptr(), obs.data_ptr(), rs.data_ptr(), dc.data_ptr())
        hip.synchronize()
        want_r, want_d = np.zeros(num_envs), np.zeros(num_envs, dtype=np.int32)
        for k in range(K):
            oc, r, d, _t = cpu.step(acts[k])
            want_r += r
            want_d += d
        np.testing.assert_array_equal(dc.cpu().numpy(), want_d, err_msg=f"episode ends, launch {l}")
        np.testing.assert_allclose(rs.cpu().numpy(), want_r, rtol=1e-9, atol=1e-7, err_msg=f"reward sums, launch {l}")
        np.testing.assert_allclose(obs.cpu().numpy(), oc, rtol=1e-5, atol=1e-6, err_msg=f"last observation, launch {l}")
        check(f"after launch {l}")
    assert hip.get("episodes").min() >= 2  # every env went through in-launch resets
    for s in range(40):  # the single-step kernel continues from what the K-step kernel left behind
        a = rng.uniform(-1, 1, size=(num_envs, n_evs)).astype(np.float32)
        oh, rh, dh, _ = hip.step(a)
        oc, rcpu, dcpu, _ = cpu.step(a)
        np.testing.assert_array_equal(dh, dcpu)
        np.testing.assert_allclose(oh, oc, rtol=1e-5, atol=1e-6, err_msg=f"obs, single step {s} after the launches")
        np.testing.assert_allclose(rh, rcpu, rtol=1e-9, atol=1e-9)
    check("after the single steps")
    hip.check_errors()
    assert not cpu.get("error_bits").any()
    hip.close()
    cpu.close()
