"""Device-side benchmark policies (`fleet_rollout_policy_dev`): K steps in one launch with the action rule of the
reference's harness evaluated on the GPU, against the CPU oracle driven step by step with host-computed actions
(uncontrolled: np.ones, benchmarking/uncontrolled_charging.py:51-54; distributed: clip(get_dist_factor(), 0, 1),
benchmarking/distributed_charging.py:50-54).  Needs an MI355X."""
import numpy as np
import pytest

from golden_util import load_trace, params_for
from fleetrl_amd import _capi

pytestmark = pytest.mark.gpu


# the two *_q7* traces: init_soh at / just above 0.9, so the sticky raised target (quirk Q7) feeds the distributed rule
# (get_dist_factor uses target_soc, fleet_environment.py:782-799) and the reset laxity fix-up of the episodes after it
@pytest.mark.parametrize("policy", [_capi.POLICY_UNCONTROLLED, _capi.POLICY_DISTRIBUTED])
@pytest.mark.parametrize("name", ["ct5_both_rainflow", "lmd5_price_linear", "lmd3_price_linear_q7cross", "ct3_both_nodeg_q7sticky"])
def test_policy_rollout_matches_stepwise_oracle(name, policy):
    import torch

    from fleetrl_amd.batch import FleetBatch
    from oracle.fleet_oracle import OracleBatch

    g = load_trace(name)
    E, K = 21, 230  # more than one 48 h episode: auto-reset inside the launch
    p = params_for(g, num_envs=E)
    rng = np.random.default_rng(1)
    starts = rng.integers(0, g.tables.T - g.ep_steps - 60, size=(3, E)).astype(np.int32)
    hip, cpu = FleetBatch(p, g.tables, g.time_feat), OracleBatch(p, g.tables, g.time_feat)
    for eng in (hip, cpu):
        eng.set_start_schedule(starts)
    np.testing.assert_array_equal(hip.reset(), cpu.reset())
    dev = torch.device("cuda", 0)
    obs = torch.zeros((E, hip.obs_dim), device=dev, dtype=torch.float32)
    rsum = torch.zeros(E, device=dev, dtype=torch.float64)
    dcount = torch.zeros(E, device=dev, dtype=torch.int32)
    hip.rollout_policy_dev(policy, K, obs.data_ptr(), rsum.data_ptr(), dcount.data_ptr())
    hip.synchronize()
    want_r = np.zeros(E)
    want_d = np.zeros(E, dtype=np.int32)
    for _ in range(K):
        a = np.ones((E, g.N)) if policy == _capi.POLICY_UNCONTROLLED else np.clip(cpu.dist_factor(), 0, 1)
        o, r, d, _t = cpu.step(a.astype(np.float64))
        want_r += r
        want_d += d
    np.testing.assert_array_equal(dcount.cpu().numpy(), want_d)
    assert want_d.min() >= 1
    np.testing.assert_allclose(rsum.cpu().numpy(), want_r, rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(obs.cpu().numpy(), o, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(hip.get("time_idx"), cpu.get("time_idx"))
    np.testing.assert_allclose(hip.get("soc"), cpu.get("soc"), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(hip.get("soh"), cpu.get("soh"), rtol=1e-9)
    np.testing.assert_array_equal(hip.get("target_soc"), cpu.get("target_soc"))
    if "q7" in name:
        assert (hip.get("target_soc") == 0.9).any()
    hip.check_errors()


@pytest.mark.parametrize("window", ["derived", (1, 30, 3), (23, 45, 5), (24, 0, 2)])
@pytest.mark.parametrize("name", ["ct5_both_rainflow", "lmd5_price_linear"])
def test_night_policy_rollout_matches_stepwise_oracle(name, window):
    """FLEET_ACT_POLICY_NIGHT (per-env window state on the device, split over several launches) against the oracle
    driven with the rule restatement `oracle.fleet_oracle.NightChargingRule`, one instance per env."""
    import torch

    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.policies import night_schedule
    from oracle.fleet_oracle import NightChargingRule, OracleBatch

    g = load_trace(name)
    E = 19
    chunks = (1, 100, 7, 150)  # the window state has to survive launch boundaries and episode resets
    p = params_for(g, num_envs=E)
    if window == "derived":
        window = night_schedule(g.tables, target_soc=p.target_soc, init_battery_cap=p.init_battery_cap,
                                charging_eff=p.charging_eff, evse_power=p.evse_power)
    rng = np.random.default_rng(5)
    starts = rng.integers(0, g.tables.T - g.ep_steps - 60, size=(3, E)).astype(np.int32)
    hip, cpu = FleetBatch(p, g.tables, g.time_feat), OracleBatch(p, g.tables, g.time_feat)
    for eng in (hip, cpu):
        eng.set_start_schedule(starts)
    np.testing.assert_array_equal(hip.reset(), cpu.reset())
    hip.set_night_policy(*window)
    rules = [NightChargingRule(window[0], window[1], window[2], g.rc.minutes, bool(p.is_caretaker)) for _ in range(E)]
    dev = torch.device("cuda", 0)
    obs = torch.zeros((E, hip.obs_dim), device=dev, dtype=torch.float32)
    rsum = torch.zeros(E, device=dev, dtype=torch.float64)
    dcount = torch.zeros(E, device=dev, dtype=torch.int32)
    got_r = np.zeros(E)
    got_d = np.zeros(E, dtype=np.int64)
    for K in chunks:
        hip.rollout_policy_dev(_capi.POLICY_NIGHT, K, obs.data_ptr(), rsum.data_ptr(), dcount.data_ptr())
        hip.synchronize()
        got_r += rsum.cpu().numpy()
        got_d += dcount.cpu().numpy()
    want_r = np.zeros(E)
    want_d = np.zeros(E, dtype=np.int64)
    n_ones = n_zeros = 0
    for _ in range(sum(chunks)):
        t = cpu.get("time_idx")
        df = cpu.dist_factor()
        a = np.stack([rules[e].action(int(t[e]), int(g.tables.hour[t[e]]), int(g.tables.minute[t[e]]), g.N, df[e]) for e in range(E)])
        n_ones += int((a == 1).all(axis=1).sum())
        n_zeros += int((a == 0).all(axis=1).sum())
        o, r, d, _t = cpu.step(a.astype(np.float64))
        want_r += r
        want_d += d
    if window[0] < 24:
        assert n_ones > 0
    assert n_zeros > 0
    np.testing.assert_array_equal(got_d, want_d)
    assert want_d.min() >= 1
    np.testing.assert_allclose(got_r, want_r, rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(obs.cpu().numpy(), o, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(hip.get("time_idx"), cpu.get("time_idx"))
    np.testing.assert_allclose(hip.get("soc"), cpu.get("soc"), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(hip.get("soh"), cpu.get("soh"), rtol=1e-9)
    hip.check_errors()


def test_night_policy_needs_configuration():
    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd._capi import FleetHipError
    import torch

    g = load_trace("lmd5_price_linear")
    hip = FleetBatch(params_for(g, num_envs=4), g.tables, g.time_feat)
    hip.reset()
    dev = torch.device("cuda", 0)
    obs = torch.zeros((4, hip.obs_dim), device=dev, dtype=torch.float32)
    rsum = torch.zeros(4, device=dev, dtype=torch.float64)
    with pytest.raises(FleetHipError, match="fleet_set_night_policy"):
        hip.rollout_policy_dev(_capi.POLICY_NIGHT, 4, obs.data_ptr(), rsum.data_ptr(), None)
    with pytest.raises(FleetHipError, match="out of range"):
        hip.set_night_policy(25, 0, 3)


def test_run_policy_driver():
    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.policies import run_policy

    g = load_trace("ct5_both_rainflow")
    E = 8
    a, b = (FleetBatch(params_for(g, num_envs=E), g.tables, g.time_feat) for _ in range(2))
    starts = np.full((1, E), 10, dtype=np.int32)
    for eng in (a, b):
        eng.set_start_schedule(starts)
        eng.reset()
    o1, r1, d1 = run_policy(a, "night", 200, chunk=64, night=(23, 45, 3))
    o2, r2, d2 = run_policy(b, "night", 200, chunk=200, night=(23, 45, 3))
    np.testing.assert_array_equal(o1, o2)
    np.testing.assert_allclose(r1, r2, rtol=1e-12)  # the sum over chunks associates differently
    np.testing.assert_array_equal(d1, d2)
    assert d1.min() >= 1
