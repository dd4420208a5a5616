"""Device-side benchmark policies (`fleet_rollout_policy_dev`): K steps in one launch with the action rule of the
reference's harness evaluated on the GPU, against the CPU oracle driven step by step with host-computed actions
(uncontrolled: np.ones, benchmarking/uncontrolled_charging.py:51-54; distributed: clip(get_dist_factor(), 0, 1),
benchmarking/distributed_charging.py:50-54).  Needs an MI355X."""
import numpy as np
import pytest

from golden_util import load_trace, params_for
from fleetrl_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("policy", [_capi.POLICY_UNCONTROLLED, _capi.POLICY_DISTRIBUTED])
@pytest.mark.parametrize("name", ["ct5_both_rainflow", "lmd5_price_linear"])
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
    hip.check_errors()
