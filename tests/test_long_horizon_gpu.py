"""How long an episode may be.  The reference accepts any `episode_length` (time_config.py:1-24,
fleet_environment.py:355).  The HIP path is float64 in the reference's operation order, so SOC is bit-identical until the first
degradation update; after it SoH agrees to ~1e-13 only (the cycle stress uses a hardware log / reciprocal square root,
DESIGN.md section 5), a SATURATED SOC (charged into the target, discharged to empty) may then differ in its last bit, and the
reference's reversal extraction compares samples exactly.  These tests bound what that means with the bench's own action
distribution -- uniform(-1, 1), 15 % zeros, i.e. saturating actions included -- on the headline geometry (50 EVs, caretaker
fleet, load + pv, rainflow), 8 envs = 8 independent start rows: 7-day and 30-day episodes, HIP against the oracle on EVERY
step, observations to the north-star tolerance (1e-5), done flags / time rows / rainflow_length exact."""
import numpy as np
import pytest

from fleetrl_amd.config import resolve_config
from fleetrl_amd.params import make_params, time_features
from fleetrl_amd.synth import synth_tables
from test_rainflow_adversarial_gpu import _cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("days", [7, 30])
def test_long_episodes_with_saturating_actions_stay_inside_the_tolerance(days):
    from fleetrl_amd.batch import FleetBatch
    from oracle.fleet_oracle import OracleBatch

    N, E = 50, 8
    tb = synth_tables("ct", N, seed=1234)
    cfg = dict(_cfg(24 * days), gen_n_evs=N)
    rc = resolve_config(cfg)
    p = make_params(rc, tb, E, seed=21, start_range=(0, 96 * 200))
    tf = time_features(tb)
    hip, cpu = FleetBatch(p, tb, tf), OracleBatch(p, tb, tf, threads=8)
    np.testing.assert_array_equal(hip.reset(), cpu.reset())
    rng = np.random.default_rng(100 + days)
    steps = 96 * days
    worst = 0.0
    for s in range(steps):
        a = rng.uniform(-1, 1, size=(E, N)).astype(np.float32)
        a[rng.random(a.shape) < 0.15] = 0.0
        oh, rh, dh, _ = hip.step(a)
        oc, rcpu, dc, _ = cpu.step(a)
        np.testing.assert_array_equal(dh, dc, err_msg=f"done, step {s}")
        err = np.abs(oh - oc) / (1e-6 / 1e-5 + np.abs(oc))      # relative error with the tests' usual absolute floor (atol 1e-6)
        worst = max(worst, float(err.max()))
        assert worst <= 1e-5, f"observation off by {worst:.2e} (relative) at step {s} of a {days}-day episode"
        np.testing.assert_allclose(rh, rcpu, rtol=1e-5, atol=1e-6, err_msg=f"reward, step {s}")
    assert dh.all()
    np.testing.assert_array_equal(hip.get("time_idx"), cpu.get("time_idx"))
    np.testing.assert_array_equal(hip.get("rf_len"), cpu.get("rf_len"))
    np.testing.assert_allclose(hip.get("soh"), cpu.get("soh"), rtol=1e-9)
    np.testing.assert_allclose(hip.get("soc"), cpu.get("soc"), rtol=1e-5, atol=1e-9)
    hip.check_errors()
    print(f"{days}-day episode, {E} envs x {N} EVs x {steps} steps: worst relative observation error {worst:.2e}")
    hip.close()
    cpu.close()
