"""CPU oracle (oracle/fleet_oracle.c) against the golden traces produced from the unmodified reference.

This is what pins the oracle: flags, indices and the float32 observation words must be
bit-identical; float64 reward / cashflow / SOC / SoH must agree to 1e-9 relative (the north-star tolerance is 1e-5).
"""
import numpy as np
import pytest

from golden_util import TRACE_NAMES, load_trace, params_for, replay
from oracle.fleet_oracle import OracleBatch, load


@pytest.mark.parametrize("name", TRACE_NAMES)
def test_oracle_matches_reference_trace(name):
    g = load_trace(name)
    eng = OracleBatch(params_for(g), g.tables, g.time_feat)
    assert eng.obs_dim == int(g.sc_obs_dim)
    worst = replay(g, eng, float_rtol=1e-9, obs_exact=True)
    print(name, worst)
    assert worst["reward"] < 1e-9 and worst["soc"] < 1e-9 and worst["soh"] < 1e-12


@pytest.mark.parametrize("name", TRACE_NAMES)
def test_params_match_reference_scalars(name):
    """grid sizing, EVSE/battery constants, scaled price multiplier and normaliser constants equal the reference's."""
    g = load_trace(name)
    p = params_for(g)
    assert p.grid_connection == float(g.sc_grid_connection)
    assert p.evse_power == float(g.sc_evse_power)
    assert p.batt_cap_nominal == float(g.sc_batt_cap_nominal)
    assert p.init_battery_cap == float(g.sc_init_battery_cap)
    assert p.price_multiplier == float(g.sc_price_multiplier)
    if g.rc.normalize_in_env:
        for k in ("max_time_left", "max_price", "min_price", "max_tariff", "min_tariff"):
            assert getattr(p, k) == float(getattr(g, f"sc_{k}")), k
        if g.rc.include_building:
            assert p.max_building == float(g.sc_max_building)
        if g.rc.include_pv:
            assert p.max_pv == float(g.sc_max_pv)
        if g.rc.aux:
            assert p.max_hours_needed == float(g.sc_max_hours_needed)
        assert float(g.sc_obs_low) == 0.0 and float(g.sc_obs_high) == 1.0
    else:
        assert np.isinf(float(g.sc_obs_low)) and np.isinf(float(g.sc_obs_high))


def test_known_answer_penalties():
    """Anchor K6: the two sigmoid penalties at fixed points, values captured from the reference."""
    lib = load()
    got = [lib.oracle_soc_violation_penalty(x) for x in (0.1, 0.3, 0.5)]
    np.testing.assert_allclose(got, [-19.15611261, -264.849956, -483.2214836], rtol=1e-9)
    got = [lib.oracle_overloading_penalty(x, 1.0) for x in (1.05, 1.1, 1.3, 2.0)]
    np.testing.assert_allclose(got, [0, -17.3063569, -260.9522766, -699.9811291], rtol=1e-9)
