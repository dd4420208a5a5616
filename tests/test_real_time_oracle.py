"""real_time (event-skipping) mode of FleetEnv.step (fleet_environment.py:453,692-699, event_manager.py:16-31): the CPU
oracle against traces of the unmodified reference run with real_time=True (oracle/gen_golden.py rt)."""
import pytest

from golden_util import RT_TRACE_NAMES, load_rt_trace, params_for, replay_rt
from oracle.fleet_oracle import OracleBatch


@pytest.mark.parametrize("name", RT_TRACE_NAMES)
def test_oracle_real_time_matches_reference(name):
    g = load_rt_trace(name)
    assert g.rc.real_time
    p = params_for(g)
    assert p.real_time == 1
    irregular = "irregular" in g.tables.meta  # set by make_params when the dates are not a regular grid
    assert irregular == name.endswith("irregular")
    # on the irregular grid time_left travels as float32 (5.6333... h is not exact): observation-level tolerance there
    worst = replay_rt(g, OracleBatch(p, g.tables, g.time_feat), float_rtol=1e-12, obs_exact=not irregular)
    assert worst["soc"] == 0.0 or worst["soc"] < 1e-12


def test_real_time_rejects_irregular_grid_and_data_log():
    import numpy as np

    from fleetrl_amd.params import make_params

    g = load_rt_trace("lmd3_price_linear")
    tb = g.tables
    dates = tb.dates.copy()
    dates[5] += np.timedelta64(60, "s")
    import dataclasses
    bad = dataclasses.replace(tb, dates=dates, meta={})
    with pytest.raises(ValueError, match="irregular"):  # price-only observer: the reference raises on such a grid
        make_params(g.rc, bad, 2, extrema=g.extrema, start_range=(0, 0))
    rc_off = dataclasses.replace(g.rc, real_time=False)
    with pytest.raises(ValueError, match="min apart"):
        make_params(rc_off, bad, 2, extrema=g.extrema, start_range=(0, 0))
    rc = dataclasses.replace(g.rc, raw={**g.rc.raw, "log_data": True})
    with pytest.raises(ValueError, match="log_data"):
        make_params(rc, tb, 2, extrema=g.extrema, start_range=(0, 0))
