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
    worst = replay_rt(g, OracleBatch(p, g.tables, g.time_feat), float_rtol=1e-12)
    assert worst["soc"] == 0.0 or worst["soc"] < 1e-12


def test_real_time_rejects_irregular_grid_and_data_log():
    import numpy as np

    from fleetrl_amd.params import make_params

    g = load_rt_trace(RT_TRACE_NAMES[0])
    tb = g.tables
    dates = tb.dates.copy()
    dates[5] += np.timedelta64(60, "s")
    import dataclasses
    bad = dataclasses.replace(tb, dates=dates)
    with pytest.raises(ValueError, match="irregular"):
        make_params(g.rc, bad, 2, extrema=g.extrema, start_range=(0, 0))
    rc = dataclasses.replace(g.rc, raw={**g.rc.raw, "log_data": True})
    with pytest.raises(ValueError, match="log_data"):
        make_params(rc, tb, 2, extrema=g.extrema, start_range=(0, 0))
