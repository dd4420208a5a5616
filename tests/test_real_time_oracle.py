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


def test_real_time_rejects_irregular_grid_without_its_tables():
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
    # log_data with real_time is supported (every row of the skipping loop is logged, tests/test_real_time_gpu.py)
    rc = dataclasses.replace(g.rc, raw={**g.rc.raw, "log_data": True})
    assert make_params(rc, tb, 2, extrema=g.extrema, start_range=(0, 0)).log_data == 1


def _irregular_random_params(num_envs):
    """The irregular trace's tables with the RANDOM picker over the whole window (minus room for an episode)."""
    import dataclasses

    from fleetrl_amd.params import make_params

    g = load_rt_trace("lmd1_both_irregular")
    raw = dict(g.rc.raw, time_picker="random")
    from fleetrl_amd.config import resolve_config

    rc = resolve_config(raw)
    p = make_params(rc, g.tables, num_envs, extrema=g.extrema, seed=9)
    return g, rc, p


def test_pickers_on_an_irregular_grid_only_draw_on_grid_rows():
    """The reference's pickers draw from a date_range at the model frequency (random_time_picker.py:25-28): on the irregular
    example that excludes the 00:07 row.  start_lo / start_hi index into the candidate list."""
    import numpy as np

    from fleetrl_amd.params import picker_range

    g, rc, p = _irregular_random_params(400)
    irr = g.tables.meta["irregular"]
    cand = irr["pick_rows"]
    assert cand[0] == 0 and cand[1] == 2 and 1 not in cand  # row 1 is 00:07
    assert np.all(np.diff(g.tables.dates[cand]).astype(np.int64) == 900)
    # the window is shorter than the 60-day end cutoff of the random picker: give the range by hand (indices into cand)
    p.start_lo, p.start_hi = 0, int(cand.size) - 1 - 2 * 96
    eng = OracleBatch(p, g.tables, g.time_feat)
    eng.reset()
    starts = eng.get("start_idx")
    assert np.isin(starts, cand).all() and len(set(starts.tolist())) > 100
    assert eng.get("time_idx").tolist() == starts.tolist()
    eng.close()
    # date-based range on a full-length table: same numbers as the row arithmetic of the regular case
    import dataclasses
    full = dataclasses.replace(g.tables, meta={})
    assert picker_range(rc, full) == (0, full.T - 1 - 60 * 96)
