"""Pre-stager (fleetrl_amd/prestage.py) against the reference's own `db` columns stored in the golden traces.

Needs the reference's input CSVs, so it only runs in the build container (marker `reference`)."""
import os

import numpy as np
import pytest

from golden_util import TRACE_NAMES, load_trace
from fleetrl_amd.params import table_extrema
from fleetrl_amd.prestage import build_tables_from_config, load_schedule_csv, stack_single_ev_schedules

INPUTS = "/root/reference/inputs"


@pytest.mark.reference
@pytest.mark.parametrize("name", TRACE_NAMES)
def test_prestage_bit_identical_to_reference_db(name):
    g = load_trace(name)
    cfg = dict(g.cfg)
    cfg["data_path"] = INPUTS
    sched = None
    if g.N > 1:
        uc = "lmd" if cfg["use_case"] == "custom" else cfg["use_case"]
        sched = stack_single_ev_schedules(load_schedule_csv(os.path.join(INPUTS, f"1_{uc}.csv")),
                                          load_schedule_csv(os.path.join(INPUTS, f"1_{uc}_eval.csv")), g.N)
    tb = build_tables_from_config(cfg, schedule=sched)
    w0 = int(g.sc_window_row0)
    w1 = w0 + g.tables.T
    assert np.array_equal(tb.dates[w0:w1], g.tables.dates)
    for k in ("there", "time_left", "soc_on_return"):
        assert np.array_equal(getattr(tb, k)[w0:w1], getattr(g.tables, k)), k
    for k in ("delu", "tariff", "prc", "trc", "load", "pv"):
        assert np.array_equal(getattr(tb, k)[w0:w1], getattr(g.tables, k), equal_nan=True), k
    ext = table_extrema(tb)
    for k, v in g.extrema.items():
        if k == "max_load" and not cfg["include_building"]:
            continue
        if k == "max_pv" and not cfg["include_pv"]:
            continue
        assert ext[k] == v, k


def test_kahan_group_sum_matches_sequential_compensated_sum():
    from fleetrl_amd.prestage import _kahan_group_sum

    rng = np.random.default_rng(1)
    vals = rng.random(1000) * 3
    label = np.sort(rng.integers(0, 40, size=1000))
    label = np.unique(label, return_inverse=True)[1]
    first = np.concatenate(([True], label[1:] != label[:-1]))
    got = _kahan_group_sum(label, vals, int(label.max()) + 1, first)
    for gidx in range(int(label.max()) + 1):
        s = c = 0.0
        for v in vals[label == gidx]:
            y = v - c
            t = s + y
            c = t - s - y
            s = t
        assert got[gidx] == s


@pytest.mark.reference
def test_prestage_real_time_keeps_the_irregular_rows():
    """real_time: no resampling (data_processing.py:54-62, 73-82); the reference's own irregular example inputs/test_lmd.csv
    against the `db` columns stored in the real_time trace made from it (window starting at row 0)."""
    from golden_util import load_rt_trace

    g = load_rt_trace("lmd1_both_irregular")
    cfg = dict(g.cfg)
    cfg["data_path"] = INPUTS
    tb = build_tables_from_config(cfg)
    n = g.tables.T
    assert tb.T == 35041  # one row more than a year of quarter hours: 00:07
    assert np.array_equal(tb.dates[:n], g.tables.dates)
    assert (np.diff(tb.dates[:4]).astype(np.int64) == np.array([420, 480, 900])).all()
    for k in ("there", "time_left", "soc_on_return"):
        assert np.array_equal(getattr(tb, k)[:n], getattr(g.tables, k)), k
    for k in ("delu", "tariff", "prc", "trc", "load", "pv"):
        assert np.array_equal(getattr(tb, k)[:n], getattr(g.tables, k), equal_nan=True), k
