"""Host pre-stage: schedule / price / load / PV CSVs -> dense tables for HBM.

Replaces the reference's one-off pandas pipeline
  * DataLoader.__init__ / compute_from_schedule / load_prices / load_feed_in /
    load_building_load / load_pv   (/root/reference/fleetrl/utils/data_processing/data_processing.py:21-370)
  * DataLoader.shape_price_reward    (data_processing.py:373-416)
  * FleetEnv.adjust_caretaker_lunch_soc (/root/reference/fleetrl/fleet_env/fleet_environment.py:951-967)
with O(T*N) NumPy code that emits the layout the device consumes (DESIGN.md "Data layout"):

  per (time row t, EV c)   [T, N]: there u8, time_left f32 (multiples of dt, exact), soc_on_return f64
  per time row t           [T]   : delu, tariff, prc, trc, load, pv  f64;  hour u8, minute u8, month u8, weekday u8

The reference keeps everything in one (N*T)-row DataFrame whose row for car c at slot t is c*T+t
and looks values up by boolean scans; here a lookup is `table[t, c]`.

pandas is used only to parse CSV text; all arithmetic is NumPy float64 in the reference's
operation order so the tables are bit-identical to the reference's `db` columns
(checked in tests/test_prestage.py against tests/golden/tables_*.npz).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

__all__ = [
    "Schedule",
    "FleetTables",
    "load_schedule_csv",
    "stack_single_ev_schedules",
    "schedule_tables",
    "upsample_backward",
    "detrend_monthly",
    "build_tables_from_config",
    "build_tables",
]


# --------------------------------------------------------------------------------------
# containers
# --------------------------------------------------------------------------------------
@dataclass
class Schedule:
    """Raw schedule rows sorted by (ID, date) -- the reference's `self.schedule` before processing."""

    date: np.ndarray  # datetime64[s] [R]
    ev_id: np.ndarray  # int64 [R]
    consumption: np.ndarray  # float64 [R]  Consumption_kWh
    power_rating: np.ndarray  # float64 [R]  PowerRating_kW
    station_none: np.ndarray  # bool [R]     ChargingStation == "none"
    station_code: np.ndarray  # int64 [R]    factorised ChargingStation (for change detection)

    @property
    def num_cars(self) -> int:
        return int(self.ev_id.max()) + 1


@dataclass
class FleetTables:
    """Dense pre-staged tables (host copies; `FleetVecEnv` uploads them once)."""

    dates: np.ndarray  # datetime64[s] [T]
    there: np.ndarray  # u8  [T, N]
    time_left: np.ndarray  # f32 [T, N]
    soc_on_return: np.ndarray  # f64 [T, N]
    consumption: np.ndarray  # f64 [T, N]  last_trip_total_consumption (kept for the ct fix-up / tests)
    delu: np.ndarray  # f64 [T]
    tariff: np.ndarray  # f64 [T]
    prc: np.ndarray  # f64 [T]  price_reward_curve
    trc: np.ndarray  # f64 [T]  tariff_reward_curve
    load: np.ndarray  # f64 [T]  (zeros when building load is not included)
    pv: np.ndarray  # f64 [T]  (zeros when pv is not included)
    hour: np.ndarray  # u8 [T]
    minute: np.ndarray  # u8 [T]
    month: np.ndarray  # u8 [T]
    weekday: np.ndarray  # u8 [T]
    minutes_per_step: int = 15
    meta: dict = field(default_factory=dict)

    @property
    def T(self) -> int:
        return int(self.there.shape[0])

    @property
    def N(self) -> int:
        return int(self.there.shape[1])


# --------------------------------------------------------------------------------------
# CSV parsing (pandas only parses text)
# --------------------------------------------------------------------------------------
def _to_s(dates) -> np.ndarray:
    return np.asarray(dates, dtype="datetime64[s]")


def load_schedule_csv(path: str) -> Schedule:
    """Parse a FleetRL/emobpy-style schedule CSV (columns date, Consumption_kWh, ChargingStation,
    ID, PowerRating_kW, ...).  Mirrors `pd.read_csv(..., parse_dates=["date"])`
    (data_processing.py:50); rows are stably sorted by (ID, date) like the reference's
    `groupby("ID").resample(freq)` output (data_processing.py:61-64)."""
    import pandas as pd

    df = pd.read_csv(path, parse_dates=["date"])
    order = np.lexsort((df["date"].values, df["ID"].values))
    df = df.iloc[order]
    station = df["ChargingStation"].astype(str).values
    codes, _ = pd.factorize(station)
    return Schedule(
        date=_to_s(df["date"].values),
        ev_id=df["ID"].values.astype(np.int64),
        consumption=df["Consumption_kWh"].values.astype(np.float64),
        power_rating=df["PowerRating_kW"].values.astype(np.float64),
        station_none=(station == "none"),
        station_code=codes.astype(np.int64),
    )


def stack_single_ev_schedules(base: Schedule, alt: Schedule | None, n_evs: int, shift_rows: int = 7 * 96) -> Schedule:
    """Build an N-EV schedule from shipped single-EV files (the multi-EV schedule
    blobs `inputs/2_*.csv` are missing from the reference checkout).  Car i takes `base` when i is even
    else `alt`; its per-row payload columns are rolled by (i//2)*shift_rows rows (whole weeks, so
    weekdays stay aligned); dates are kept; ID = i."""
    parts = {k: [] for k in ("date", "ev_id", "consumption", "power_rating", "station_none", "station_code")}
    for i in range(n_evs):
        src = base if (i % 2 == 0 or alt is None) else alt
        if src.num_cars != 1:
            raise ValueError("stacking expects single-EV schedules")
        k = (i // 2) * shift_rows
        parts["date"].append(src.date)
        parts["ev_id"].append(np.full(src.date.shape, i, dtype=np.int64))
        parts["consumption"].append(np.roll(src.consumption, k))
        parts["power_rating"].append(np.roll(src.power_rating, k))
        parts["station_none"].append(np.roll(src.station_none, k))
        parts["station_code"].append(np.roll(src.station_code, k))
    return Schedule(**{k: np.concatenate(v) for k, v in parts.items()})


def _resample_regular(s: Schedule, minutes: int) -> Schedule:
    """Equivalent of `groupby("ID").resample(freq).agg(first/sum/mean)` (data_processing.py:61-64) for
    schedules whose rows fall into regular `minutes` buckets.  Rows already on the grid (all shipped
    files) pass through untouched; finer rows are aggregated (sum consumption, mean power, first station)."""
    step = np.timedelta64(minutes * 60, "s")
    out = {k: [] for k in ("date", "ev_id", "consumption", "power_rating", "station_none", "station_code")}
    for c in range(s.num_cars):
        m = s.ev_id == c
        d = s.date[m]
        if d.size == 0:
            raise ValueError(f"schedule has no rows for ID {c}")
        origin = d[0].astype("datetime64[D]").astype("datetime64[s]")
        b = ((d - origin) // step).astype(np.int64)
        b -= b[0]
        nb = int(b[-1]) + 1
        if nb == d.size and np.all(np.diff(b) == 1) and np.all((d - d[0]) == b * step):
            for k, v in (("date", d), ("ev_id", s.ev_id[m]), ("consumption", s.consumption[m]),
                         ("power_rating", s.power_rating[m]), ("station_none", s.station_none[m]),
                         ("station_code", s.station_code[m])):
                out[k].append(v)
            continue
        counts = np.bincount(b, minlength=nb)
        if np.any(counts == 0):
            raise ValueError("schedule has empty time buckets after resampling; fill the gaps first")
        first = np.concatenate(([0], np.cumsum(counts)[:-1]))
        start = origin + ((d[0] - origin) // step) * step
        out["date"].append(start + np.arange(nb) * step)
        out["ev_id"].append(np.full(nb, c, dtype=np.int64))
        out["consumption"].append(np.bincount(b, weights=s.consumption[m], minlength=nb))
        out["power_rating"].append(np.bincount(b, weights=s.power_rating[m], minlength=nb) / counts)
        out["station_none"].append(s.station_none[m][first])
        out["station_code"].append(s.station_code[m][first])
    return Schedule(**{k: np.concatenate(v) for k, v in out.items()})


# --------------------------------------------------------------------------------------
# schedule -> there / time_left / soc_on_return     (data_processing.py:120-223)
# --------------------------------------------------------------------------------------
def schedule_tables(s: Schedule, minutes: int, target_soc: float, init_battery_cap: float):
    """Returns (dates[T], there[T,N] u8, time_left[T,N] f64, soc_on_return[T,N] f64, consumption[T,N] f64).

    Trip = maximal run of rows with ChargingStation == "none" *in the concatenated (ID, date) frame*
    -- the reference detects station changes with one `shift(1)` over the whole frame
    (data_processing.py:133-136), so a run may in principle straddle a car boundary; the trip is then
    credited to the ID of its first row (:165).  That is replicated, not fixed.
    """
    R = s.date.size
    N = s.num_cars
    if R % N:
        raise ValueError("every EV must cover the same date range")
    T = R // N
    ids = s.ev_id
    if not np.array_equal(ids, np.repeat(np.arange(N), T)):
        raise ValueError("schedule rows must be sorted by (ID, date) with equal row counts per ID")
    dates = s.date[:T]
    for c in range(1, N):
        if not np.array_equal(s.date[c * T:(c + 1) * T], dates):
            raise ValueError("every EV must cover the same dates")

    there = (s.power_rating != 0)
    # station change -> group id (one pass over the concatenated frame)
    change = np.ones(R, dtype=bool)
    change[1:] = s.station_code[1:] != s.station_code[:-1]
    group = np.cumsum(change)
    none = s.station_none
    rows = np.nonzero(none)[0]

    cons = np.zeros(R)  # last_trip_total_consumption per row
    tl = np.zeros(R)  # time_left (hours)
    if rows.size:
        g = group[rows]
        # trips in order of appearance
        first_of_trip = np.concatenate(([True], g[1:] != g[:-1]))
        trip_index = np.cumsum(first_of_trip) - 1
        n_trips = int(trip_index[-1]) + 1
        trip_first_row = rows[first_of_trip]
        last_of_trip = np.concatenate((g[1:] != g[:-1], [True]))
        trip_last_row = rows[last_of_trip]
        trip_cons = _kahan_group_sum(trip_index, s.consumption[rows], n_trips, first_of_trip)
        trip_id = ids[trip_first_row]
        dep_date = s.date[trip_first_row]
        ret_date = s.date[trip_last_row] + np.timedelta64(minutes * 60, "s")
        date_i = s.date.astype(np.int64)
        for c in range(N):
            sl = slice(c * T, (c + 1) * T)
            mine = trip_id == c
            if not np.any(mine):
                continue
            d_c = date_i[sl]
            # backward as-of on return dates (data_processing.py:176-181): latest return <= date
            r_dates = ret_date[mine].astype(np.int64)
            order = np.argsort(r_dates, kind="stable")
            r_sorted = r_dates[order]
            k = np.searchsorted(r_sorted, d_c, side="right") - 1
            vals = trip_cons[mine][order]
            cons[sl] = np.where(k >= 0, vals[np.clip(k, 0, None)], 0.0)
            # forward as-of on departure dates (:196-201): earliest departure >= date
            dd = dep_date[mine].astype(np.int64)
            dd.sort()
            j = np.searchsorted(dd, d_c, side="left")
            ok = j < dd.size
            nxt = dd[np.clip(j, 0, dd.size - 1)]
            tl[sl] = np.where(ok, (nxt - d_c) / 3600.0, 0.0)
    cons[~there] = 0.0  # :187
    tl[~there] = 0.0  # :206
    sor = target_soc - cons / init_battery_cap  # :221
    sor[~there] = 0.0  # :223

    shp = (N, T)
    return (
        dates,
        there.reshape(shp).T.astype(np.uint8).copy(),
        tl.reshape(shp).T.copy(),
        sor.reshape(shp).T.copy(),
        cons.reshape(shp).T.copy(),
    )


def _kahan_group_sum(label: np.ndarray, values: np.ndarray, n_groups: int, first_of_group: np.ndarray) -> np.ndarray:
    """Per-group sums with Kahan compensation in row order -- what pandas' `groupby().sum()` computes
    for float64 (its group_sum kernel keeps a compensation term per group), which is what
    data_processing.py:144-145 calls.  Needed for bit-identical `SOC_on_return`.  `label` must be
    non-decreasing (rows of a group are contiguous); vectorised over groups, looping over the position
    inside the group."""
    start = np.nonzero(first_of_group)[0]
    length = np.diff(np.concatenate((start, [label.size])))
    total = np.zeros(n_groups)
    comp = np.zeros(n_groups)
    for j in range(int(length.max()) if length.size else 0):
        act = np.nonzero(length > j)[0]
        v = values[start[act] + j]
        y = v - comp[act]
        t = total[act] + y
        c = t - total[act] - y
        c[np.isnan(c)] = 0.0
        comp[act] = c
        total[act] = t
    return total


# --------------------------------------------------------------------------------------
# hourly series -> model grid, reward curves
# --------------------------------------------------------------------------------------
def _shift_years(dates: np.ndarray, years: int) -> np.ndarray:
    """`df["date"] + pd.DateOffset(years=k)` (data_processing.py:424-426): same month/day/time in another
    year; Feb 29 maps to Feb 28 when the target year has none (pandas DateOffset semantics)."""
    if years == 0:
        return dates
    import pandas as pd

    return _to_s((pd.DatetimeIndex(dates) + pd.DateOffset(years=years)).values)


def upsample_backward(grid: np.ndarray, src_dates: np.ndarray, src_vals: np.ndarray) -> np.ndarray:
    """`pd.merge_asof(date_range, df.sort_values("date"), on="date", direction="backward")`
    (data_processing.py:288-292, 313-318, 338-342, 365-368): value of the latest source row with
    date <= grid date (last one among equal dates); NaN before the first source row.
    Applies `_date_checker` (:419-431): if the start years differ the source is shifted by whole years."""
    grid = _to_s(grid)
    src_dates = _to_s(src_dates)
    gy = int(str(grid[0].astype("datetime64[Y]")))
    sy = int(str(src_dates[0].astype("datetime64[Y]")))
    if gy != sy:
        src_dates = _shift_years(src_dates, gy - sy)
    if src_dates[0] != grid[0]:
        raise AssertionError("Invalid start time.")
    if int(str(src_dates[-1].astype("datetime64[Y]"))) != int(str(grid[-1].astype("datetime64[Y]"))):
        raise AssertionError("Invalid end year.")
    order = np.argsort(src_dates, kind="stable")
    sd = src_dates[order].astype(np.int64)
    sv = np.asarray(src_vals, dtype=np.float64)[order]
    k = np.searchsorted(sd, grid.astype(np.int64), side="right") - 1
    out = np.where(k >= 0, sv[np.clip(k, 0, None)], np.nan)
    return out


def detrend_monthly(values: np.ndarray, dates: np.ndarray) -> np.ndarray:
    """`shape_price_reward` (data_processing.py:373-416): shift every calendar month so that its mean
    equals the whole-series mean: group - group.mean() + total_mean.  NaNs are dropped first, like
    `db["DELU"].dropna()`; the result is laid back on the first len(valid) rows (the reference's
    `reset_index` + concat does exactly that)."""
    v = np.asarray(values, dtype=np.float64)
    ok = ~np.isnan(v)
    vv = v[ok]
    dd = _to_s(dates)[: vv.size]  # reference re-indexes the dropna'ed series with car-0 dates
    total = _pd_mean(vv)
    mon = dd.astype("datetime64[M]").astype(np.int64)
    out = np.full(v.shape, np.nan)
    res = np.empty_like(vv)
    bounds = np.nonzero(np.concatenate(([True], mon[1:] != mon[:-1], [True])))[0]
    for a, b in zip(bounds[:-1], bounds[1:]):
        chunk = vv[a:b]
        res[a:b] = chunk - _pd_mean(chunk) + total
    out[: vv.size] = res
    return out


def _pd_mean(x: np.ndarray) -> float:
    """pandas' Series.mean for float64 without NaNs = sum / count; pandas delegates the sum to
    bottleneck/numpy pairwise summation depending on the build, so go through pandas when present
    to stay bit-identical with the reference, else NumPy."""
    try:
        import pandas as pd

        return float(pd.Series(x).mean())
    except Exception:  # pragma: no cover - pandas is present in every supported image
        return float(np.mean(x))


# --------------------------------------------------------------------------------------
# top level
# --------------------------------------------------------------------------------------
def _read_hourly(path: str, column: str, sep: str, decimal: str):
    import pandas as pd

    df = pd.read_csv(path, delimiter=sep, decimal=decimal, parse_dates=["date"])
    col = df[column]
    return _to_s(df["date"].values), col.astype(float).values


_SPOT_COLUMN = "Deutschland/Luxemburg [€/MWh] Original resolutions"  # renamed to DELU at data_processing.py:279


def build_tables(
    schedule: Schedule,
    *,
    minutes: int,
    target_soc: float,
    target_soc_lunch: float,
    init_battery_cap: float,
    is_caretaker: bool,
    spot: tuple[np.ndarray, np.ndarray],
    tariff: tuple[np.ndarray, np.ndarray],
    load: tuple[np.ndarray, np.ndarray] | None,
    pv: tuple[np.ndarray, np.ndarray] | None,
    fixed_markup: float,
    variable_multiplier: float,
    feed_in_deduction: float,
    real_time: bool = False,
) -> FleetTables:
    """Assemble every table from parsed inputs.  `spot`/`tariff`/`load`/`pv` are (dates, values) pairs.
    `real_time`: the schedule is used as it is, without resampling to the model frequency, and its own dates are the
    grid (data_processing.py:54-62, 73-82) -- the rows may then be irregularly spaced."""
    if not real_time:
        schedule = _resample_regular(schedule, minutes)
    dates, there, tl, sor, cons = schedule_tables(schedule, minutes, target_soc, init_battery_cap)
    T = dates.size
    if real_time:
        grid = dates  # date_range = schedule["date"].unique() (:81-82)
    else:
        # model grid: date_range(min, max, freq) (data_processing.py:76-79) == the per-car dates for regular data
        grid = dates[0] + np.arange(T) * np.timedelta64(minutes * 60, "s")
        if not np.array_equal(grid, dates):
            raise ValueError("schedule dates are not a regular grid")

    delu = upsample_backward(grid, *spot)
    trf = upsample_backward(grid, *tariff)
    ld = upsample_backward(grid, *load) if load is not None else np.zeros(T)
    pvv = upsample_backward(grid, *pv) if pv is not None else np.zeros(T)

    hours = ((grid - grid.astype("datetime64[D]")) // np.timedelta64(3600, "s")).astype(np.int64)
    minute = (((grid - grid.astype("datetime64[h]")) // np.timedelta64(60, "s"))).astype(np.int64)
    month = (grid.astype("datetime64[M]").astype(np.int64) % 12) + 1
    weekday = ((grid.astype("datetime64[D]").astype(np.int64) + 3) % 7)  # 1970-01-01 was a Thursday (=3)

    if is_caretaker:
        # fleet_environment.py:951-967 -- rows with hour in [0,10] or [15,23] use target_soc_lunch
        m = ((hours >= 0) & (hours <= 10)) | ((hours >= 15) & (hours <= 23))
        sor[m, :] = target_soc_lunch - cons[m, :] / init_battery_cap
        sor[there == 0] = 0.0

    # reward curves (shape_price_reward): (DELU + markup) * mult, tariff * (1 - fee), month-detrended
    prc = detrend_monthly((delu + fixed_markup) * variable_multiplier, grid)
    trc = detrend_monthly(trf * (1 - feed_in_deduction), grid)

    return FleetTables(
        dates=grid,
        there=there,
        time_left=tl.astype(np.float32),
        soc_on_return=sor,
        consumption=cons,
        delu=delu,
        tariff=trf,
        prc=prc,
        trc=trc,
        load=ld,
        pv=pvv,
        hour=hours.astype(np.uint8),
        minute=minute.astype(np.uint8),
        month=month.astype(np.uint8),
        weekday=weekday.astype(np.uint8),
        minutes_per_step=minutes,
        meta={"time_left_exact": bool(np.array_equal(tl.astype(np.float32).astype(np.float64), tl))},
    )


def build_tables_from_config(cfg: dict, schedule: Schedule | None = None) -> FleetTables:
    """CSV front end driven by the reference's config dict (same keys as /root/reference/config.json).
    `schedule` may be passed pre-built (e.g. from `stack_single_ev_schedules`)."""
    from .config import resolve_config

    rc = resolve_config(cfg)
    path = cfg["data_path"]
    if schedule is None:
        name = cfg["schedule_name"]
        if cfg.get("gen_schedule"):
            # `FleetEnv.auto_gen` (fleet_environment.py:180-182, 969-992): generate `gen_n_evs` vehicles, save them as
            # <data_path>/<gen_name>.csv and use that file -- here with the vectorised generator, in milliseconds
            from .schedule_gen import generate_from_config

            name = generate_from_config(cfg)
        schedule = load_schedule_csv(os.path.join(path, name))
    spot = _read_hourly(os.path.join(path, cfg["price_name"]), _SPOT_COLUMN, ";", ",")
    tariff = _read_hourly(os.path.join(path, cfg["tariff_name"]), "tariff", ";", ",")
    load = pv = None
    if cfg["include_building"]:
        load = _read_hourly(os.path.join(path, cfg["building_name"]), "load", ",", ".")
    if cfg["include_pv"]:
        pv_name = cfg["pv_name"] if cfg.get("pv_name") is not None else cfg["building_name"]
        # the reference parses this file with decimal="," and then `.astype(float)` (data_processing.py:358-361):
        # the column stays text and goes through Python's correctly rounded float(); keep that path bit-identical
        pv = _read_hourly(os.path.join(path, pv_name), "pv", ",", ",")
    return build_tables(
        schedule,
        minutes=rc.minutes,
        target_soc=rc.target_soc,
        target_soc_lunch=rc.target_soc_lunch,
        init_battery_cap=rc.init_battery_cap,
        is_caretaker=rc.is_caretaker,
        spot=spot,
        tariff=tariff,
        load=load,
        pv=pv,
        fixed_markup=rc.fixed_markup,
        variable_multiplier=rc.variable_multiplier,
        feed_in_deduction=rc.feed_in_deduction,
        real_time=rc.real_time,
    )
