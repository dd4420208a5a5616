"""FleetParams assembly: resolved config + pre-staged tables -> the scalar block of the C ABI.

Covers the parts of `FleetEnv.__init__` that need the data
(/root/reference/fleetrl/fleet_env/fleet_environment.py:260-311): `num_cars`, `max_load`,
`LoadCalculation` sizing, `OracleNormalization.__init__` constants
(utils/normalization/oracle_normalization.py:34-54), plus the start-row ranges of the three time
pickers (utils/time_picker/*.py) expressed as table-row indices.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .config import DEG_RAINFLOW, ResolvedConfig
from .prestage import FleetTables

__all__ = ["make_params", "table_extrema", "time_features", "obs_dim", "static_start_row", "picker_range", "validate_supported"]


def _pymax(x: np.ndarray) -> float:
    """Python's builtin `max(series)` as the reference uses it (oracle_normalization.py:34-46,
    fleet_environment.py:266) on a column that may hold NaN after row 0: NaN never wins a `>` test, so the
    result is the max over the non-NaN values provided the first value is not NaN."""
    return float(np.nanmax(x))


def _pymin(x: np.ndarray) -> float:
    return float(np.nanmin(x))


def obs_dim(rc: ResolvedConfig, num_cars: int) -> int:
    """`detect_dim_and_bounds` (fleet_environment.py:854-949)."""
    n, L, B = num_cars, rc.price_lookahead, rc.bl_pv_lookahead
    dim = 2 * n + (L + 1) * 2
    if rc.include_building and rc.include_pv:
        dim += 2 * (B + 1)
    elif rc.include_building or rc.include_pv:
        dim += B + 1
    if rc.aux:
        dim += 5 * n + 1 + 6
        if rc.include_building:
            dim += 3
    return dim


def validate_supported(rc: ResolvedConfig) -> None:
    """Supported matrix = what the reference itself can run; everything else is
    rejected with a clear error instead of the reference's KeyError/ValueError deep inside step()."""
    if not rc.include_price:
        raise ValueError("include_price=False is not runnable on the reference either (KeyError 'price_reward_curve', "
                         "ev_charger.py:155); unsupported")
    if rc.normalize_in_env and rc.include_pv and not rc.include_building:
        raise ValueError("normalize_in_env with include_pv and without include_building crashes in the reference "
                         "(oracle_normalization.py:121); unsupported")
    if rc.deg_mode == DEG_RAINFLOW and rc.init_soh != 1.0:
        raise ValueError("rainflow/SEI degradation with init_soh != 1.0 is ill-defined in the reference "
                         "(rainflow_sei_degradation.py:184); unsupported")
    if 60 % rc.minutes:
        raise ValueError("minutes per step must divide 60")


def irregular_grid_tables(tables: FleetTables, rc: ResolvedConfig) -> dict:
    """What the step needs on an irregular time grid (real_time only), as row-indexed tables:
    dt_row      hours to the next row -- `get_next_dt` (fleet_environment.py:994-1008): (date[t+1] - date[t]).total_seconds()/3600
    finish_row  the row whose date equals date[t] + episode_length hours (`finish_time`, :355; the done test is an exact
                date comparison, :627), -1 if there is none
    lookahead_row[t, k-1]  first row of clock hour floor_hour(date[t]) + k (`resample("H", on="date").first()` of the slice
                starting at t, observer_bl_pv.py:53-79), -1 past the end of the table
    second      seconds of the clock time (EventManager.check_event wants minute == 15 and second == 0)."""
    d = tables.dates.astype("datetime64[s]").astype(np.int64)
    T = d.size
    dt = np.empty(T)
    dt[:-1] = (d[1:] - d[:-1]) / 3600.0
    dt[-1] = dt[-2] if T > 1 else rc.dt
    want = d + rc.episode_length * 3600
    k = np.searchsorted(d, want)
    finish = np.where((k < T) & (d[np.clip(k, 0, T - 1)] == want), k, -1).astype(np.int32)
    cols = max(rc.price_lookahead, rc.bl_pv_lookahead, 1)
    hour0 = (d // 3600) * 3600
    look = np.empty((T, cols), dtype=np.int32)
    for j in range(1, cols + 1):
        r = np.searchsorted(d, hour0 + j * 3600)  # first row at or after the top of that hour ...
        ok = (r < T) & (d[np.clip(r, 0, T - 1)] < hour0 + (j + 1) * 3600)  # ... that still lies inside it
        look[:, j - 1] = np.where(ok, r, -1)
    # rows the reference's pickers can draw: pd.date_range(min_date, ..., freq) -- the model-frequency grid anchored at the
    # first date (random_time_picker.py:25-28, eval_time_picker.py:33-36); off-grid rows are never start rows
    on_grid = np.nonzero((d - d[0]) % (rc.minutes * 60) == 0)[0].astype(np.int32)
    return dict(dt_row=dt, finish_row=finish, lookahead_row=look, second=(d % 60).astype(np.uint8), pick_rows=on_grid)


def time_features(tables: FleetTables) -> np.ndarray:
    """[T,6] float32: month/week/hour sin,cos exactly as the observers compute them
    (observer_bl_pv.py:100-107: `np.sin(2 * np.pi * time.month/12)` ... on Python scalars, then the
    float32 cast of `np.array(..., dtype=np.float32)`).  Evaluated per distinct value with the same NumPy
    scalar expression so the float32 words are identical to the reference's on the same NumPy."""
    T = tables.T
    out = np.empty((T, 6), dtype=np.float32)
    lut_m = {m: (np.sin(2 * np.pi * m / 12), np.cos(2 * np.pi * m / 12)) for m in range(1, 13)}
    lut_w = {w: (np.sin(2 * np.pi * w / 7), np.cos(2 * np.pi * w / 7)) for w in range(7)}
    lut_h = {h: (np.sin(2 * np.pi * h / 24), np.cos(2 * np.pi * h / 24)) for h in range(24)}
    m = np.array([lut_m[int(v)] for v in range(1, 13)])
    w = np.array([lut_w[int(v)] for v in range(7)])
    h = np.array([lut_h[int(v)] for v in range(24)])
    out[:, 0:2] = m[tables.month.astype(np.int64) - 1]
    out[:, 2:4] = w[tables.weekday.astype(np.int64)]
    out[:, 4:6] = h[tables.hour.astype(np.int64)]
    return out


def static_start_row(tables: FleetTables, start_time: str = "01/02/2021 19:00") -> int:
    """`StaticTimePicker` (utils/time_picker/static_time_picker.py:12-32): "01/02/2021 19:00" parsed
    month-first, re-based to the first year of the table when its year is outside the table's range."""
    import pandas as pd

    ts = pd.to_datetime(start_time)
    first_year = int(str(tables.dates[0].astype("datetime64[Y]")))
    last_year = int(str(tables.dates[-1].astype("datetime64[Y]")))
    if ts.year < first_year or ts.year > last_year:
        ts = ts + pd.DateOffset(years=first_year - ts.year)
    row = int(np.searchsorted(tables.dates, np.datetime64(ts, "s")))
    if row >= tables.T or tables.dates[row] != np.datetime64(ts, "s"):
        raise ValueError(f"static start time {ts} is not a table row")
    return row


def picker_range(rc: ResolvedConfig, tables: FleetTables) -> tuple[int, int]:
    """Inclusive start-row range.
    random: `pd.date_range(min, max - end_cutoff days, freq)`  (random_time_picker.py:25-28)
    eval  : `pd.date_range(max - end_cutoff days, max - 2*episode_length hours, freq)` (eval_time_picker.py:33-36)
    static: the single row of `StaticTimePicker`."""
    irr = tables.meta.get("irregular")
    if irr is not None:
        # irregular grid: the same date arithmetic, as INDICES into the candidate list `pick_rows` (on-grid rows)
        d = tables.dates.astype("datetime64[s]").astype(np.int64)
        cand = d[irr["pick_rows"]]

        def last_at_or_before(ts):
            return int(np.searchsorted(cand, ts, side="right") - 1)

        if rc.time_picker == "static":
            k = int(np.searchsorted(irr["pick_rows"], static_start_row(tables)))
            return k, k
        cut = last_at_or_before(d[-1] - rc.end_cutoff * 86400)
        if rc.time_picker == "random":
            return 0, cut
        first_eval = int(np.searchsorted(cand, d[-1] - rc.end_cutoff * 86400, side="left"))
        return first_eval, last_at_or_before(d[-1] - 2 * rc.episode_length * 3600)
    T = tables.T
    sph = 60 // rc.minutes
    if rc.time_picker == "static":
        r = static_start_row(tables)
        return r, r
    cut = T - 1 - rc.end_cutoff * 24 * sph
    if rc.time_picker == "random":
        return 0, cut
    return cut, T - 1 - 2 * rc.episode_length * sph


def table_extrema(tables: FleetTables) -> dict:
    """Whole-table maxima/minima used for grid sizing (fleet_environment.py:265-268) and by
    `OracleNormalization.__init__` (oracle_normalization.py:34-46)."""
    return dict(
        max_time_left=_pymax(tables.time_left.astype(np.float64)),
        max_delu=_pymax(tables.delu), min_delu=_pymin(tables.delu),
        max_tariff=_pymax(tables.tariff), min_tariff=_pymin(tables.tariff),
        max_load=_pymax(tables.load), max_pv=_pymax(tables.pv),
    )


def make_params(rc: ResolvedConfig, tables: FleetTables, num_envs: int, *, auto_reset: bool = True,
                env_id_offset: int = 0, seed: int | None = None, extrema: dict | None = None,
                start_range: tuple[int, int] | None = None) -> _capi.FleetParams:
    """`extrema` overrides `table_extrema(tables)` (needed when `tables` is a cut window of a longer table);
    `start_range` overrides the picker's inclusive start-row range."""
    validate_supported(rc)
    n = tables.N
    sph = 60 // rc.minutes
    step = np.diff(tables.dates.astype("datetime64[s]").astype(np.int64))
    if step.size and not np.all(step == rc.minutes * 60):
        if not rc.real_time:
            raise ValueError(f"table rows must be {rc.minutes} min apart unless real_time=True (the reference resamples "
                             "the schedule to the model frequency otherwise, data_processing.py:54-62)")
        if not (rc.include_building and rc.include_pv):
            raise ValueError("an irregular time grid only runs with include_building and include_pv in the reference (the "
                             "other observers look the window end up by exact date, observer_price_only.py:51); unsupported")
        tables.meta["irregular"] = irregular_grid_tables(tables, rc)  # travels to the library through _capi.pack_tables
    else:
        tables.meta.pop("irregular", None)
    ext = table_extrema(tables) if extrema is None else extrema
    max_load = ext["max_load"] if rc.include_building else 0  # fleet_environment.py:265-268
    grid, evse, batt = rc.company(n, max_load)
    lo, hi = picker_range(rc, tables) if start_range is None else start_range
    p = _capi.FleetParams()
    p.abi_version = _capi.ABI_VERSION
    p.struct_bytes = C.sizeof(_capi.FleetParams)
    p.num_envs = int(num_envs)
    p.num_cars = n
    p.table_rows = tables.T
    p.episode_steps = rc.episode_length * sph
    p.price_lookahead = rc.price_lookahead
    p.bl_pv_lookahead = rc.bl_pv_lookahead
    p.steps_per_hour = sph
    p.hour_phase = int(tables.minute[0]) // rc.minutes
    p.include_building = int(rc.include_building)
    p.include_pv = int(rc.include_pv)
    p.aux = int(rc.aux)
    p.normalize = int(rc.normalize_in_env)
    p.is_caretaker = int(rc.is_caretaker)
    p.deg_mode = rc.deg_mode
    p.picker_mode = rc.picker_mode
    p.start_lo, p.start_hi = lo, hi
    p.auto_reset = int(auto_reset)
    p.env_id_offset = int(env_id_offset)
    p.log_data = int(bool(rc.raw.get("log_data", False)))
    p.log_capacity = int(rc.raw.get("log_capacity", 0) or 0)  # extension key: rows per env of the device-side log ring (0 = default)
    p.real_time = int(rc.real_time)
    s = rc.seed if seed is None else seed
    p.seed = int(s) if s is not None else 0
    p.dt = rc.dt
    p.evse_power = evse
    p.obc_max_power = rc.obc_max_power
    p.batt_cap_nominal = batt
    p.init_battery_cap = rc.init_battery_cap
    p.grid_connection = grid
    p.charging_eff = rc.charging_eff
    p.discharging_eff = rc.discharging_eff
    p.fixed_markup = rc.fixed_markup
    p.variable_multiplier = rc.variable_multiplier
    p.feed_in_deduction = rc.feed_in_deduction
    p.price_multiplier = rc.price_multiplier
    p.penalty_invalid_action = rc.penalty_invalid_action
    p.penalty_overcharging = rc.penalty_overcharging
    p.clip_overcharging = rc.clip_overcharging
    p.penalty_overloading = rc.penalty_overloading
    p.fully_charged_reward = rc.fully_charged_reward
    p.target_soc = rc.target_soc
    p.target_soc_lunch = rc.target_soc_lunch
    p.eps = rc.eps
    p.def_soc = rc.def_soc
    p.min_laxity = rc.min_laxity
    p.init_soh = rc.init_soh
    p.temperature = rc.temperature
    # OracleNormalization.__init__ (oracle_normalization.py:34-54)
    p.max_time_left = ext["max_time_left"]
    p.max_price = (ext["max_delu"] + rc.fixed_markup) * rc.variable_multiplier
    p.min_price = (ext["min_delu"] + rc.fixed_markup) * rc.variable_multiplier
    p.max_tariff = ext["max_tariff"] * (1 - rc.feed_in_deduction)
    p.min_tariff = ext["min_tariff"] * (1 - rc.feed_in_deduction)
    p.max_building = ext["max_load"] if rc.include_building else 1.0
    p.max_pv = ext["max_pv"] if rc.include_pv else 1.0
    p.max_soc = rc.target_soc
    p.max_hours_needed = (rc.target_soc * rc.init_battery_cap) / (evse * rc.charging_eff)
    p.max_laxity = 5
    p.max_evse = evse
    p.max_grid = grid
    return p
