"""Vectorised schedule generator: what `gen_schedule=True` builds.

Reference: `ScheduleGenerator` (/root/reference/fleetrl/utils/schedule/schedule_generator.py:64-691) with the statistics of
`ScheduleConfig` (schedule_config.py:26-172), called once per vehicle by `FleetEnv.auto_gen`
(fleet_env/fleet_environment.py:969-992).  The reference walks every 15-minute row of the year and assigns six cells per row
through a boolean scan of the frame -- O(T^2) per vehicle, "1-3 hours" by its own docstring.  The sampling RULES are simple,
though, and this module applies them to whole arrays at once (O(T) per vehicle, milliseconds):

  per day        departure / return (caretaker: + lunch pause) times ~ Normal, hour = trunc clipped to the use case's window,
                 minute = the quarter nearest to trunc(frac * 60) (never rounds up into the next hour); day's distance ~ Normal
                 clipped; weekday / Saturday / Sunday rules per use case (delivery: no Sunday; utility: 5 % of the Sundays;
                 caretaker: every day, weekend statistics on both weekend days)
  per trip row   Distance_km = distance / trip rows; consumption rating ~ Normal clipped to [min, max] and to
                 `total_cons_clip / distance`, drawn anew for EVERY row; Consumption_kWh = Distance_km * rating
  caretaker      both half-day trips carry the day's whole distance each (the reference divides the same `total_distance` by
                 either trip's rows, :344 / :362); the afternoon trip has its own energy clip; a 2 % chance per day of an
                 emergency trip 02:00-04:00 (nine rows, the end row inclusive) on that same day (:381-408)
  home rows      Location = ChargingStation = "home", PowerRating_kW = the use case's charger; trip rows "driving" / "none" / 0

Distributionally identical to the reference, not stream-identical: the reference draws from NumPy's global legacy stream in
row order AND re-seeds it with the same seed for every vehicle (:31-32), so all its vehicles are copies of each other; here
every vehicle has its own counter-based Philox stream keyed by (seed, vehicle) -- `identical_vehicles=True` gives the
reference's behaviour.  tests/test_schedule_gen.py checks the output against summary statistics of the reference's own
generator (tests/golden/schedule_stats.json, produced by oracle/gen_schedule_stats.py in the build container).

Output: the reference's CSV schema (an unnamed index column, date, Distance_km, Consumption_kWh, Location, ChargingStation,
ID, PowerRating_kW), which `fleetrl_amd.prestage.load_schedule_csv` reads back, or a `prestage.Schedule` directly.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from .prestage import Schedule

__all__ = ["ScheduleStats", "schedule_stats_for", "generate_schedule", "generate_schedule_arrays", "generate_schedule_frame", "write_schedule_csv",
           "generate_from_config"]


@dataclass
class ScheduleStats:
    """`ScheduleConfig` (schedule_config.py:26-172) as plain numbers; *_we = Saturday (and Sunday for the caretaker)."""

    kind: str  # "delivery" | "caretaker" | "utility" | "custom"
    dep_wd: tuple
    ret_wd: tuple
    dep_we: tuple
    ret_we: tuple
    dep_lim: tuple  # (min_dep, max_dep)
    ret_lim_wd: tuple  # (min_return, max_return_hour)
    ret_lim_we: tuple
    dist_wd: tuple
    dist_we: tuple
    dist_lim: tuple
    cons: tuple  # mean, std, min, max  [kWh/km]
    clip: float  # max kWh of a trip
    power: float  # charger kW
    sunday_prob: float = 0.0  # probability that a Sunday is an operating day (weekend statistics)
    # caretaker only
    pause_beg_wd: tuple = (12, 0.25)
    pause_end_wd: tuple = (13.5, 0.25)
    pause_beg_we: tuple = (12, 0.25)
    pause_end_we: tuple = (13, 0.25)
    clip_afternoon: float = 10.0
    prob_emergency: float = 0.02
    dist_em: tuple = (15, 5)
    min_em_distance: float = 5.0


def schedule_stats_for(use_case: str, env_config: dict | None = None) -> ScheduleStats:
    """Use case ("lmd" | "ct" | "ut" | "custom") -> statistics, schedule_config.py:26-172."""
    c = 0.167463672468669
    if use_case == "lmd":
        return ScheduleStats("delivery", (7, 1), (19, 1), (9, 1.5), (17, 1.5), (3, 11), (12, 23), (12, 23), (150, 25), (75, 25),
                             (20, 280), (0.213, c, 0.0994, 0.453), 50.0, 11.0)
    if use_case == "ut":
        return ScheduleStats("utility", (7, 1), (19, 1), (9, 2), (16, 2), (3, 11), (12, 23), (12, 23), (120, 30), (80, 25),
                             (20, 220), (0.224, c, 0.0994, 0.453), 41.0, 22.0, sunday_prob=0.05)
    if use_case == "ct":
        return ScheduleStats("caretaker", (6, 1), (19, 1), (9, 1.5), (15, 1.5), (3, 10), (15, 23), (15, 23), (30, 10), (15, 15),
                             (5, 50), (0.17, c, 0.0994, 0.453), 13.5, 4.7, sunday_prob=1.0)
    if use_case == "custom":
        g = (env_config or {}).get
        return ScheduleStats(
            "custom", (g("custom_weekday_departure_time_mean", 7), g("custom_weekday_departure_time_std", 1)),
            (g("custom_weekday_return_time_mean", 19), g("custom_weekday_return_time_std", 1)),
            (g("custom_weekend_departure_time_mean", 9), g("custom_weekend_departure_time_std", 1.5)),
            (g("custom_weekend_return_time_mean", 17), g("custom_weekend_return_time_std", 1.5)),
            (g("custom_earliest_hour_of_departure", 3), g("custom_latest_hour_of_departure", 11)),
            (g("custom_earliest_hour_of_return", 12), g("custom_latest_hour_of_return", 23)),
            (g("custom_earliest_hour_of_return", 12), g("custom_latest_hour_of_return", 23)),
            (g("custom_weekday_distance_mean", 300), g("custom_weekday_distance_std", 25)),
            (g("custom_weekend_distance_mean", 150), g("custom_weekend_distance_std", 25)),
            (g("custom_minimum_distance", 20), g("custom_max_distance", 400)),
            (g("custom_consumption_mean", 1.3), g("custom_consumption_std", c), g("custom_minimum_consumption", 0.3994),
             g("custom_maximum_consumption", 2.5)),
            g("custom_maximum_consumption_per_trip", 500), g("custom_ev_charger_power_in_kw", 120))
    raise ValueError(f"unknown use case {use_case!r}")


def _clock(x: np.ndarray, lo: int | None, hi: int | None) -> np.ndarray:
    """Normal sample [hours] -> quarter-hour slot of the day the way the reference rounds it (:99-105): hour = integer part
    (clipped when the use case has a window), minute = the entry of (0, 15, 30, 45) nearest to int(fraction * 60)."""
    hour = np.trunc(x).astype(np.int64)
    if lo is not None:
        hour = np.clip(hour, lo, hi)
    frac_min = np.trunc((x - np.trunc(x)) * 60).astype(np.int64)
    quarter = np.abs(np.array([0, 15, 30, 45])[None, :] - frac_min[:, None]).argmin(axis=1)
    return hour * 4 + quarter


def _vehicle(stats: ScheduleStats, rng: np.random.Generator, dow: np.ndarray, spd: int):
    """One vehicle: per-row (driving, distance_km, consumption_kwh) over len(dow) days of `spd` = 96 rows."""
    days = dow.size
    T = days * spd
    wd = dow < 5
    if stats.kind == "caretaker":
        active = np.ones(days, dtype=bool)
    else:
        sunday_on = rng.random(days) < stats.sunday_prob  # utility: `np.random.random() > 0.95` (:497)
        active = wd | (dow == 5) | ((dow == 6) & sunday_on)
    pick = lambda a, b: np.where(wd, rng.normal(a[0], a[1], days), rng.normal(b[0], b[1], days))  # noqa: E731
    dep = np.where(wd, _clock(rng.normal(*stats.dep_wd, days), *stats.dep_lim), _clock(rng.normal(*stats.dep_we, days), *stats.dep_lim))
    ret = np.where(wd, _clock(rng.normal(*stats.ret_wd, days), *stats.ret_lim_wd), _clock(rng.normal(*stats.ret_we, days), *stats.ret_lim_we))
    dist = np.clip(pick(stats.dist_wd, stats.dist_we), *stats.dist_lim)
    driving = np.zeros(T, dtype=bool)
    distance = np.zeros(T)
    consumption = np.zeros(T)

    def paint(day_idx, s0, s1, total_distance, clip):
        """trip rows [s0, s1) of the given days: equal distance per row, a fresh consumption rating per row (:172-183)"""
        n = s1 - s0
        ok = n > 0
        day_idx, s0, n, total_distance = day_idx[ok], s0[ok], n[ok], total_distance[ok]
        if day_idx.size == 0:
            return
        rows = np.repeat(day_idx * spd + s0, n) + (np.arange(n.sum()) - np.repeat(np.cumsum(n) - n, n))
        per_row = np.repeat(total_distance / n, n)
        rating = np.minimum(np.clip(rng.normal(stats.cons[0], stats.cons[1], rows.size), stats.cons[2], stats.cons[3]),
                            np.repeat(clip / total_distance, n))
        driving[rows] = True
        distance[rows] = per_row
        consumption[rows] = per_row * rating

    on = np.nonzero(active)[0]
    if stats.kind == "caretaker":
        pb = np.where(wd, _clock(rng.normal(*stats.pause_beg_wd, days), None, None), _clock(rng.normal(*stats.pause_beg_we, days), None, None))
        pe = np.where(wd, _clock(rng.normal(*stats.pause_end_wd, days), None, None), _clock(rng.normal(*stats.pause_end_we, days), None, None))
        pe = np.where(pe < pb, pb + 1, pe)  # the pause ends a quarter of an hour after it began at the earliest (:252-255)
        # (a trip whose end lies before its start has no rows; the reference's `step >= a and step < b` is empty then too)
        paint(on, dep[on], pb[on], dist[on], stats.clip)
        paint(on, pe[on], ret[on], dist[on], stats.clip_afternoon)
        em = np.nonzero(rng.random(days) < stats.prob_emergency)[0]  # `np.random.random() > 0.98` at 23:45 (:381-382)
        if em.size:
            d_em = np.maximum(rng.normal(stats.dist_em[0], stats.dist_em[1], em.size), stats.min_em_distance)
            # 02:00 ... 04:00 inclusive = nine rows that each carry distance / 8 (:385-392): overwrites the day's own rows
            n9 = np.full(em.size, 9)
            rows = np.repeat(em * spd + 8, n9) + np.tile(np.arange(9), em.size)
            per_row = np.repeat(d_em / 8.0, n9)
            rating = np.minimum(np.clip(rng.normal(stats.cons[0], stats.cons[1], rows.size), stats.cons[2], stats.cons[3]),
                                np.repeat(stats.clip / d_em, n9))
            driving[rows] = True
            distance[rows] = per_row
            consumption[rows] = per_row * rating
    else:
        paint(on, dep[on], ret[on], dist[on], stats.clip)
    return driving, distance, consumption


def generate_schedule_arrays(use_case: str, n_evs: int, start_date, end_date, *, seed: int = 0, env_config: dict | None = None,
                             identical_vehicles: bool = False, minutes: int = 15):
    """-> (dates datetime64[s] [T], driving bool [n_evs, T], distance_km [n_evs, T], consumption_kwh [n_evs, T], charger kW) for
    the rows `pd.date_range(start, end, freq=15 min)` (end inclusive; a delivery / utility / custom schedule that would start
    on a Sunday starts on the Monday, :74-83)."""
    if minutes != 15:
        raise ValueError("the reference's generator is written for 15-minute rows (trip rows = hours * 4)")
    stats = schedule_stats_for(use_case, env_config)
    t0 = np.datetime64(str(start_date).replace(" ", "T"), "s")
    t1 = np.datetime64(str(end_date).replace(" ", "T"), "s")
    step = np.timedelta64(900, "s")
    dates = t0 + np.arange(int((t1 - t0) // step) + 1) * step
    weekday = lambda d: int((d.astype("datetime64[D]").astype(np.int64) + 3) % 7)  # noqa: E731  Monday = 0
    if stats.kind != "caretaker":
        while dates.size and weekday(dates[0]) == 6:
            dates = dates[1:]
    if dates.size == 0:
        raise ValueError("empty date range")
    spd = 96
    day0 = dates[0].astype("datetime64[D]")
    first_slot = int((dates[0] - day0.astype("datetime64[s]")) // step)
    days = int((dates[-1].astype("datetime64[D]") - day0).astype(np.int64)) + 1
    dow = (weekday(dates[0]) + np.arange(days)) % 7
    T = dates.size
    driving = np.zeros((n_evs, T), dtype=bool)
    distance = np.zeros((n_evs, T))
    consumption = np.zeros((n_evs, T))
    for ev in range(n_evs):
        rng = np.random.Generator(np.random.Philox(key=[int(seed), 0 if identical_vehicles else ev + 1]))
        drv, dist, cons = _vehicle(stats, rng, dow, spd)
        sl = slice(first_slot, first_slot + T)
        driving[ev], distance[ev], consumption[ev] = drv[sl], dist[sl], cons[sl]
    return dates, driving, distance, consumption, float(stats.power)


def generate_schedule_frame(use_case: str, n_evs: int, start_date, end_date, **kw):
    """-> pandas DataFrame in the reference's schema for vehicles 0..n_evs-1 (see generate_schedule_arrays)."""
    import pandas as pd

    dates, driving, distance, consumption, power = generate_schedule_arrays(use_case, n_evs, start_date, end_date, **kw)
    parts = []
    for ev in range(n_evs):
        drv = driving[ev]
        parts.append(pd.DataFrame({
            "date": pd.DatetimeIndex(dates), "Distance_km": distance[ev], "Consumption_kWh": consumption[ev],
            "Location": np.where(drv, "driving", "home"), "ChargingStation": np.where(drv, "none", "home"),
            "ID": ev, "PowerRating_kW": np.where(drv, 0.0, power)}))
    return pd.concat(parts)  # like `pd.concat(gen_sched)` (:986): every vehicle keeps its own 0..T-1 index


def write_schedule_csv(frame, path: str) -> str:
    """`complete_schedule.to_csv(...)` (fleet_environment.py:989): the index becomes the unnamed first column."""
    frame.to_csv(path)
    return path


def generate_schedule(use_case: str, n_evs: int, start_date, end_date, **kw) -> Schedule:
    """The same as a `prestage.Schedule` (what the pre-stager consumes), without the DataFrame / CSV detour."""
    dates, driving, _distance, consumption, power = generate_schedule_arrays(use_case, n_evs, start_date, end_date, **kw)
    T = dates.size
    drv = driving.reshape(-1)
    return Schedule(date=np.tile(dates, n_evs), ev_id=np.repeat(np.arange(n_evs, dtype=np.int64), T),
                    consumption=consumption.reshape(-1), power_rating=np.where(drv, 0.0, power), station_none=drv.copy(),
                    station_code=drv.astype(np.int64))


def generate_from_config(cfg: dict) -> str:
    """`FleetEnv.auto_gen` (fleet_environment.py:969-992): generate `gen_n_evs` vehicles from `gen_start_date` to
    `gen_end_date`, save them as `<data_path>/<gen_name>.csv` and return the file name (the caller uses it as
    `schedule_name`)."""
    name = cfg["gen_name"]
    if not name.endswith(".csv"):
        name = name + ".csv"
    frame = generate_schedule_frame(cfg["use_case"], int(cfg["gen_n_evs"]), cfg["gen_start_date"], cfg["gen_end_date"],
                                    seed=int(cfg.get("seed", 0) or 0), env_config=cfg,
                                    identical_vehicles=bool(cfg.get("gen_identical_vehicles", False)))
    write_schedule_csv(frame, os.path.join(cfg["data_path"], name))
    return name


def summarize_schedule(frame) -> dict:
    """Per use case comparable summary of a schedule frame (any number of vehicles): for weekdays / Saturdays / Sundays the
    share of operating days and mean / std / n of the first trip row, the last trip row, the number of trip rows, the day's
    distance and energy; the per-row consumption rating; for two-trip days (caretaker) the lunch pause; the share of days
    with rows driven between 02:00 and 04:00 (caretaker emergencies).  Used on the reference generator's output
    (oracle/gen_schedule_stats.py -> tests/golden/schedule_stats.json) and on ours (tests/test_schedule_gen.py)."""
    import pandas as pd

    f = frame.copy()
    f["date"] = pd.to_datetime(f["date"])
    f["day"] = f["date"].dt.normalize()
    f["slot"] = f["date"].dt.hour * 4 + f["date"].dt.minute // 15
    f["drv"] = (f["ChargingStation"].astype(str) == "none")
    out = {}

    def ms(x):
        x = np.asarray(x, dtype=np.float64)
        return {"mean": float(x.mean()) if x.size else 0.0, "std": float(x.std()) if x.size else 0.0, "n": int(x.size)}

    day_rows = []
    for (ev, day), g in f.groupby(["ID", "day"], sort=False):
        if len(g) < 96:
            continue  # partial day at the edge of the range
        drv = g["drv"].values
        slot = g["slot"].values
        night = drv & (slot >= 8) & (slot <= 16)
        main = drv & ~((slot >= 8) & (slot <= 16) & (night.sum() == 9) & (slot <= 16))  # emergency rows do not count as the day's trips
        rec = {"dow": int(day.weekday()), "operating": bool(main.any()), "emergency": bool(night.sum() == 9 and not drv[slot == 17].any())}
        if main.any():
            s = slot[main]
            rec.update(first=int(s.min()), last=int(s.max()), rows=int(main.sum()), distance=float(g["Distance_km"].values[main].sum()),
                       energy=float(g["Consumption_kWh"].values[main].sum()))
            gaps = np.nonzero(np.diff(s) > 1)[0]
            if gaps.size == 1:  # two trips: the lunch pause
                rec.update(pause_beg=int(s[gaps[0]] + 1), pause_end=int(s[gaps[0] + 1]))
        day_rows.append(rec)
    d = pd.DataFrame(day_rows)
    for name, sel in (("weekday", d.dow < 5), ("saturday", d.dow == 5), ("sunday", d.dow == 6)):
        k = d[sel]
        op = k[k.operating] if len(k) else k
        o = {"days": int(len(k)), "operating_share": float(k.operating.mean()) if len(k) else 0.0}
        for col in ("first", "last", "rows", "distance", "energy", "pause_beg", "pause_end"):
            if col in op:
                o[col] = ms(op[col].dropna().values)
        out[name] = o
    drv = f[f["drv"] & (f["Distance_km"] > 0)]
    out["rating"] = ms((drv["Consumption_kWh"] / drv["Distance_km"]).values)
    out["emergency_share"] = float(d.emergency.mean()) if len(d) else 0.0
    out["home_power"] = float(f.loc[~f["drv"], "PowerRating_kW"].iloc[0])
    return out
