"""Synthetic year-long inputs for scale runs (bench.py, large-batch parity tests).

The reference ships only single-EV schedules (the multi-EV blobs `inputs/2_*.csv` are missing)
and its own generator is an O(T^2) pandas loop that takes hours (`ScheduleGenerator`,
/root/reference/fleetrl/utils/schedule/schedule_generator.py:64-691).  This module draws N-EV schedules directly
in array form, O(T*N), with the same per-use-case statistics (`ScheduleConfig`, schedule_config.py:26-132:
departure/return time means and deviations, weekday/weekend split, caretaker lunch pause, distance and
consumption distributions with their clips) and an independent random stream per vehicle (the reference seeds
every vehicle identically, schedule_generator.py:31-32, which makes all vehicles the same -- deliberately not
replicated).  Prices, building load and PV are hourly synthetic series with the magnitude and daily/seasonal shape
of the shipped German 2020 data.  Everything is seeded, nothing is read from disk.
"""
from __future__ import annotations

import numpy as np

from .prestage import FleetTables, Schedule, build_tables

__all__ = ["synth_schedule", "synth_hourly", "synth_tables", "USE_CASES"]

# (dep mean/std wd, ret mean/std wd, dep mean/std we, ret mean/std we, min_dep, max_dep, min_ret, max_ret,
#  dist mean/std wd, dist mean/std we, min/max dist, cons mean/std/min/max kWh/km, trip clip kWh, charger kW)
USE_CASES = {
    "lmd": dict(dep_wd=(7, 1), ret_wd=(19, 1), dep_we=(9, 1.5), ret_we=(17, 1.5), dep_lim=(3, 11), ret_lim=(12, 23),
                dist_wd=(150, 25), dist_we=(75, 25), dist_lim=(20, 280), cons=(0.213, 0.167463672468669, 0.0994, 0.453),
                clip=50.0, power=11.0, sunday=0.0, saturday=1.0),
    "ut": dict(dep_wd=(7, 1), ret_wd=(19, 1), dep_we=(9, 2), ret_we=(16, 2), dep_lim=(3, 11), ret_lim=(12, 23),
               dist_wd=(120, 30), dist_we=(80, 25), dist_lim=(20, 220), cons=(0.224, 0.167463672468669, 0.0994, 0.453),
               clip=41.0, power=22.0, sunday=0.05, saturday=1.0),
    "ct": dict(dep_wd=(6, 1), ret_wd=(19, 1), dep_we=(9, 1.5), ret_we=(15, 1.5), dep_lim=(3, 10), ret_lim=(15, 23),
               pause_beg=(12, 0.25), pause_end_wd=(13.5, 0.25), pause_end_we=(13, 0.25),
               dist_wd=(30, 10), dist_we=(15, 15), dist_lim=(5, 50), cons=(0.17, 0.167463672468669, 0.0994, 0.453),
               clip=13.5, clip_afternoon=10.0, power=4.7, sunday=1.0, saturday=1.0),
}


def _clip_normal(rng, mean_std, lo, hi, size):
    return np.clip(rng.normal(mean_std[0], mean_std[1], size=size), lo, hi)


def synth_schedule(use_case: str, n_evs: int, seed: int = 1234, year: int = 2020, days: int = 365,
                   minutes: int = 15) -> Schedule:
    """N-EV schedule on a regular grid starting `year`-01-01 00:00 (same span as the shipped files: 365 days)."""
    uc = USE_CASES[use_case]
    sph = 60 // minutes
    spd = 24 * sph
    T = days * spd
    start = np.datetime64(f"{year}-01-01T00:00:00", "s")
    dates = start + np.arange(T) * np.timedelta64(minutes * 60, "s")
    weekday0 = int((start.astype("datetime64[D]").astype(np.int64) + 3) % 7)
    dow = (weekday0 + np.arange(days)) % 7  # Monday = 0

    driving = np.zeros((n_evs, T), dtype=bool)
    cons = np.zeros((n_evs, T))
    for ev in range(n_evs):
        rng = np.random.default_rng([seed, ev])
        we = dow >= 5
        active = np.where(dow == 6, rng.random(days) < uc["sunday"], np.where(dow == 5, rng.random(days) < uc["saturday"], True))
        dep = np.where(we, _clip_normal(rng, uc["dep_we"], *uc["dep_lim"], days), _clip_normal(rng, uc["dep_wd"], *uc["dep_lim"], days))
        ret = np.where(we, _clip_normal(rng, uc["ret_we"], *uc["ret_lim"], days), _clip_normal(rng, uc["ret_wd"], *uc["ret_lim"], days))
        dist = np.where(we, _clip_normal(rng, uc["dist_we"], *uc["dist_lim"], days), _clip_normal(rng, uc["dist_wd"], *uc["dist_lim"], days))
        c_km = np.clip(rng.normal(uc["cons"][0], uc["cons"][1], size=days), uc["cons"][2], uc["cons"][3])
        legs = []  # (start slot, end slot (exclusive), kWh) per day
        if use_case == "ct":
            pb = _clip_normal(rng, uc["pause_beg"], 11.0, 12.75, days)
            pe = np.where(we, _clip_normal(rng, uc["pause_end_we"], 12.5, 14.0, days), _clip_normal(rng, uc["pause_end_wd"], 13.0, 14.5, days))
            pe = np.maximum(pe, pb + 0.5)
            ret = np.maximum(ret, pe + 1.0)
            share = rng.uniform(0.4, 0.6, size=days)
            legs.append((dep, pb, np.minimum(dist * share * c_km, uc["clip"])))
            legs.append((pe, ret, np.minimum(dist * (1 - share) * c_km, uc["clip_afternoon"])))
        else:
            ret = np.maximum(ret, dep + 1.0)
            legs.append((dep, ret, np.minimum(dist * c_km, uc["clip"])))
        for a, b, kwh in legs:
            sa = np.floor(a * sph).astype(np.int64)
            sb = np.maximum(np.ceil(b * sph).astype(np.int64), sa + 1)
            sb = np.minimum(sb, spd - 1)  # always home again before midnight
            days_on = np.nonzero(active)[0]
            r0 = days_on * spd + sa[days_on]
            r1 = days_on * spd + sb[days_on]
            # paint the [r0, r1) runs with a difference array (runs of one leg never overlap)
            per_slot = kwh[days_on] / (r1 - r0)
            diff = np.zeros(T + 1)
            np.add.at(diff, r0, per_slot)
            np.add.at(diff, r1, -per_slot)
            flag = np.zeros(T + 1, dtype=np.int64)
            np.add.at(flag, r0, 1)
            np.add.at(flag, r1, -1)
            on = np.cumsum(flag[:T]) > 0
            # exact per-slot value (a running float sum would accumulate rounding): index of the run each slot belongs to
            run_id = np.cumsum(np.isin(np.arange(T), r0)) - 1
            cons[ev, on] += per_slot[run_id[on]]
            driving[ev, on] = True
    power = np.where(driving, 0.0, uc["power"])
    return Schedule(
        date=np.tile(dates, n_evs),
        ev_id=np.repeat(np.arange(n_evs, dtype=np.int64), T),
        consumption=cons.reshape(-1),
        power_rating=power.reshape(-1),
        station_none=driving.reshape(-1),
        station_code=driving.reshape(-1).astype(np.int64),
    )


# level / spread of the two price years the reference ships (inputs/spot_2020_new.csv: mean 30.4, std 17.5 EUR/MWh;
# inputs/spot_2021_new.csv: mean 96.8, std 73.7 with a steep rise over the year) and its fixed feed-in tariff
# (inputs/fixed_feed_in.csv: 60.2 EUR/MWh all year)
PRICE_YEARS = {"2020": dict(mean=30.0, scale=1.0, trend=0.0), "2021": dict(mean=50.0, scale=2.6, trend=95.0)}
FIXED_FEED_IN = 60.2


def synth_hourly(use_case: str, n_evs: int, seed: int = 1234, year: int = 2020, days: int = 366, price_year: str = "2020"):
    """(dates, spot EUR/MWh, load kW, pv kW) hourly.  Spot: AR(1) noise around a daily double-peak shape, occasional
    negative hours; `price_year` "2021" gives the level, spread and rise over the year of the reference's spot_2021 file.
    Load: office-like weekday plateau scaled to the fleet size.  PV: clear-sky bell with a seasonal amplitude and
    day-to-day cloud factor."""
    rng = np.random.default_rng([seed, 10_000])
    H = days * 24
    dates = np.datetime64(f"{year}-01-01T00:00:00", "s") + np.arange(H) * np.timedelta64(3600, "s")
    hod = np.arange(H) % 24
    doy = np.arange(H) // 24
    dow = (int((dates[0].astype("datetime64[D]").astype(np.int64) + 3) % 7) + doy) % 7
    ar = np.zeros(H)
    eps = rng.normal(0, 6.0, size=H)
    for i in range(1, H):
        ar[i] = 0.92 * ar[i - 1] + eps[i]
    shape = 8 * np.exp(-0.5 * ((hod - 8) / 2.0) ** 2) + 12 * np.exp(-0.5 * ((hod - 19) / 2.5) ** 2) - 6 * np.exp(-0.5 * ((hod - 3) / 2.5) ** 2)
    py = PRICE_YEARS[price_year]
    spot = np.round(py["mean"] + py["trend"] * (doy / 366.0) ** 2 + py["scale"] * (shape + ar - 4 * (dow >= 5)), 2)
    per_ev = {"lmd": 4.0, "ut": 7.0, "ct": 2.5}[use_case]
    base = 5.0 + per_ev * n_evs
    occ = np.where(dow < 5, 1.0, 0.45) * (0.35 + 0.65 * np.exp(-0.5 * ((hod - 12.5) / 4.0) ** 2))
    load = np.round(base * (0.3 + occ) * (1 + 0.05 * rng.normal(size=H)).clip(0.8, 1.2), 6)
    season = 0.35 + 0.65 * np.sin(np.pi * (doy % 366) / 366) ** 2
    cloud = np.repeat(rng.uniform(0.25, 1.0, size=days), 24)
    pv = np.clip(np.cos((hod - 12.5) / 24 * 2 * np.pi), 0, None) ** 1.5 * season * cloud * (0.6 * base)
    return dates, spot, load, np.round(pv, 6)


def synth_tables(use_case: str, n_evs: int, *, seed: int = 1234, target_soc: float = 0.85, target_soc_lunch: float = 0.65,
                 fixed_markup: float = 10, variable_multiplier: float = 1.5, feed_in_deduction: float = 0.25,
                 include_building: bool = True, include_pv: bool = True, minutes: int = 15, price_year: str = "2020",
                 feed_in: str = "spot") -> FleetTables:
    """`price_year`: "2020" | "2021" (see PRICE_YEARS); `feed_in`: "spot" (the tariff follows the spot price, like the
    reference's spot_*_tariff files) or "fixed" (a constant feed-in tariff, like inputs/fixed_feed_in.csv)."""
    init_cap = {"lmd": 60.0, "ut": 50.0, "ct": 16.7}[use_case]
    sched = synth_schedule(use_case, n_evs, seed=seed, minutes=minutes)
    dates, spot, load, pv = synth_hourly(use_case, n_evs, seed=seed, price_year=price_year)
    tariff = spot if feed_in == "spot" else np.full_like(spot, FIXED_FEED_IN)
    return build_tables(
        sched, minutes=minutes, target_soc=target_soc, target_soc_lunch=target_soc_lunch, init_battery_cap=init_cap,
        is_caretaker=(use_case == "ct"), spot=(dates, spot), tariff=(dates, tariff),
        load=(dates, load) if include_building else None, pv=(dates, pv) if include_pv else None,
        fixed_markup=fixed_markup, variable_multiplier=variable_multiplier, feed_in_deduction=feed_in_deduction,
    )
