"""Synthetic year-long inputs for scale runs (bench.py, large-batch parity tests).

The reference ships only single-EV schedules (the multi-EV blobs `inputs/2_*.csv` are missing)
and its own generator is an O(T^2) pandas loop that takes hours (`ScheduleGenerator`,
/root/reference/fleetrl/utils/schedule/schedule_generator.py:64-691).  N-EV schedules come from the package's vectorised
generator (fleetrl_amd/schedule_gen.py: the reference's sampling rules and `ScheduleConfig` statistics, O(T*N), an
independent random stream per vehicle).  Prices, building load and PV are hourly synthetic series with the magnitude and
daily/seasonal shape of the shipped German data.  Everything is seeded, nothing is read from disk.
"""
from __future__ import annotations

import numpy as np

from .prestage import FleetTables, Schedule, build_tables

__all__ = ["synth_schedule", "synth_hourly", "synth_tables"]

def synth_schedule(use_case: str, n_evs: int, seed: int = 1234, year: int = 2020, days: int = 365,
                   minutes: int = 15) -> Schedule:
    """N-EV schedule on a regular grid starting `year`-01-01 00:00 (same span as the shipped files: 365 days), drawn by the
    package's schedule generator (fleetrl_amd/schedule_gen.py: the reference's sampling rules, one Philox stream per vehicle)."""
    from .schedule_gen import generate_schedule

    start = np.datetime64(f"{year}-01-01T00:00:00", "s")
    end = start + np.timedelta64((days * 24 * 60 // minutes - 1) * minutes * 60, "s")
    return generate_schedule(use_case, n_evs, str(start), str(end), seed=seed, minutes=minutes)


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
