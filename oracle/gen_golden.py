#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):   python oracle/gen_golden.py [name ...]

For every configuration below it builds `E` independent reference `FleetEnv` objects, drives each through
`episodes` consecutive episodes with seeded float32-valued actions (fed to the reference as float64 copies,
SURVEY.md section 7 "NumPy-version trap"), emulating SB3's vec-env auto-reset (on done: keep the terminal
observation, call reset(), continue with the reset observation), and records everything the parity tests
compare:

  per step : obs f32, reward, done, cashflow, soc, hours_left, soh, soc_deg, target_soc, time row
  per reset: reset obs
  at the end of every episode: the persistent degradation state (rainflow_length, fd_cyc, fd_cal, l)
  tables   : the reference's own `db` columns, cut to the rows the episodes touch (plus look-ahead)
  scalars  : grid connection, EVSE power, battery sizes, scaled price multiplier, normaliser constants, obs_dim

Episode start rows are injected (the reference's pickers use unseeded `random`, quirk Q12).
The fixture is data only: no reference source text is stored.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_harness import make_ref_env, set_static_start, stacked_inputs_dir  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

# name -> (config overrides, n_evs (0 = shipped single-EV file), E, episodes, action mode)
CONFIGS = {
    # BASELINE C1 family: 1 env x 1 EV, lmd, price-only obs, linear degradation (quirk Q1 shim), 24 h
    "lmd1_price_linear": (dict(use_case="lmd", schedule_name="lmd_sched_single.csv", include_building=False,
                               include_pv=False, calculate_degradation=True, deg_emp=True, episode_length=24), 0, 2, 3, "wide"),
    # BASELINE C2 family: 5 EVs, lmd, price-only obs, linear degradation, 48 h
    "lmd5_price_linear": (dict(use_case="lmd", include_building=False, include_pv=False,
                               calculate_degradation=True, deg_emp=True, episode_length=48), 5, 3, 2, "charge"),
    # BASELINE C3 family: ct, load+pv obs, rainflow/SEI degradation, 48 h, 4 episodes (cross-episode state, quirk Q6)
    "ct5_both_rainflow": (dict(use_case="ct", building_name="load_ct.csv", include_building=True, include_pv=True,
                               calculate_degradation=True, deg_emp=False, episode_length=48), 5, 3, 4, "charge"),
    # BASELINE C4 family: ut, load+pv, normalised observations, rainflow, 48 h
    "ut3_both_norm_rainflow": (dict(use_case="ut", building_name="load_ut.csv", include_building=True, include_pv=True,
                                    normalize_in_env=True, calculate_degradation=True, deg_emp=False,
                                    episode_length=48), 3, 2, 2, "wide"),
    # the headline shape itself: 50 EVs per env (the reference needs ~1.6 s per step here, so one env, one 48 h episode)
    "ct50_both_rainflow": (dict(use_case="ct", building_name="load_ct.csv", include_building=True, include_pv=True,
                                calculate_degradation=True, deg_emp=False, episode_length=48), 50, 1, 1, "wide"),
    # the headline shape over an episode boundary: 50 EVs per env, 2 envs x 2 consecutive 48 h episodes -- quirk Q6 (rainflow_length /
    # fd_cyc / l carried across reset()) at the headline geometry, pinned by the reference itself (about 25 minutes of reference time)
    "ct50x2_both_rainflow_2ep": (dict(use_case="ct", building_name="load_ct.csv", include_building=True, include_pv=True,
                                      calculate_degradation=True, deg_emp=False, episode_length=48), 50, 2, 2, "charge"),
    # remaining observer variants
    "lmd2_building_norm_noaux": (dict(use_case="lmd", building_name="load_lmd.csv", include_building=True,
                                      include_pv=False, normalize_in_env=True, aux=False,
                                      calculate_degradation=False, episode_length=24), 2, 2, 2, "wide"),
    "ct2_pv_nodeg": (dict(use_case="ct", building_name="load_ct.csv", include_building=False, include_pv=True,
                          calculate_degradation=False, episode_length=24), 2, 2, 2, "charge"),
    # small grid connection so that the overload penalty fires; custom use case
    "custom3_both_overload": (dict(use_case="custom", building_name="load_lmd.csv", include_building=True,
                                   include_pv=True, calculate_degradation=True, deg_emp=False, episode_length=24,
                                   custom_ev_charger_power_in_kw=22, custom_ev_battery_size_in_kwh=40,
                                   custom_grid_connection_in_kw=200, init_battery_cap=40, obc_max_power=11,
                                   max_batt_cap_in_all_use_cases=60), 3, 2, 2, "full"),
    # same, with the reference's DataLogger switched on: pins `get_log()` (utils/data_logger/data_logger.py:21-68)
    "custom3_both_overload_log": (dict(use_case="custom", building_name="load_lmd.csv", include_building=True,
                                       include_pv=True, calculate_degradation=True, deg_emp=False, episode_length=24,
                                       custom_ev_charger_power_in_kw=22, custom_ev_battery_size_in_kwh=40,
                                       custom_grid_connection_in_kw=120, init_battery_cap=40, obc_max_power=22,
                                       max_batt_cap_in_all_use_cases=60, log_data=True), 3, 2, 2, "full"),
    # quirk Q7 (fleet_environment.py:263,382-392,613-618): `soh <= 0.9 -> target_soc[car] = 0.9`, sticky across resets.
    # init_soh just above 0.9 with the linear model: cars cross on their first or second daily 14:45 call (the calendar term
    # alone is 1.9e-7 per call), episode.soh restarts at init_soh on reset() while target_soc stays raised
    "lmd3_price_linear_q7cross": (dict(use_case="lmd", include_building=False, include_pv=False, calculate_degradation=True,
                                       deg_emp=True, episode_length=48, init_soh=0.9000003), 3, 2, 3, "wide"),
    # init_soh = 0.9 exactly and no degradation: the target is raised by the very first step and the reset laxity fix-up
    # (:382-392) of every later episode runs with 0.9; caretaker fleet, so the lunch target (:536-554) stays 0.65
    "ct3_both_nodeg_q7sticky": (dict(use_case="ct", building_name="load_ct.csv", include_building=True, include_pv=True,
                                     calculate_degradation=False, episode_length=24, init_soh=0.9), 3, 2, 3, "charge"),
    # same with normalised observations (oracle_normalization.py:127-131 divides the raised target by the configured one)
    "ut2_both_norm_linear_q7cross": (dict(use_case="ut", building_name="load_ut.csv", include_building=True, include_pv=True,
                                          normalize_in_env=True, calculate_degradation=True, deg_emp=True, episode_length=48,
                                          init_soh=0.9000003), 2, 2, 2, "wide"),
    # BASELINE C5 family: spot_2021 prices (year re-based by `_date_checker`, data_processing.py:419-431) and a feed-in
    # tariff that is NOT the spot price (`load_feed_in`, :297-318; inputs/fixed_feed_in.csv)
    "ct4_both_rainflow_spot21_feedin": (dict(use_case="ct", building_name="load_ct.csv", include_building=True, include_pv=True,
                                             calculate_degradation=True, deg_emp=False, episode_length=48,
                                             price_name="spot_2021_new.csv", tariff_name="fixed_feed_in.csv"), 4, 2, 2, "wide"),
    "lmd3_both_linear_spot21_tariff21": (dict(use_case="lmd", building_name="load_lmd.csv", include_building=True, include_pv=True,
                                              calculate_degradation=True, deg_emp=True, episode_length=24,
                                              price_name="spot_2021_new.csv", tariff_name="spot_2021_new_tariff.csv"), 3, 2, 2, "wide"),
    "ut3_both_norm_rainflow_spot21_feedin": (dict(use_case="ut", building_name="load_ut.csv", include_building=True, include_pv=True,
                                                  normalize_in_env=True, calculate_degradation=True, deg_emp=False, episode_length=24,
                                                  price_name="spot_2021_new.csv", tariff_name="fixed_feed_in.csv"), 3, 2, 2, "wide"),
    # all four `ignore_*` switches of adjust_score_config (fleet_environment.py:1070-1078) at once: price reward, overloading,
    # invalid-action and overcharging penalties zeroed; small grid connection and hard charging, so that every ignored term
    # would have fired (same fleet as custom3_both_overload)
    "custom3_both_overload_ignoreall": (dict(use_case="custom", building_name="load_lmd.csv", include_building=True,
                                             include_pv=True, calculate_degradation=True, deg_emp=False, episode_length=24,
                                             custom_ev_charger_power_in_kw=22, custom_ev_battery_size_in_kwh=40,
                                             custom_grid_connection_in_kw=200, init_battery_cap=40, obc_max_power=11,
                                             max_batt_cap_in_all_use_cases=60, ignore_price_reward=True,
                                             ignore_overloading_penalty=True, ignore_invalid_penalty=True,
                                             ignore_overcharging_penalty=True), 3, 2, 2, "full"),
}


def make_actions(rng: np.random.Generator, steps: int, n: int, mode: str) -> np.ndarray:
    if mode == "wide":
        a = rng.uniform(-1, 1, size=(steps, n))
    elif mode == "charge":
        a = rng.uniform(-0.3, 1, size=(steps, n))
    else:  # "full": mostly hard charging so that the grid connection overloads
        a = rng.uniform(0.4, 1, size=(steps, n))
        a[rng.random((steps, n)) < 0.1] = -1.0
    a[rng.random((steps, n)) < 0.15] = 0.0
    a[rng.random((steps, n)) < 0.03] = 1.0
    return a.astype(np.float32)


def run_config(name: str):
    ov, n_evs, E, episodes, mode = CONFIGS[name]
    ov = dict(ov)
    uc = ov["use_case"]
    if n_evs:
        sched_uc = "lmd" if uc == "custom" else uc
        dp, sched = stacked_inputs_dir(sched_uc, n_evs)
        ov.update(data_path=dp, schedule_name=sched)
    ov.setdefault("target_soc", 0.85)
    rng = np.random.default_rng(sum(map(ord, name.replace("_log", ""))))
    ep_steps = ov["episode_length"] * 4
    total = ep_steps * episodes
    rec = None
    starts = np.zeros((episodes, E), dtype=np.int32)
    t0 = time.time()
    db0 = None
    scalars = {}
    for e in range(E):
        env = make_ref_env(ov)
        N = int(env.num_cars)
        T = len(env.db) // N
        if rec is None:
            D = env.observation_space.shape[0]
            rec = dict(
                actions=np.zeros((E, total, N), np.float32), obs=np.zeros((E, total, D), np.float32),
                terminal_obs=np.zeros((E, episodes, D), np.float32), reset_obs=np.zeros((E, episodes, D), np.float32),
                reward=np.zeros((E, total)), done=np.zeros((E, total), np.uint8), cashflow=np.zeros((E, total)),
                soc=np.zeros((E, total, N)), hours_left=np.zeros((E, total, N)), soh=np.zeros((E, total, N)),
                soc_deg=np.zeros((E, total, N)), target_soc=np.zeros((E, total, N)), time_idx=np.zeros((E, total), np.int32),
                rf_len=np.zeros((E, episodes, N), np.int32), fd_cyc=np.zeros((E, episodes, N)),
                fd_cal=np.zeros((E, episodes, N)), sei_l=np.zeros((E, episodes, N)),
                reset_soc=np.zeros((E, episodes, N)), reset_hours_left=np.zeros((E, episodes, N)),
                dist_factor=np.zeros((E, episodes, N)),
            )
            db0 = env.db
            lc = env.load_calculation
            scalars = dict(
                grid_connection=lc.grid_connection, evse_power=lc.evse_max_power, batt_cap_nominal=lc.batt_cap,
                init_battery_cap=env.ev_config.init_battery_cap, price_multiplier=env.score_config.price_multiplier,
                obs_dim=D, num_cars=N, table_rows_full=T,
                obs_low=float(env.observation_space.low.min()), obs_high=float(env.observation_space.high.max()),
            )
            if ov.get("normalize_in_env"):
                nz = env.normalizer
                for k in ("max_time_left", "max_price", "min_price", "max_tariff", "min_tariff"):
                    scalars[k] = float(getattr(nz, k))
                if nz.building_flag:
                    scalars["max_building"] = float(nz.max_building)
                if nz.pv_flag:
                    scalars["max_pv"] = float(nz.max_pv)
                if nz.aux:
                    scalars["max_hours_needed"] = float(nz.max_hours_needed)
        dates0 = env.db["date"].values[:T]
        acts = make_actions(rng, total, N, mode)
        rec["actions"][e] = acts
        # start rows: anywhere in the training range, plus a forced 14:30 start (degradation on the very first step)
        # (kept inside one 45-day span per configuration so that the stored table window stays small)
        if e == 0:
            span0 = int(rng.integers(0, T - 1 - 60 * 96 - 45 * 96))
        cand = rng.integers(span0, span0 + 45 * 96, size=episodes)
        if e == 0 and episodes > 1:
            cand[1] = (cand[1] // 96) * 96 + 58
        starts[:, e] = cand
        k = 0
        for ep in range(episodes):
            set_static_start(env, int(starts[ep, e]))
            obs, _ = env.reset()
            rec["reset_obs"][e, ep] = obs
            rec["reset_soc"][e, ep] = np.asarray(env.episode.soc, dtype=np.float64)
            rec["reset_hours_left"][e, ep] = np.asarray(env.episode.hours_left, dtype=np.float64)
            if ov.get("aux", True):
                rec["dist_factor"][e, ep] = np.asarray(env.get_dist_factor(), dtype=np.float64)
            for _ in range(ep_steps):
                a64 = acts[k].astype(np.float64)
                obs, r, d, _tr, _info = env.step(a64)
                rec["obs"][e, k] = obs
                rec["reward"][e, k] = r
                rec["done"][e, k] = d
                rec["cashflow"][e, k] = env.episode.current_charging_expense
                rec["soc"][e, k] = np.asarray(env.episode.soc, dtype=np.float64)
                rec["hours_left"][e, k] = np.asarray(env.episode.hours_left, dtype=np.float64)
                rec["soh"][e, k] = np.asarray(env.episode.soh, dtype=np.float64)
                rec["soc_deg"][e, k] = np.asarray(env.episode.soc_deg, dtype=np.float64)
                rec["target_soc"][e, k] = np.asarray(env.target_soc, dtype=np.float64)
                rec["time_idx"][e, k] = int(np.searchsorted(dates0, np.datetime64(env.episode.time)))
                k += 1
            assert d, "episode must end exactly after episode_length hours"
            rec["terminal_obs"][e, ep] = obs
            if ov["calculate_degradation"] and not ov["deg_emp"]:
                sd = env.sei_deg
                rec["rf_len"][e, ep] = sd.rainflow_length
                rec["fd_cyc"][e, ep] = sd.fd_cyc
                rec["fd_cal"][e, ep] = sd.fd_cal
                rec["sei_l"][e, ep] = sd.l
        if ov.get("log_data"):
            lg = env.data_logger.log.reset_index(drop=True)
            rows = len(lg)
            if "log_reward" not in rec:
                rec.update(log_reward=np.zeros((E, rows)), log_cashflow=np.zeros((E, rows)), log_penalty=np.zeros((E, rows)),
                           log_grid=np.zeros((E, rows)), log_socv=np.zeros((E, rows)), log_episode=np.zeros((E, rows), np.int32),
                           log_time=np.zeros((E, rows), np.int64), log_deg=np.zeros((E, rows, N)),
                           log_charge=np.zeros((E, rows, N)), log_soh=np.zeros((E, rows, N)),
                           log_obs=np.zeros((E, rows, rec["obs"].shape[2]), np.float32), log_action=np.zeros((E, rows, N)))
            rec["log_reward"][e] = lg["Reward"].astype(float).values
            rec["log_cashflow"][e] = lg["Cashflow"].astype(float).values
            rec["log_penalty"][e] = lg["Penalties"].astype(float).values
            rec["log_grid"][e] = lg["Grid overloading"].astype(float).values
            rec["log_socv"][e] = lg["SOC violation"].astype(float).values
            rec["log_episode"][e] = lg["Episode"].astype(int).values
            rec["log_time"][e] = lg["Time"].values.astype("datetime64[s]").astype(np.int64)
            for k in range(rows):
                rec["log_deg"][e, k] = np.broadcast_to(np.asarray(lg["Degradation"].iloc[k], dtype=np.float64), (N,))
                rec["log_charge"][e, k] = np.asarray(lg["Charging energy"].iloc[k], dtype=np.float64)
                rec["log_soh"][e, k] = np.asarray(lg["SOH"].iloc[k], dtype=np.float64)
                rec["log_obs"][e, k] = np.asarray(lg["Observation"].iloc[k], dtype=np.float32)
                rec["log_action"][e, k] = np.asarray(lg["Action"].iloc[k], dtype=np.float64)
        print(f"  {name}: env {e + 1}/{E} done ({time.time() - t0:.0f}s)", flush=True)

    # ---- tables: the reference's db columns for the rows the episodes touch --------------------------------
    N = scalars["num_cars"]
    T = scalars["table_rows_full"]
    L2 = (ov.get("price_lookahead", 8) + 2) * 4 + 1
    w0 = (int(starts.min()) // 96) * 96
    w1 = min(T, int(starts.max()) + ep_steps + L2 + 1)
    col = lambda c: db0[c].values.reshape(N, T).T[w0:w1]  # noqa: E731
    dates = db0["date"].values[:T][w0:w1].astype("datetime64[s]")
    tables = dict(
        dates=dates.astype(np.int64), there=col("There").astype(np.uint8), time_left=col("time_left").astype(np.float64),
        soc_on_return=col("SOC_on_return").astype(np.float64),
        delu=db0["DELU"].values[:T][w0:w1], tariff=db0["tariff"].values[:T][w0:w1],
        prc=db0["price_reward_curve"].values[:T][w0:w1], trc=db0["tariff_reward_curve"].values[:T][w0:w1],
    )
    tables["load"] = db0["load"].values[:T][w0:w1] if "load" in db0 else np.zeros(w1 - w0)
    tables["pv"] = db0["pv"].values[:T][w0:w1] if "pv" in db0 else np.zeros(w1 - w0)
    # whole-table extrema the normaliser / grid sizing need (the window alone cannot give them)
    ext = dict(max_time_left=float(np.nanmax(db0["time_left"].values)), max_delu=float(np.nanmax(db0["DELU"].values)),
               min_delu=float(np.nanmin(db0["DELU"].values)), max_tariff=float(np.nanmax(db0["tariff"].values)),
               min_tariff=float(np.nanmin(db0["tariff"].values)),
               max_load=float(np.nanmax(db0["load"].values)) if "load" in db0 else 0.0,
               max_pv=float(np.nanmax(db0["pv"].values)) if "pv" in db0 else 0.0)
    scalars.update({f"ext_{k}": v for k, v in ext.items()})
    scalars["window_row0"] = w0

    import json

    from oracle.ref_harness import base_config

    full_cfg = base_config()
    full_cfg.update(ov)
    full_cfg["data_path"] = "<not shipped>"
    out = {f"tab_{k}": v for k, v in tables.items()}
    out.update({f"sc_{k}": np.asarray(v) for k, v in scalars.items()})
    out["cfg_json"] = np.asarray(json.dumps(full_cfg))
    out.update(rec)
    out["starts"] = starts - w0  # re-based to the window
    out["time_idx"] = rec["time_idx"] - w0
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, f"trace_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), window rows [{w0},{w1})", flush=True)


# ---------------------------------------------------------------------------------------------------------------
# real_time (event-skipping) traces: one agent step spans a variable number of table rows, an episode a variable
# number of agent steps -> own trace layout (tests/golden/rttrace_<name>.npz, read by tests/test_real_time_*.py)
# ---------------------------------------------------------------------------------------------------------------
RT_CONFIGS = {
    # name -> (overrides, n_evs, E, episodes)
    "ct5_both_rainflow": (dict(use_case="ct", building_name="load_ct.csv", include_building=True, include_pv=True,
                               calculate_degradation=True, deg_emp=False, episode_length=48, real_time=True), 5, 2, 2),
    "lmd3_price_linear": (dict(use_case="lmd", include_building=False, include_pv=False, calculate_degradation=True,
                               deg_emp=True, episode_length=24, real_time=True), 3, 2, 2),
    "ut2_both_norm_nodeg": (dict(use_case="ut", building_name="load_ut.csv", include_building=True, include_pv=True,
                                 normalize_in_env=True, calculate_degradation=False, episode_length=24, real_time=True), 2, 2, 2),
    # the reference's own irregular example: inputs/test_lmd.csv has rows at 00:00, 00:07, 00:15, then every 15 min (per-row dt,
    # fleet_environment.py:994-1022).  Only the load+pv observer runs on it (the others look the window end up by exact date).
    # n_evs = 0: the shipped single-EV file; the first episode of env 0 starts on row 0 and walks the irregular rows.
    # the reference's DataLogger in real_time mode: a row for EVERY table row of the skipping loop (fleet_environment.py:677-690)
    # real_time at the headline geometry: 50 EVs per env (1.6 s of reference time per table row), one env over an episode boundary
    "ct50_both_rainflow": (dict(use_case="ct", building_name="load_ct.csv", include_building=True, include_pv=True,
                                calculate_degradation=True, deg_emp=False, episode_length=48, real_time=True), 50, 1, 2),
    "ct3_both_rainflow_log": (dict(use_case="ct", building_name="load_ct.csv", include_building=True, include_pv=True,
                                   calculate_degradation=True, deg_emp=False, episode_length=24, real_time=True, log_data=True), 3, 2, 2),
    "lmd1_both_irregular": (dict(use_case="lmd", schedule_name="test_lmd.csv", building_name="load_lmd.csv", include_building=True,
                                 include_pv=True, calculate_degradation=True, deg_emp=True, episode_length=24, real_time=True), 0, 2, 2),
}


def make_actions_rt(rng: np.random.Generator, steps: int, n: int) -> np.ndarray:
    """Quiet actions (many exact zeros, small charging powers) so that rows really get skipped, with occasional bursts that
    trigger the penalty events."""
    a = rng.uniform(0.0, 0.5, size=(steps, n))
    a[rng.random((steps, n)) < 0.55] = 0.0
    burst = rng.random(steps) < 0.12
    a[burst] = rng.uniform(-1, 1, size=(int(burst.sum()), n))
    return a.astype(np.float32)


def run_config_rt(name: str):
    import json

    from oracle.ref_harness import base_config

    ov, n_evs, E, episodes = RT_CONFIGS[name]
    ov = dict(ov)
    if n_evs:
        dp, sched = stacked_inputs_dir(ov["use_case"], n_evs)
        ov.update(data_path=dp, schedule_name=sched)
    ov.setdefault("target_soc", 0.85)
    rng = np.random.default_rng(sum(map(ord, "rt_" + name)))
    ep_rows = ov["episode_length"] * 4
    cap = ep_rows * episodes  # an agent step spans >= 1 row
    rec, scalars, db0 = None, {}, None
    starts = np.zeros((episodes, E), dtype=np.int32)
    t0 = time.time()
    for e in range(E):
        env = make_ref_env(ov)
        N = int(env.num_cars)
        T = len(env.db) // N
        if rec is None:
            D = env.observation_space.shape[0]
            rec = dict(
                actions=np.zeros((E, cap, N), np.float32), obs=np.zeros((E, cap, D), np.float32),
                reset_obs=np.zeros((E, episodes, D), np.float32), n_steps=np.zeros((E, episodes), np.int32),
                reward=np.zeros((E, cap)), done=np.zeros((E, cap), np.uint8), cashflow=np.zeros((E, cap)),
                ep_return=np.zeros((E, cap)), soc=np.zeros((E, cap, N)), hours_left=np.zeros((E, cap, N)),
                soh=np.zeros((E, cap, N)), soc_deg=np.zeros((E, cap, N)), time_idx=np.zeros((E, cap), np.int32),
                rf_len=np.zeros((E, episodes, N), np.int32), fd_cyc=np.zeros((E, episodes, N)), sei_l=np.zeros((E, episodes, N)),
            )
            db0 = env.db
            lc = env.load_calculation
            scalars = dict(grid_connection=lc.grid_connection, evse_power=lc.evse_max_power, batt_cap_nominal=lc.batt_cap,
                           init_battery_cap=env.ev_config.init_battery_cap, price_multiplier=env.score_config.price_multiplier,
                           obs_dim=D, num_cars=N, table_rows_full=T)
        dates0 = env.db["date"].values[:T]
        acts = make_actions_rt(rng, cap, N)
        rec["actions"][e] = acts
        if e == 0:
            span0 = 0 if name.endswith("irregular") else int(rng.integers(0, T - 1 - 60 * 96 - 20 * 96))
        starts[:, e] = rng.integers(span0, span0 + 20 * 96, size=episodes)
        if name.endswith("irregular") and e == 0:
            starts[0, 0] = 0
        k = 0
        for ep in range(episodes):
            set_static_start(env, int(starts[ep, e]))
            obs, _ = env.reset()
            rec["reset_obs"][e, ep] = obs
            d, n = False, 0
            while not d:
                obs, r, d, _tr, _info = env.step(acts[k].astype(np.float64))
                rec["obs"][e, k] = obs
                rec["reward"][e, k] = r
                rec["done"][e, k] = d
                rec["cashflow"][e, k] = env.episode.current_charging_expense
                rec["ep_return"][e, k] = env.episode.cumulative_reward
                rec["soc"][e, k] = np.asarray(env.episode.soc, dtype=np.float64)
                rec["hours_left"][e, k] = np.asarray(env.episode.hours_left, dtype=np.float64)
                rec["soh"][e, k] = np.asarray(env.episode.soh, dtype=np.float64)
                rec["soc_deg"][e, k] = np.asarray(env.episode.soc_deg, dtype=np.float64)
                rec["time_idx"][e, k] = int(np.searchsorted(dates0, np.datetime64(env.episode.time)))
                k += 1
                n += 1
            rec["n_steps"][e, ep] = n
            if ov["calculate_degradation"] and not ov["deg_emp"]:
                sd = env.sei_deg
                rec["rf_len"][e, ep] = sd.rainflow_length
                rec["fd_cyc"][e, ep] = sd.fd_cyc
                rec["sei_l"][e, ep] = sd.l
        if ov.get("log_data"):
            lg = env.data_logger.log.reset_index(drop=True)
            rows = len(lg)
            if "log_reward" not in rec:
                rec.update(log_rows=np.zeros(E, np.int32), log_reward=np.zeros((E, cap)), log_cashflow=np.zeros((E, cap)),
                           log_penalty=np.zeros((E, cap)), log_grid=np.zeros((E, cap)), log_socv=np.zeros((E, cap)),
                           log_episode=np.zeros((E, cap), np.int32), log_time=np.zeros((E, cap), np.int64), log_deg=np.zeros((E, cap, N)),
                           log_charge=np.zeros((E, cap, N)), log_soh=np.zeros((E, cap, N)),
                           log_obs=np.zeros((E, cap, rec["obs"].shape[2]), np.float32), log_action=np.zeros((E, cap, N)))
            assert rows <= cap
            rec["log_rows"][e] = rows
            rec["log_reward"][e, :rows] = lg["Reward"].astype(float).values
            rec["log_cashflow"][e, :rows] = lg["Cashflow"].astype(float).values
            rec["log_penalty"][e, :rows] = lg["Penalties"].astype(float).values
            rec["log_grid"][e, :rows] = lg["Grid overloading"].astype(float).values
            rec["log_socv"][e, :rows] = lg["SOC violation"].astype(float).values
            rec["log_episode"][e, :rows] = lg["Episode"].astype(int).values
            rec["log_time"][e, :rows] = lg["Time"].values.astype("datetime64[s]").astype(np.int64)
            for q in range(rows):
                rec["log_deg"][e, q] = np.broadcast_to(np.asarray(lg["Degradation"].iloc[q], dtype=np.float64), (N,))
                rec["log_charge"][e, q] = np.asarray(lg["Charging energy"].iloc[q], dtype=np.float64)
                rec["log_soh"][e, q] = np.asarray(lg["SOH"].iloc[q], dtype=np.float64)
                rec["log_obs"][e, q] = np.asarray(lg["Observation"].iloc[q], dtype=np.float32)
                rec["log_action"][e, q] = np.asarray(lg["Action"].iloc[q], dtype=np.float64)
        print(f"  rt {name}: env {e + 1}/{E}: {rec['n_steps'][e].tolist()} agent steps for {ep_rows} rows per episode "
              f"({time.time() - t0:.0f}s)", flush=True)
    N, T = scalars["num_cars"], scalars["table_rows_full"]
    L2 = (ov.get("price_lookahead", 8) + 2) * 4 + 1
    w0 = (int(starts.min()) // 96) * 96
    w1 = min(T, int(starts.max()) + ep_rows + L2 + 1)
    col = lambda c: db0[c].values.reshape(N, T).T[w0:w1]  # noqa: E731
    one = lambda c: db0[c].values[:T][w0:w1] if c in db0 else np.zeros(w1 - w0)  # noqa: E731
    tables = dict(dates=db0["date"].values[:T][w0:w1].astype("datetime64[s]").astype(np.int64), there=col("There").astype(np.uint8),
                  time_left=col("time_left").astype(np.float64), soc_on_return=col("SOC_on_return").astype(np.float64),
                  delu=one("DELU"), tariff=one("tariff"), prc=one("price_reward_curve"), trc=one("tariff_reward_curve"),
                  load=one("load"), pv=one("pv"))
    ext = dict(max_time_left=float(np.nanmax(db0["time_left"].values)), max_delu=float(np.nanmax(db0["DELU"].values)),
               min_delu=float(np.nanmin(db0["DELU"].values)), max_tariff=float(np.nanmax(db0["tariff"].values)),
               min_tariff=float(np.nanmin(db0["tariff"].values)),
               max_load=float(np.nanmax(db0["load"].values)) if "load" in db0 else 0.0,
               max_pv=float(np.nanmax(db0["pv"].values)) if "pv" in db0 else 0.0)
    scalars.update({f"ext_{k}": v for k, v in ext.items()})
    scalars["window_row0"] = w0
    full_cfg = base_config()
    full_cfg.update(ov)
    full_cfg["data_path"] = "<not shipped>"
    used = int(rec["n_steps"].sum(axis=1).max())
    out = {f"tab_{k}": v for k, v in tables.items()}
    out.update({f"sc_{k}": np.asarray(v) for k, v in scalars.items()})
    out["cfg_json"] = np.asarray(json.dumps(full_cfg))
    out.update({k: (v[:, :used] if (v.ndim > 1 and v.shape[1] == cap and not k.startswith("log_")) else v) for k, v in rec.items()})
    out["starts"] = starts - w0
    out["time_idx"] = out["time_idx"] - w0
    path = os.path.join(GOLDEN, f"rttrace_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"rt {name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), window rows [{w0},{w1})", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "rt":
        for n in args[1:] or list(RT_CONFIGS):
            run_config_rt(n)
    else:
        for n in args or list(CONFIGS):
            run_config(n)
