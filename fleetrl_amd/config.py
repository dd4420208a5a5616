"""Config resolution: the reference's flat config dict -> every scalar the step needs.

Mirrors, in order, what `FleetEnv.__init__` does with its `env_config`
(/root/reference/fleetrl/fleet_env/fleet_environment.py:121-325) and the three parameter classes it
builds (`EvConfig` fleet_env/config/ev_config.py:1-18, `ScoreConfig` score_config.py:3-24,
`TimeConfig` time_config.py:1-24), plus `LoadCalculation._import_company`
(utils/load_calculation/load_calculation.py:15-60).  Same keys, same defaults, same mandatory keys
(a missing mandatory key raises KeyError exactly where the reference indexes with `[]`).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass

__all__ = ["ResolvedConfig", "resolve_config", "read_config", "DEG_NONE", "DEG_LINEAR", "DEG_RAINFLOW",
           "PICK_STATIC", "PICK_RANDOM", "PICK_EVAL"]

DEG_NONE, DEG_LINEAR, DEG_RAINFLOW = 0, 1, 2
PICK_STATIC, PICK_RANDOM, PICK_EVAL = 0, 1, 2

# keys the reference reads with env_config["..."] (fleet_environment.py:129,139-173,193,198-199,204,210-211,220,231,243,285)
MANDATORY_KEYS = (
    "seed", "include_price", "include_building", "include_pv", "aux", "normalize_in_env", "data_path",
    "gen_schedule", "schedule_name", "gen_name", "gen_start_date", "gen_end_date", "gen_n_evs",
    "price_name", "tariff_name", "building_name", "pv_name", "use_case", "spot_markup", "spot_mul",
    "feed_in_ded", "max_batt_cap_in_all_use_cases", "episode_length", "target_soc",
    "ignore_price_reward", "ignore_overloading_penalty", "ignore_invalid_penalty",
    "ignore_overcharging_penalty", "verbose", "calculate_degradation", "log_data", "time_picker",
    "init_soh", "real_time", "deg_emp",
)


def read_config(conf_path: str) -> dict:
    """`FleetEnv.read_config` (fleet_environment.py:1024-1028)."""
    with open(conf_path, "r") as f:
        return json.load(f)


@dataclass
class ResolvedConfig:
    # flags
    include_price: bool
    include_building: bool
    include_pv: bool
    aux: bool
    normalize_in_env: bool
    calc_deg: bool
    deg_emp: bool
    real_time: bool
    use_case: str
    is_caretaker: bool
    time_picker: str
    # TimeConfig
    episode_length: int
    end_cutoff: int
    price_lookahead: int
    bl_pv_lookahead: int
    minutes: int
    time_steps_per_hour: int
    dt: float
    # EvConfig
    init_battery_cap: float
    obc_max_power: float
    charging_eff: float
    discharging_eff: float
    def_soc: float
    temperature: float
    target_soc: float
    target_soc_lunch: float
    min_laxity: float
    fixed_markup: float
    variable_multiplier: float
    feed_in_deduction: float
    # ScoreConfig
    price_multiplier: float
    fully_charged_reward: float
    penalty_invalid_action: float
    penalty_overcharging: float
    penalty_overloading: float
    clip_overcharging: float
    # misc
    init_soh: float
    eps: float
    seed: int | None
    raw: dict

    @property
    def deg_mode(self) -> int:
        if not self.calc_deg:
            return DEG_NONE
        return DEG_LINEAR if self.deg_emp else DEG_RAINFLOW

    @property
    def picker_mode(self) -> int:
        return {"static": PICK_STATIC, "random": PICK_RANDOM, "eval": PICK_EVAL}[self.time_picker]

    def company(self, num_cars: int, max_load: float):
        """(grid_connection, evse_max_power, batt_cap) -- `LoadCalculation._import_company`
        (load_calculation.py:15-60), including the hard-coded 1000 kW for utility fleets with N>1 (Q14)."""
        cfg = self.raw
        if self.use_case == "lmd":
            evse = 11
            grid = max(max_load * 1.1, max_load + 0.5 * num_cars * evse)
            batt = 60
        elif self.use_case == "ut":
            evse = 22
            grid = max(max_load * 1.1, max_load + 0.5 * num_cars * evse)
            if num_cars > 1:
                grid = 1000
            batt = 50
        elif self.use_case == "ct":
            evse = 4.6
            grid = max(max_load * 1.1, max_load + 0.5 * num_cars * evse)
            batt = 16.7
        elif self.use_case == "custom":
            evse = cfg.get("custom_ev_charger_power_in_kw", 120)
            grid = cfg.get("custom_grid_connection_in_kw", 500)
            batt = cfg.get("custom_ev_battery_size_in_kwh", 60)
        else:  # unreachable: resolve_config rejects unknown use cases like the reference (:1059-1060)
            raise TypeError("Company not recognised.")
        return float(grid), float(evse), float(batt)


def resolve_config(env_config: str | dict) -> ResolvedConfig:
    assert (env_config.__class__ == dict) or (env_config.__class__ == str), "Invalid config type."
    if env_config.__class__ == str:
        assert os.path.isfile(env_config), f"Config file not found at {env_config}."
        cfg = read_config(env_config)
    else:
        cfg = env_config
    for k in MANDATORY_KEYS:
        if k not in cfg:
            raise KeyError(k)

    use_case = cfg["use_case"]
    init_cap = cfg.get("init_battery_cap", 60.0)
    # specify_company_and_battery_size (fleet_environment.py:1041-1060)
    if use_case == "ct":
        init_cap = 16.7
    elif use_case == "ut":
        init_cap = 50.0
    elif use_case == "lmd":
        init_cap = 60.0
    elif use_case == "custom":
        init_cap = cfg["custom_ev_battery_size_in_kwh"]
    else:
        raise TypeError("Company not recognised.")

    fixed_markup = cfg.get("fixed_markup", 10)
    variable_multiplier = cfg.get("variable_multiplier", 1.5)
    feed_in_deduction = cfg.get("feed_in_deduction", 0.25)
    # change_markups (:1062-1068)
    if cfg["spot_markup"] is not None:
        fixed_markup = cfg["spot_markup"]
    if cfg["spot_mul"] is not None:
        variable_multiplier = cfg["spot_mul"]
    if cfg["feed_in_ded"] is not None:
        feed_in_deduction = cfg["feed_in_ded"]

    # price multiplier scaled with battery size (:193-195)
    price_multiplier = cfg.get("price_multiplier", 3.33) * (cfg["max_batt_cap_in_all_use_cases"] / init_cap)
    p_inv = cfg.get("penalty_invalid_action", -0.2)
    p_oc = cfg.get("penalty_overcharging", -0.0055)
    p_ovl = cfg.get("penalty_overloading", 1)
    # adjust_score_config (:1070-1078)
    if cfg["ignore_price_reward"]:
        price_multiplier = 0
    if cfg["ignore_overloading_penalty"]:
        p_ovl = 0
    if cfg["ignore_invalid_penalty"]:
        p_inv = 0
    if cfg["ignore_overcharging_penalty"]:
        p_oc = 0

    minutes = cfg.get("minutes", 15)
    if cfg["time_picker"] not in ("static", "random", "eval"):
        raise TypeError("Time picker type not recognised")

    return ResolvedConfig(
        include_price=bool(cfg["include_price"]),
        include_building=bool(cfg["include_building"]),
        include_pv=bool(cfg["include_pv"]),
        aux=bool(cfg["aux"]),
        normalize_in_env=bool(cfg["normalize_in_env"]),
        calc_deg=bool(cfg["calculate_degradation"]),
        deg_emp=bool(cfg["deg_emp"]),
        real_time=bool(cfg["real_time"]),
        use_case=use_case,
        is_caretaker=(use_case == "ct"),
        time_picker=cfg["time_picker"],
        episode_length=int(cfg["episode_length"]),
        end_cutoff=int(cfg.get("end_cutoff", 60)),
        price_lookahead=int(cfg.get("price_lookahead", 8)),
        bl_pv_lookahead=int(cfg.get("bl_pv_lookahead", 4)),
        minutes=int(minutes),
        time_steps_per_hour=int(cfg.get("time_steps_per_hour", 4)),
        dt=minutes / 60,
        init_battery_cap=float(init_cap),
        obc_max_power=float(cfg.get("obc_max_power", 100.0)),
        charging_eff=float(cfg.get("charging_eff", 0.91)),
        discharging_eff=float(cfg.get("discharging_eff", 0.91)),
        def_soc=float(cfg.get("def_soc", 0.5)),
        temperature=float(cfg.get("temperature", 25.0)),
        target_soc=float(cfg["target_soc"]),
        target_soc_lunch=float(cfg.get("target_soc_lunch", 0.65)),
        min_laxity=float(cfg.get("min_laxity", 2)),
        fixed_markup=float(fixed_markup),
        variable_multiplier=float(variable_multiplier),
        feed_in_deduction=float(feed_in_deduction),
        price_multiplier=float(price_multiplier),
        fully_charged_reward=float(cfg.get("fully_charged_reward", 1)),
        penalty_invalid_action=float(p_inv),
        penalty_overcharging=float(p_oc),
        penalty_overloading=float(p_ovl),
        clip_overcharging=float(cfg.get("clip_overcharging", -0.2)),
        init_soh=float(cfg["init_soh"]),
        eps=0.005,  # fleet_environment.py:230
        seed=cfg["seed"],
        raw=cfg,
    )
