"""Harness that imports the UNMODIFIED Python reference from /root/reference (TEST INFRASTRUCTURE ONLY).

Runs only in the build container: /root/reference does not exist on the GPU box, and
nothing under tests/ -m gpu, bench.py or smoke() may import this module.  It is used by
`oracle/gen_golden.py` to produce the committed fixtures under tests/golden/ and by the
container-only tests that compare our restatements with the live reference.

Recipe:
  * sys.path = [oracle/stubs, /root/reference]  (stand-ins for `gymnasium`, `rainflow`)
  * matplotlib Agg backend (the reference imports a renderer)
  * base config = /root/reference/config.json + overrides
  * linear degradation mode needs the one-line shim `env.sei_deg = env.emp_deg`
    because FleetEnv.step always calls self.sei_deg
    (/root/reference/fleetrl/fleet_env/fleet_environment.py:285-288 vs :666) -- quirk Q1.
  * actions are fed as float64 copies of float32 values: under the reference's pinned
    numpy 1.26 `python_float * np.float32` is float64, under numpy>=2 it is float32.
"""
import contextlib
import io
import json
import os
import sys
import warnings

REFERENCE_ROOT = "/root/reference"
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "fleetrl", "fleet_env", "fleet_environment.py"))


def _prepare_imports():
    if not reference_available():
        raise RuntimeError("reference sources are not present (only available in the build container)")
    for p in (REFERENCE_ROOT, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    import matplotlib

    matplotlib.use("Agg")
    warnings.filterwarnings("ignore")


def base_config() -> dict:
    with open(os.path.join(REFERENCE_ROOT, "config.json")) as f:
        cfg = json.load(f)
    cfg.update(
        data_path=os.path.join(REFERENCE_ROOT, "inputs"),
        price_name="spot_2020_new.csv",
        tariff_name="spot_2020_new_tariff.csv",
        include_price=True,
        aux=True,
        normalize_in_env=False,
        time_picker="static",
        init_battery_cap=60,
        obc_max_power=100,
        max_batt_cap_in_all_use_cases=60,
        min_laxity=2,
        seed=0,
        verbose=0,
        log_data=False,
    )
    return cfg


def make_ref_env(overrides: dict, quiet: bool = True):
    """Build a reference FleetEnv from base_config() + overrides; applies the Q1 shim when deg_emp."""
    _prepare_imports()
    from fleetrl.fleet_env.fleet_environment import FleetEnv  # noqa: E402

    cfg = base_config()
    cfg.update(overrides)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext():
        env = FleetEnv(cfg)
    if cfg["deg_emp"]:
        env.sei_deg = env.emp_deg  # Q1 shim: linear mode invoked at the reference's own call site
    return env


def set_static_start(env, start_index: int):
    """Make the reference start its next episode at table row `start_index` (Q12: the reference's
    pickers use unseeded python `random`; parity tests inject start indices instead)."""
    import pandas as pd

    dates = env.db.loc[env.db["ID"] == 0, "date"].reset_index(drop=True)
    ts = dates.iloc[int(start_index)]

    class _Fixed:
        def choose_time(self, db, freq, end_cutoff):
            return pd.Timestamp(ts)

    env.time_picker = _Fixed()
    return ts


def stacked_inputs_dir(use_case: str, n_evs: int, workdir: str = "/tmp/fleetrl_oracle_inputs") -> tuple[str, str]:
    """Write an N-EV schedule CSV for the reference to read (quirk Q3: the multi-EV blobs are missing).

    Recipe (same as fleetrl_amd.prestage.stack_single_ev_schedules): car i takes
    inputs/1_{uc}.csv if i is even else inputs/1_{uc}_eval.csv; its payload columns are np.roll-ed
    by (i//2)*7*96 rows; ID=i; concatenated in ID order.  The directory also links every other
    input file so it can serve as `data_path`.  Returns (data_path, schedule_name).
    """
    import numpy as np
    import pandas as pd

    os.makedirs(workdir, exist_ok=True)
    src = os.path.join(REFERENCE_ROOT, "inputs")
    for f in os.listdir(src):
        dst = os.path.join(workdir, f)
        if not os.path.exists(dst):
            os.symlink(os.path.join(src, f), dst)
    name = f"stack{n_evs}_{use_case}.csv"
    out = os.path.join(workdir, name)
    if not os.path.exists(out):
        base = pd.read_csv(os.path.join(src, f"1_{use_case}.csv"), index_col=0)
        alt = pd.read_csv(os.path.join(src, f"1_{use_case}_eval.csv"), index_col=0)
        parts = []
        for i in range(n_evs):
            df = (base if i % 2 == 0 else alt).copy()
            k = (i // 2) * 7 * 96
            for col in ("Distance_km", "Consumption_kWh", "Location", "ChargingStation", "PowerRating_kW"):
                df[col] = np.roll(df[col].values, k)
            df["ID"] = i
            parts.append(df)
        pd.concat(parts, ignore_index=True).to_csv(out)
    return workdir, name
