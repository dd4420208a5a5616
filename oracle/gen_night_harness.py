#!/usr/bin/env python3
"""Golden action sequences of the reference's night-charging benchmark (TEST INFRASTRUCTURE, build container only).

Runs the UNMODIFIED `fleetrl.benchmarking.night_charging.NightCharging.run_benchmark` from /root/reference -- the loop whose
action rule `fleet_rollout_policy_dev(FLEET_ACT_POLICY_NIGHT)` and `oracle.fleet_oracle.NightChargingRule` restate
(benchmarking/night_charging.py:50-98) -- and records, for every step the harness takes, the env's clock and table row, the
action vector the harness passed to `step`, and `get_dist_factor()` at that moment (the lunch-hour rule of the caretaker
use case multiplies by it).  The harness needs stable-baselines3, which this image does not have: its two entry points are
replaced IN THIS PROCESS ONLY by an in-process one-env stand-in with SB3's semantics (`step` auto-resets a finished env,
`env_method` fans out, `VecNormalize` passes actions through unchanged), which is also where the recording happens.

    python oracle/gen_night_harness.py        ->  tests/golden/night_harness_<use_case>.npz

The fixture is data only (clock, rows, actions, the scalars the window is derived from)."""
from __future__ import annotations

import contextlib
import io
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
REC: list = []


class _OneEnvVec:
    """What `make_vec_env(FleetEnv, n_envs=1, vec_env_cls=SubprocVecEnv, env_kwargs=...)` hands to the harness, in process."""

    def __init__(self, env):
        self.env = env

    def reset(self):
        return np.asarray(self.env.reset()[0])[None]

    def step(self, actions):
        a = np.asarray(actions[0], dtype=np.float64)
        t = self.env.get_time()
        REC.append((t, int((t - self.env.db["date"].iloc[0]).total_seconds() // (60 * self.env.time_conf.minutes)), a.copy(),
                    np.asarray(self.env.get_dist_factor(), dtype=np.float64).copy()))
        obs, r, done, _trunc, info = self.env.step(a)
        if done:  # SB3 vec envs reset a finished env inside step()
            obs = self.env.reset()[0]
        return np.asarray(obs)[None], np.asarray([r]), np.asarray([done]), [info]

    def env_method(self, name, *args, **kw):
        return [getattr(self.env, name)(*args, **kw)]


class _PassThroughNormalize:
    def __init__(self, venv, **_kw):
        self.venv = venv

    def __getattr__(self, name):
        return getattr(self.venv, name)


def _install_sb3_stand_ins():
    sb3, common = types.ModuleType("stable_baselines3"), types.ModuleType("stable_baselines3.common")
    vec, util = types.ModuleType("stable_baselines3.common.vec_env"), types.ModuleType("stable_baselines3.common.env_util")
    vec.SubprocVecEnv = object
    vec.VecNormalize = _PassThroughNormalize
    util.make_vec_env = lambda env_cls, n_envs=1, vec_env_cls=None, env_kwargs=None, seed=None: _OneEnvVec(env_cls(**env_kwargs))
    sb3.common, common.vec_env, common.env_util = common, vec, util
    sys.modules.update({"stable_baselines3": sb3, "stable_baselines3.common": common,
                        "stable_baselines3.common.vec_env": vec, "stable_baselines3.common.env_util": util})


def generate(use_case: str, sched: str, hours: int, episodes: int, **extra):
    ref_harness._prepare_imports()
    _install_sb3_stand_ins()
    from fleetrl.benchmarking.night_charging import NightCharging

    cfg = ref_harness.base_config()
    cfg.update(use_case=use_case, schedule_name=sched, include_building=False, include_pv=False, calculate_degradation=False,
               deg_emp=False, episode_length=hours, time_picker="static", log_data=True)
    cfg.update(extra)
    REC.clear()
    bench = NightCharging(n_steps=hours, n_evs=1, n_episodes=episodes, n_envs=1, time_steps_per_hour=4)
    with contextlib.redirect_stdout(io.StringIO()):
        bench.run_benchmark(use_case=use_case, env_kwargs={"env_config": cfg}, seed=0)
        env = ref_harness.make_ref_env(dict(cfg))
    df = env.db
    leaving = df[(df["Location"].shift() == "home") & (df["Location"] == "driving")]
    out = dict(
        hour=np.asarray([t.hour for t, *_ in REC], dtype=np.int32), minute=np.asarray([t.minute for t, *_ in REC], dtype=np.int32),
        row=np.asarray([r for _, r, *_ in REC], dtype=np.int64), actions=np.stack([a for *_, a, _d in REC]),
        dist_factor=np.stack([d for *_, d in REC]),
        leave_hour=leaving["date"].dt.hour.to_numpy(np.int32), leave_minute=leaving["date"].dt.minute.to_numpy(np.int32),
        evse=np.float64(env.load_calculation.evse_max_power), cap=np.float64(env.ev_config.init_battery_cap),
        target_soc=np.float64(env.ev_config.target_soc), eff=np.float64(env.ev_config.charging_eff),
        minutes_per_step=np.int32(env.time_conf.minutes), episode_steps=np.int32(hours * 4), is_ct=np.bool_(use_case == "ct"))
    path = os.path.join(GOLDEN, f"night_harness_{use_case}.npz")
    np.savez_compressed(path, **out)
    a = out["actions"][:, 0]
    print(f"{path}: {len(REC)} steps, {int((a == 1).sum())} charging, {int((a == 0).sum())} idle, {int(((a > 0) & (a < 1)).sum())} partial")


if __name__ == "__main__":
    generate("lmd", "lmd_sched_single.csv", 72, 3)
    generate("ct", "ct_sched_single.csv", 72, 3)
    generate("ut", "ut_sched_single.csv", 48, 2)
