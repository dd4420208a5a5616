#!/usr/bin/env python3
"""The reference's evaluation loop (benchmarking/uncontrolled_charging.py:33-56) on the MI355X step, two ways:

  1. the SB3-style host loop: `FleetVecEnv.step(actions)` once per 15-minute row, NumPy in / out;
  2. the same policy evaluated on the device: `run_policy(batch, "uncontrolled", steps)`, a handful of launches,

and the reference's DataLogger frame (`get_log()`) after either.  Needs an MI355X; inputs are synthetic (no CSV files):

    python examples/vec_env_loop.py [num_envs] [n_evs]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bench_config  # noqa: E402  (the reference's config dict with the benchmark's values)
from fleetrl_amd import FleetVecEnv  # noqa: E402
from fleetrl_amd.policies import run_policy  # noqa: E402
from fleetrl_amd.synth import synth_tables  # noqa: E402


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cfg = bench_config(E, N, "ct")
    cfg.update(log_data=True, time_picker="static", episode_length=48)
    steps = 4 * 48 - 1  # one episode minus its last step (which the reference does not log)
    tables = synth_tables("ct", N, seed=7)

    host = FleetVecEnv(cfg, E, tables=tables)
    host.reset()
    t0 = time.perf_counter()
    ret = np.zeros(E)
    for _ in range(steps):
        _obs, rew, _dones, _infos = host.step(np.ones((E, N), dtype=np.float32))  # "uncontrolled charging": always full power
        ret += rew
    t_host = time.perf_counter() - t0

    dev = FleetVecEnv(cfg, E, tables=tables)
    dev.reset()
    t0 = time.perf_counter()
    _obs, rsum, _dcount = run_policy(dev.core.batch, "uncontrolled", steps)
    t_dev = time.perf_counter() - t0

    log_h, log_d = host.env_method("get_log")[0], dev.env_method("get_log")[0]
    print(f"{E} envs x {N} EVs, {steps} steps of uncontrolled charging")
    print(f"  host loop     : {t_host * 1e3:8.1f} ms, mean return {ret.mean():.3f}")
    print(f"  on the device : {t_dev * 1e3:8.1f} ms, mean return {rsum.mean():.3f}")
    print(f"  log rows per env {len(log_h)} / {len(log_d)}, columns {list(log_d.columns)}")
    assert np.allclose(ret, rsum, rtol=1e-5, atol=1e-3) and len(log_h) == len(log_d)
    print(f"  cashflow of env 0 over the episode: {log_d['Cashflow'].sum():.2f} EUR, SOC violations: {(log_d['SOC violation'] > 0).sum()}")
    host.close()
    dev.close()


if __name__ == "__main__":
    main()
