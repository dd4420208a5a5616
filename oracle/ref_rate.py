#!/usr/bin/env python3
"""Step rate of the UNMODIFIED Python reference on this container's host cores (build container only).

    python oracle/ref_rate.py            ->  profiles/r02_reference_python_rate.json

What BASELINE.md quotes from the survey session, re-measured with a committed harness: env-steps/s of one reference `FleetEnv`
process at 1, 5 and 50 EVs (caretaker fleet, load+pv observations, rainflow degradation, 48 h episodes, random actions) -- the
"reference" CPU baseline beside `bench.py`'s `cpu_baseline` (kind "port": the C oracle).  The reference cannot run on the GPU box
(/root/reference is not there), so this figure is from the build container's CPU and is reported as such.
"""
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_harness import make_ref_env, set_static_start, stacked_inputs_dir  # noqa: E402


def rate(n_evs: int, budget_s: float):
    ov = dict(use_case="ct", building_name="load_ct.csv", include_building=True, include_pv=True, calculate_degradation=True,
              deg_emp=False, episode_length=48)
    if n_evs > 1:
        dp, sched = stacked_inputs_dir("ct", n_evs)
        ov.update(data_path=dp, schedule_name=sched)
    else:
        ov.update(schedule_name="ct_sched_single.csv")
    env = make_ref_env(ov)
    rng = np.random.default_rng(0)
    set_static_start(env, 96 * 40)
    env.reset()
    for _ in range(2):
        env.step(rng.uniform(-1, 1, n_evs))
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s and steps < 180:
        env.step(rng.uniform(-1, 1, n_evs))
        steps += 1
    dt = time.perf_counter() - t0
    return {"evs": n_evs, "steps": steps, "seconds": dt, "env_steps_per_s": steps / dt}


def main():
    out = {"what": "one process of the unmodified reference FleetEnv, ct fleet, load+pv, rainflow, numpy " + np.__version__,
           "host": platform.processor() or platform.machine(), "cpus": os.cpu_count(),
           "rates": [rate(1, 10), rate(5, 15), rate(50, 40)]}
    path = os.path.join(ROOT, "profiles", "r02_reference_python_rate.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
