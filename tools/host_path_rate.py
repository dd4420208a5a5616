"""PCIe-inclusive rate of the host-pointer path (`fleet_step_host` through FleetVecEnv.step): actions H2D, kernel,
observations / rewards / dones D2H, per step.  Reported in DESIGN.md section 6; never used as bench.py's `value`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bench_config  # noqa: E402
from fleetrl_amd import FleetVecEnv  # noqa: E402
from fleetrl_amd.synth import synth_tables  # noqa: E402

E, N = 4096, 50
venv = FleetVecEnv(bench_config(E, N, "ct"), E, tables=synth_tables("ct", N))
venv.reset()
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, size=(8, E, N)).astype(np.float32)
for i in range(20):
    venv.step(acts[i % 8])
t0 = time.perf_counter()
n = 300
for i in range(n):
    venv.step(acts[i % 8])
dt = time.perf_counter() - t0
print(f"host path: {dt / n * 1e6:.1f} us/step, {E * n / dt:.3e} env-steps/s (4096 envs x 50 EVs, NumPy in/out)")
