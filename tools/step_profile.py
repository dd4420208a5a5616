"""Diagnostic: per-launch duration of the step kernel when every env sits on the SAME table row (all envs start at one
row): shows what the rare rows cost -- the daily 14:45 degradation row, the episode end -- compared with an ordinary row,
i.e. how long the slowest wavefronts of a launch are when start rows are random.  GPU box: python3 tools/step_profile.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import bench_config  # noqa: E402
from fleetrl_amd.batch import FleetBatch  # noqa: E402
from fleetrl_amd.config import resolve_config  # noqa: E402
from fleetrl_amd.params import make_params, time_features  # noqa: E402
from fleetrl_amd.synth import synth_tables  # noqa: E402

E, N = int(os.environ.get("E", 4096)), 50
rc = resolve_config(bench_config(E, N, "ct"))
tb = synth_tables("ct", N)
b = FleetBatch(make_params(rc, tb, E, seed=0), tb, time_features(tb))
start = 96 * 10 + 40  # 10:00 of day 10
b.set_start_schedule(np.full((1, E), start, dtype=np.int32))
dev = torch.device("cuda", 0)
L = 64
tape = torch.rand((L, E, N), device=dev) * 2 - 1
tape[torch.rand((L, E, N), device=dev) < 0.15] = 0
obs = torch.empty((E, b.obs_dim), device=dev)
rew = torch.empty(E, device=dev, dtype=torch.float64)
done = torch.empty(E, device=dev, dtype=torch.uint8)
b.reset_dev(obs.data_ptr())
for rep in range(3):
    ms = b.time_steps_dev(192, tape.data_ptr(), L, obs.data_ptr(), rew.data_ptr(), done.data_ptr())
us = ms * 1e3
rows = (start + 1 + np.arange(192)) % 96  # row reached by each step (15-min slot of the day); the episode is 192 steps
deg = rows == 59  # 14:45
print("ordinary rows: median %.2f us, p10 %.2f, p90 %.2f" % (np.median(us[~deg][:-1]), np.percentile(us[~deg][:-1], 10), np.percentile(us[~deg][:-1], 90)))
print("14:45 rows   :", np.round(us[deg], 2))
print("episode end  : %.2f us" % us[-1])
print("by slot of day (us):", {int(r): round(float(np.mean(us[rows == r])), 1) for r in (20, 24, 28, 44, 52, 56, 59, 60, 64, 76, 90)})
