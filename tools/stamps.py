"""Diagnostic: where a wavefront of the step kernel spends its cycles (s_memtime stamps, FLEET_STAMPS build only).
Run on the GPU box:  FLEET_EXTRA_HIPCC_FLAGS=-DFLEET_STAMPS python3 -c "from fleetrl_amd import build; build.build(force=True)"; python3 tools/stamps.py
Read the SHARES, not the absolute run time (the stamps serialise the schedule)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import bench_config  # noqa: E402
from fleetrl_amd import _capi  # noqa: E402
from fleetrl_amd.batch import FleetBatch  # noqa: E402
from fleetrl_amd.config import resolve_config  # noqa: E402
from fleetrl_amd.params import make_params, time_features  # noqa: E402
from fleetrl_amd.synth import synth_tables  # noqa: E402

E, N = int(os.environ.get("E", 4096)), 50
rc = resolve_config(bench_config(E, N, "ct"))
tb = synth_tables("ct", N)
b = FleetBatch(make_params(rc, tb, E, seed=0), tb, time_features(tb))
dev = torch.device("cuda", 0)
tape = torch.rand((16, E, N), device=dev) * 2 - 1
obs = torch.empty((E, b.obs_dim), device=dev)
rew = torch.empty(E, device=dev, dtype=torch.float64)
done = torch.empty(E, device=dev, dtype=torch.uint8)
b.reset_dev(obs.data_ptr())
b.run_tape_dev(int(os.environ.get("STEPS", 300)), tape.data_ptr(), 16, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), use_graph=False)  # STEPS=20000: the steady state bench.py measures
b.synchronize()
lib = _capi.load_library()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.fleet_debug_read_stamps.argtypes = [C.c_void_p]
assert lib.fleet_debug_read_stamps(buf.ctypes.data) == 0
s = buf.reshape(4096, 16)[: min(E, 4096), :9].astype(np.int64)  # one row per wavefront (G = 64: one env each)
valid = (s > 0).all(axis=1)
s = s[valid]
names = ["entry->env head ready", "stage-2 issue + hot loads ready", "charge + state machine", "observation stores",
         "rainflow update", "state stores", "reductions + leader", "SEI pass / reset", ]
d = np.diff(s, axis=1)
print("cycles per segment, median over the wavefronts (last step of the run):")
for k, n in enumerate(names):
    print(f"  {n:34s} {np.median(d[:, k]):8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}")
tot = s[:, 8] - s[:, 0]
print(f"  {'total':34s} {np.median(tot):8.0f}   p90 {np.percentile(tot, 90):8.0f}   max {tot.max():8.0f}")
long = d[:, 7] > 4 * np.median(d[:, 7])
print(f"wavefronts with a long last segment (burst / daily evaluation / reset): {long.mean() * 100:.1f} %")
for name, sel in (("ordinary", ~long), ("long", long)):
    if sel.any():
        print(f"  {name:9s} last segment median {np.median(d[sel, 7]):8.0f}  p90 {np.percentile(d[sel, 7], 90):8.0f}   total median "
              f"{np.median(tot[sel]):8.0f}  p90 {np.percentile(tot[sel], 90):8.0f}")
# inside the daily evaluation (stamps 11..13 are only written by the wavefronts on the 14:45 row)
full = buf.reshape(4096, 16)[: min(E, 4096)].astype(np.int64)
sel = (full[:, 11] > full[:, 7]) & (full[:, 13] > full[:, 11]) & (full[:, 8] > full[:, 13]) & (full[:, 8] - full[:, 7] < 10**6)
if sel.any():
    f = full[sel]
    seg = {"second pass entered -> records arrived": f[:, 11] - f[:, 7], "stack walk + cycle stresses": f[:, 12] - f[:, 11],
           "SEI model (3 exp)": f[:, 13] - f[:, 12], "stores -> exit": f[:, 8] - f[:, 13]}
    print(f"daily evaluation, {int(sel.sum())} wavefronts:", {k: (int(np.median(v)), int(np.percentile(v, 90))) for k, v in seg.items()})

if (full[:, 14] > full[:, 6]).any():  # split push: stamp 14 = env-level work done, before the second half of the lane's step
    m14 = (full[:, 14] > full[:, 6]) & (full[:, 7] > full[:, 14])
    print("split step: reductions + leader", int(np.median(full[m14, 14] - full[m14, 6])), " finish push + state stores",
          int(np.median(full[m14, 7] - full[m14, 14])), "p90", int(np.percentile(full[m14, 7] - full[m14, 14], 90)))
# inside the auto-reset of an env whose episode ended in this step (stamps 14, 15)
sel = (full[:, 14] > full[:, 7]) & (full[:, 15] > full[:, 14]) & (full[:, 8] > full[:, 15]) & (full[:, 8] - full[:, 7] < 10**6)
if sel.any():
    f = full[sel]
    seg = {"second pass entered -> start row chosen": f[:, 14] - f[:, 7], "table records -> state written": f[:, 15] - f[:, 14],
           "observation tail, env record -> exit": f[:, 8] - f[:, 15]}
    print(f"auto-reset, {int(sel.sum())} wavefronts:", {k: (int(np.median(v)), int(np.percentile(v, 90))) for k, v in seg.items()})

# the launch's timeline from the chip-wide 100 MHz counter (10 ns ticks)
rt = buf.reshape(4096, 16)[: min(E, 4096), 9:11].astype(np.int64)
rt = rt[(rt > 0).all(axis=1)]
t0 = rt[:, 0].min()
us = lambda x: round(float(x) / 100.0, 2)  # noqa: E731
print("launch timeline [us from the first wave's entry]: wave entry median / p90 / last",
      [us(np.percentile(rt[:, 0], q) - t0) for q in (50, 90, 100)], " wave exit p10 / median / p90 / p99 / last",
      [us(np.percentile(rt[:, 1], q) - t0) for q in (10, 50, 90, 99, 100)], " wave life median / p90 / max",
      [us(np.percentile(rt[:, 1] - rt[:, 0], q)) for q in (50, 90, 100)])
# who finishes last: the 24 latest exits of the launch, by kind of extra work after the step
rt_all = buf.reshape(4096, 16)[: min(E, 4096)].astype(np.int64)
ok = (rt_all[:, 9] > 0) & (rt_all[:, 10] > 0)
t0 = rt_all[ok, 9].min()
late = np.argsort(np.where(ok, rt_all[:, 10], 0))[-24:][::-1]
kinds = []
for w in late:
    f = rt_all[w]
    evald = f[13] > f[7] and f[8] > f[13] and f[8] - f[7] < 10**6
    kinds.append(("eval" if evald else ("reset/other" if f[8] - f[7] > 2500 else "ordinary"), round((f[10] - t0) / 100.0, 2), int(f[8] - f[7]), int(f[7] - f[0])))
print("latest exits (kind, exit us, cycles after the step, cycles of the step):", kinds)
print("debug: waves with stamp14>0:", int((full[:, 14] > 0).sum()), " with 14>7:", int((full[:, 14] > full[:, 7]).sum()), " 15>14:", int((full[:, 15] > full[:, 14]).sum()),
      " 8>15:", int((full[:, 8] > full[:, 15]).sum()))
# where the slowest ordinary wavefronts (no extra work after the step) spend their time, against the median one
ordn = ~long
tt = tot.copy()
tt[~ordn] = 0
slow = np.argsort(tt)[-max(8, len(tt) // 50):]
print("slowest 2 % of the ordinary wavefronts, median cycles per segment (all ordinary in brackets):")
for k, n in enumerate(names):
    print(f"  {n:34s} {np.median(d[slow, k]):8.0f}   ({np.median(d[ordn, k]):.0f})")
