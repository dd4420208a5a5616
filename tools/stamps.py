"""Diagnostic: where a wavefront of the step kernel spends its cycles (s_memtime stamps, FLEET_STAMPS build only).
Build (build container): python3 -c "from fleetrl_amd import build; build.build_variant('ab_stamps/x.so', ['-DFLEET_STAMPS'])"; run on
the GPU box: bash tools/r04_stamps.sh <tag> [E ...] (copies each ab_stamps/*.so over the library for the run and restores it)
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
rc = resolve_config(bench_config(E, N, "ct", deg=os.environ.get("DEG", "rainflow")))  # DEG=none|linear: diagnostics
tb = synth_tables("ct", N)
b = FleetBatch(make_params(rc, tb, E, seed=0), tb, time_features(tb))
dev = torch.device("cuda", 0)
tape = torch.rand((16, E, N), device=dev) * 2 - 1
obs = torch.empty((E, b.obs_dim), device=dev)
rew = torch.empty(E, device=dev, dtype=torch.float64)
done = torch.empty(E, device=dev, dtype=torch.uint8)
b.reset_dev(obs.data_ptr())
import time as _time
_n = int(os.environ.get("STEPS", 300))
LAUNCH = int(os.environ.get("LAUNCH", 0))  # 0: one hipLaunchKernel per step; 2: the library's own queue (FLEET_LAUNCH_DIRECT)
b.run_tape_dev(_n - 2000 if _n > 4000 else 0, tape.data_ptr(), 16, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), use_graph=LAUNCH)  # STEPS=20000: the steady state bench.py measures
b.synchronize()
_t0 = _time.perf_counter()
b.run_tape_dev(2000 if _n > 4000 else _n, tape.data_ptr(), 16, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), use_graph=LAUNCH)
b.synchronize()
print("launch period (%s, stamped build)" % ('own queue' if LAUNCH == 2 else 'eager') + " "
      ": %.2f us" % ((_time.perf_counter() - _t0) * 1e6 / (2000 if _n > 4000 else _n)))
lib = _capi.load_library()
SLOTS = 32
buf = np.zeros(4096 * SLOTS, dtype=np.uint64)
if LAUNCH == 2:  # the kernels ran from the code object the library loaded through HSA: its own stamp buffer
    lib.fleet_debug_read_stamps_direct.argtypes = [C.c_void_p, C.c_size_t]
    assert lib.fleet_debug_read_stamps_direct(buf.ctypes.data, buf.nbytes) == 0
else:
    lib.fleet_debug_read_stamps.argtypes = [C.c_void_p]
    assert lib.fleet_debug_read_stamps(buf.ctypes.data) == 0
s = buf.reshape(4096, SLOTS)[: min(E, 4096), :9].astype(np.int64)  # one row per wavefront (G = 64: one env each)
valid = (s > 0).all(axis=1)
s = s[valid]
names = ["entry->env head ready", "stage-2 issue + hot loads ready", "charge + state machine", "observation stores",
         "money terms + rainflow push", "state stores", "reductions + leader", "SEI pass / reset", ]
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
# inside the daily rainflow pass (stamps 11, 14, 15, 12, 13 are only written by the wavefronts on the 14:45 row)
full = buf.reshape(4096, SLOTS)[: min(E, 4096)].astype(np.int64)
sel = (full[:, 11] > full[:, 7]) & (full[:, 13] > full[:, 11]) & (full[:, 8] > full[:, 13]) & (full[:, 8] - full[:, 7] < 10**6)
if sel.any():
    f = full[sel]
    lds = (f[:, 14] > f[:, 11]) & (f[:, 15] > f[:, 14])
    seg = {"pass entered -> area taken, copy requested": f[:, 11] - f[:, 7]}
    if lds.any():
        g = f[lds]
        seg.update({"copy into the LDS": g[:, 14] - g[:, 11], "count of the pending points": g[:, 15] - g[:, 14],
                    "forced point + residuals": g[:, 12] - g[:, 15]})
    if not lds.any():
        seg["stack walk + cycle stresses"] = f[:, 12] - f[:, 11]
    seg.update({"SEI model (3 exp)": f[:, 13] - f[:, 12], "write-back -> exit": f[:, 8] - f[:, 13]})
    print(f"daily pass, {int(sel.sum())} wavefronts ({int(lds.sum())} through the LDS):",
          {k: (int(np.median(v)), int(np.percentile(v, 90)), int(v.max())) for k, v in seg.items()}, "(median, p90, max cycles)")

# the launch's timeline from the chip-wide 100 MHz counter (10 ns ticks)
rt = buf.reshape(4096, SLOTS)[: min(E, 4096), 9:11].astype(np.int64)
rt = rt[(rt > 0).all(axis=1)]
t0 = rt[:, 0].min()
us = lambda x: round(float(x) / 100.0, 2)  # noqa: E731
print("launch timeline [us from the first wave's entry]: wave entry median / p90 / last",
      [us(np.percentile(rt[:, 0], q) - t0) for q in (50, 90, 100)], " wave exit p10 / median / p90 / p99 / last",
      [us(np.percentile(rt[:, 1], q) - t0) for q in (10, 50, 90, 99, 100)], " wave life median / p90 / max",
      [us(np.percentile(rt[:, 1] - rt[:, 0], q)) for q in (50, 90, 100)])
# who finishes last: the 24 latest exits of the launch, by kind of extra work after the step
rt_all = buf.reshape(4096, SLOTS)[: min(E, 4096)].astype(np.int64)
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

# exit time by class: plain / daily row / episode end (+ auto-reset), from the chip-wide counter
daily = (rt_all[:, 11] > rt_all[:, 0]) & (rt_all[:, 8] > rt_all[:, 11])
heavy_tail = (rt_all[:, 8] - rt_all[:, 7]) > 2500
cls = np.where(daily, 1, np.where(heavy_tail, 2, 0))
print("exit time by class [us from the first wave's entry] (n, entry median, exit median / p90 / max, life median / max):")
for k, name in enumerate(("plain", "daily row", "episode end + reset")):
    m = ok & (cls == k)
    if m.any():
        en, ex = (rt_all[m, 9] - t0) / 100.0, (rt_all[m, 10] - t0) / 100.0
        print(f"  {name:20s} n={int(m.sum()):5d} entry {np.median(en):5.2f}  exit {np.median(ex):5.2f} / {np.percentile(ex, 90):5.2f} / {ex.max():5.2f}"
              f"  life {np.median(ex - en):5.2f} / {(ex - en).max():5.2f}")

# counting waves (rainflow count of the pending points inside the step, stamps 14 / 15 between 4 and 5)
cw = (rt_all[:, 14] > rt_all[:, 4]) & (rt_all[:, 15] > rt_all[:, 14]) & (rt_all[:, 5] >= rt_all[:, 15]) & (rt_all[:, 4] > rt_all[:, 0])
if cw.any():
    f = rt_all[cw]
    seg = {"money terms done -> window in the LDS (wait)": f[:, 14] - f[:, 4], "count + write-back": f[:, 15] - f[:, 14],
           "whole wave": f[:, 8] - f[:, 0]}
    print(f"counting wavefronts: {int(cw.sum())}:", {k: (int(np.median(v)), int(np.percentile(v, 90)), int(v.max())) for k, v in seg.items()},
          "(median, p90, max cycles);  non-counting whole wave median", int(np.median((rt_all[~cw & ok, 8] - rt_all[~cw & ok, 0]))))
    m = ok & cw
    en, ex = (rt_all[m, 9] - t0) / 100.0, (rt_all[m, 10] - t0) / 100.0
    print(f"  counting waves exit [us] median {np.median(ex):5.2f} p90 {np.percentile(ex, 90):5.2f} max {ex.max():5.2f}")

if cw.any():
    f = rt_all[cw]
    names2 = ["0 entry", "1 head", "2 hot ready (+window requested)", "3 state machine done", "4 window waited for, obs stored, money done", "14 count starts", "15 count + write-back done",
              "5 ev_finish", "6 before reductions", "7 leader done", "8 exit"]
    order = [0, 1, 2, 3, 4, 14, 15, 5, 6, 7, 8]
    print("counting wavefronts, median cycles between consecutive stamps:")
    for a_, b_, n_ in zip(order[:-1], order[1:], names2[1:]):
        dlt = f[:, b_] - f[:, a_]
        print(f"   -> {n_:52s} {int(np.median(dlt)):7d}  p90 {int(np.percentile(dlt, 90)):7d}  max {int(dlt.max()):7d}")


# ---- where the slow wavefronts run (round 5): slot 16 = HW_REG_HW_ID, 17 = HW_REG_XCC_ID -------------------------
hw = rt_all[:, 16]
if (hw != 0).any():
    simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, rt_all[:, 17] & 15
    life = (rt_all[:, 10] - rt_all[:, 9]) / 100.0
    exit_us = (rt_all[:, 10] - t0) / 100.0
    entry_us = (rt_all[:, 9] - t0) / 100.0
    m = ok & (cls == 0)
    cu_key = xcc * 10000 + se * 1000 + sh * 100 + cu
    simd_key = cu_key * 10 + simd
    import collections
    per_cu = collections.Counter(cu_key[ok].tolist())
    per_simd = collections.Counter(simd_key[ok].tolist())
    print("wavefronts per CU:", dict(collections.Counter(per_cu.values())), " per SIMD:", dict(collections.Counter(per_simd.values())),
          " CUs used:", len(per_cu))
    print("plain wavefronts, life [us] by die (XCC):", {int(x): (int((m & (xcc == x)).sum()), round(float(np.median(life[m & (xcc == x)])), 2),
          round(float(life[m & (xcc == x)].max()), 2)) for x in np.unique(xcc[m])}, "(n, median, max)")
    nw_cu = np.array([per_cu[k] for k in cu_key.tolist()])
    nw_simd = np.array([per_simd[k] for k in simd_key.tolist()])
    for name, arr in (("wavefronts on its CU", nw_cu), ("wavefronts on its SIMD", nw_simd)):
        print("plain wavefronts, life [us] by", name + ":", {int(v): (int((m & (arr == v)).sum()), round(float(np.median(life[m & (arr == v)])), 2),
              round(float(np.percentile(life[m & (arr == v)], 90)), 2), round(float(exit_us[m & (arr == v)].max()), 2)) for v in np.unique(arr[m])}, "(n, median, p90, last exit)")
    # is the tail a property of the CU?  spread of the per-CU median life, and the CUs of the 40 latest exits
    cu_med = {k: float(np.median(life[m & (cu_key == k)])) for k in per_cu if (m & (cu_key == k)).any()}
    v = np.array(list(cu_med.values()))
    print("median life per CU [us]: p10 %.2f p50 %.2f p90 %.2f max %.2f" % (*np.percentile(v, [10, 50, 90]), v.max()))
    late = np.argsort(np.where(m, exit_us, 0))[-40:][::-1]
    print("the 40 latest plain exits: (exit us, entry us, die, SE, CU, SIMD, wavefronts on its CU / SIMD, daily-row or reset wavefronts on its SIMD)")
    heavy_simd = collections.Counter(simd_key[ok & (cls != 0)].tolist())
    for w in late:
        print("   %.2f  %.2f  xcc %d se %d cu %2d simd %d   %d / %d   %d" % (exit_us[w], entry_us[w], xcc[w], se[w], sh[w] * 16 + cu[w], simd[w], nw_cu[w], nw_simd[w],
                                                                       heavy_simd.get(int(simd_key[w]), 0)))
    with_heavy = np.array([heavy_simd.get(int(k), 0) > 0 for k in simd_key.tolist()])
    for name, sel in (("shares its SIMD with a daily-row / reset wavefront", m & with_heavy), ("does not", m & ~with_heavy)):
        if sel.any():
            print("plain wavefronts whose SIMD %-52s n %5d life median %.2f p90 %.2f max %.2f  last exit %.2f" %
                  (name, sel.sum(), np.median(life[sel]), np.percentile(life[sel], 90), life[sel].max(), exit_us[sel].max()))
    # entry time against exit time: do the late starters end the launch?
    q = np.argsort(entry_us[m])
    ee, xx = entry_us[m][q], exit_us[m][q]
    n4 = len(ee) // 4
    print("plain wavefronts by entry quartile: entry median -> exit median / max:", [(round(float(np.median(ee[i * n4:(i + 1) * n4])), 2),
          round(float(np.median(xx[i * n4:(i + 1) * n4])), 2), round(float(xx[i * n4:(i + 1) * n4].max()), 2)) for i in range(4)])
