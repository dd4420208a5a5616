# usage (GPU box): bash tools/r05_outliers.sh [bench args] -- which launches are the 4-5x outliers of the single-step kernel?
#   Per-launch durations in launch order (rocprofv3 --kernel-trace); every env of the bench starts its first
# episode in the same launch and episodes are 192 steps, so launch k (counted from the first reset) has every env at episode
# step k mod 192: the 20 slowest launches are listed with that phase, with their distance to the previous launch (a launch
# behind a host gap starts on cold caches / a ramping clock) and with the durations of their neighbours.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/outl; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/r05; cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline --no-host-path "$@" > $OUT/bench.log 2>&1
python3 - $OUT <<'PY' | tee $R/gpurun_out/r05/outliers.log
import glob, sys
import numpy as np, pandas as pd
f = glob.glob(f"{sys.argv[1]}/kt/*/*kernel_trace.csv")[0]
c = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
names = c.Kernel_Name.str.replace(r"\(.*", "", regex=True)
step = c.Kernel_Name.str.contains(r"fleet_step_kernel<\d+, \d+, false,", regex=True).to_numpy()
multi = c.Kernel_Name.str.contains(r"fleet_step_kernel<\d+, \d+, true,", regex=True).to_numpy()
reset = c.Kernel_Name.str.contains("fleet_reset_kernel").to_numpy()
print("kernels in the trace:", dict(names.value_counts().head(8)))
st, en = c.Start_Timestamp.to_numpy().astype(np.int64), c.End_Timestamp.to_numpy().astype(np.int64)
d = (en - st).astype(float)
# episode phase: single-step launches since the last reset kernel / K-step launch (which advance the batch by other amounts: the
# phase is only tracked inside runs of single-step launches that follow a reset kernel)
phase = np.full(len(c), -1)
k = -1
for i in range(len(c)):
    if reset[i]:
        k = 0
    elif multi[i]:
        k = -1
    elif step[i] and k >= 0:
        phase[i] = k % 192
        k += 1
idx = np.flatnonzero(step)
ds = d[idx]
gap = np.r_[1e9, (st[idx][1:] - en[idx][:-1]).astype(float)]
print("single-step launches %d: p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f mean %.0f ns" % (len(ds), *np.percentile(ds, [50, 90, 99, 99.9]), ds.max(), ds.mean()))
order = np.argsort(ds)[::-1][:20]
print("the 20 slowest: (duration ns, launch index, episode step of every env = launches since the reset kernel mod 192, gap to the previous launch ns, the two launches before / after)")
for o in order:
    i = idx[o]
    nb = [int(ds[j]) if 0 <= j < len(ds) else -1 for j in (o - 2, o - 1, o + 1, o + 2)]
    print("  %6.0f  #%-6d step %-4d gap %-10.0f neighbours %s" % (ds[o], o, phase[i], gap[o], nb))
known = phase[idx] >= 0
for name, sel in (("episode step 191 (the launch in which every env ends its episode and resets)", phase[idx] == 191),
                  ("episode step 0 (first step after the reset)", phase[idx] == 0),
                  ("first launch after a host gap > 20 us", gap > 20000),
                  ("all other launches", known & (phase[idx] != 191) & (phase[idx] != 0) & (gap <= 20000))):
    if sel.any():
        print("%-84s n %-6d mean %.0f p50 %.0f max %.0f ns" % (name, sel.sum(), ds[sel].mean(), np.median(ds[sel]), ds[sel].max()))
if known.any():
    by = np.array([ds[known & (phase[idx] == p)].mean() if (known & (phase[idx] == p)).any() else np.nan for p in range(192)])
    print("mean duration by episode step, steps 0..191 in rows of 16 (ns):")
    for r in range(0, 192, 16):
        print("  " + " ".join("%6.0f" % v for v in by[r:r + 16]))
    print("what the reset launch adds to the mean launch: %.0f ns" % ((by[191] - np.nanmedian(by)) / 192.0))
PY
rm -rf $OUT
