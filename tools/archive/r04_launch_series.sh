# usage (GPU box): bash tools/r04_launch_series.sh [bench args] -- per-launch durations of the single-step kernel in launch order
# (rocprofv3 --kernel-trace): percentiles, and what the slow launches have in common (position in the replayed graph, period)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/series; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/r04; cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline --no-host-path "$@" > $OUT/bench.log 2>&1
python3 - $OUT <<'PY' | tee $R/gpurun_out/r04/launch_series.log
import glob, sys
import numpy as np, pandas as pd
f = glob.glob(f"{sys.argv[1]}/kt/*/*kernel_trace.csv")[0]
c = pd.read_csv(f)
c = c[c.Kernel_Name.str.contains(r"fleet_step_kernel<\d+, \d+, false,", regex=True)].sort_values("Start_Timestamp")
d = (c.End_Timestamp - c.Start_Timestamp).to_numpy().astype(float)
gap = (c.Start_Timestamp.to_numpy()[1:] - c.End_Timestamp.to_numpy()[:-1]).astype(float)
print("launches", len(d), "duration ns: p10 %.0f p50 %.0f p90 %.0f p99 %.0f mean %.0f" % (*np.percentile(d, [10, 50, 90, 99]), d.mean()))
print("gap to the next launch ns: p10 %.0f p50 %.0f p90 %.0f" % tuple(np.percentile(gap, [10, 50, 90])))
# steady part only (the long back-to-back runs): launches whose gap to the previous one is small
steady = np.r_[False, gap < 3000]
ds = d[steady]
print("back-to-back launches", len(ds), "p10 %.0f p50 %.0f p90 %.0f p99 %.0f mean %.0f" % (*np.percentile(ds, [10, 50, 90, 99]), ds.mean()))
# periodicity: autocorrelation of the steady series at small lags and at the candidate periods (tape length, a day = 96 rows)
x = ds[: 1 << 15] - ds[: 1 << 15].mean()
ac = np.fft.irfft(np.abs(np.fft.rfft(x, 2 * len(x))) ** 2)[: len(x)]
ac /= ac[0]
print("autocorrelation at lags 1 2 3 4 8 16 32 64 96 128 192 256:", " ".join("%.2f" % ac[k] for k in (1, 2, 3, 4, 8, 16, 32, 64, 96, 128, 192, 256)))
top = np.argsort(ac[2:2000])[::-1][:8] + 2
print("strongest lags in 2..2000:", [(int(k), round(float(ac[k]), 2)) for k in top])
# runs of slow launches
slow = ds > np.percentile(ds, 90)
runs = np.diff(np.flatnonzero(np.diff(np.r_[0, slow.astype(int), 0])))[::2]
print("slow (> p90) launches come in runs of length: median %d p90 %d max %d (n runs %d)" % (np.median(runs), np.percentile(runs, 90), runs.max(), len(runs)))
idx = np.flatnonzero(np.diff(np.r_[0, slow.astype(int), 0]) == 1)
long_runs = [(int(i), int(n), float(ds[i:i + n].mean())) for i, n in zip(idx, runs) if n >= 4][:14]
print("runs of >= 4 slow launches (first index, length, mean ns):", long_runs)
st = c.Start_Timestamp.to_numpy()[steady.nonzero()[0]]
print("  their start times, ms from the first:", [round(float(st[i] - st[0]) / 1e6, 3) for i, _, _ in long_runs])
excess = (ds[slow] - np.median(ds)).sum() / len(ds)
print("the slowest 10 %% of the launches add %.0f ns to the mean; mean of the other 90 %%: %.0f" % (excess, ds[~slow].mean()))
k = 256
blocks = ds[: len(ds) // k * k].reshape(-1, k).mean(axis=1)
print("mean duration of consecutive blocks of %d launches: min %.0f p50 %.0f max %.0f" % (k, blocks.min(), np.median(blocks), blocks.max()))
PY
rm -rf $OUT
