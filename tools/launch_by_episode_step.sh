# usage (GPU box): bash tools/launch_by_episode_step.sh [bench args] -- the single-step kernel's launch duration by position in the
# (phase-locked) 192-step episode: rocprofv3 --kernel-trace of `bench.py --lean`, the long back-to-back runs folded modulo the episode length
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/fold; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/r06; cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline --no-host-path --lean --steps 1920 --warmup 192 "$@" > $OUT/bench.log 2>&1
python3 - $OUT <<'PY' | tee $R/gpurun_out/r06/launch_by_episode_step.log
import glob, sys
import numpy as np, pandas as pd
c = pd.read_csv(glob.glob(f"{sys.argv[1]}/kt/*/*kernel_trace.csv")[0])
c = c[c.Kernel_Name.str.contains(r"fleet_step_kernel<\d+, \d+, false,", regex=True)].sort_values("Start_Timestamp")
d = (c.End_Timestamp - c.Start_Timestamp).to_numpy().astype(float)
gap = np.r_[1e9, (c.Start_Timestamp.to_numpy()[1:] - c.End_Timestamp.to_numpy()[:-1]).astype(float)]
# runs of launches (a gap above 200 us starts a new one: another phase of the bench); keep those of at least two episodes
starts = np.flatnonzero(gap > 200000)
ends = np.r_[starts[1:], len(d)]
P = 192
fold = np.zeros(P); cnt = np.zeros(P)
print("runs:", [(int(a), int(b - a)) for a, b in zip(starts, ends)][:12])
for a, b in zip(starts, ends):
    if b - a < 2 * P: continue
    x = d[a:b][: (b - a) // P * P].reshape(-1, P)
    m = x.mean(axis=0)
    sh = (P - 1) - int(np.argmax(m))          # the launch that resets every env is the episode's last step
    fold += np.roll(x, sh, axis=1).sum(axis=0); cnt += x.shape[0]
f = fold / np.maximum(cnt, 1)
print("launches folded:", int(cnt.sum()), " mean %.0f ns  median position %.0f ns" % (f.mean(), np.median(f)))
for k in range(0, P, 8):
    print("episode steps %3d..%3d  mean %6.0f ns   %s" % (k, k + 7, f[k:k + 8].mean(), " ".join("%5.0f" % v for v in f[k:k + 8])))
base = np.median(f)
print("excess over the median position, summed over the episode / 192: %.0f ns; of which the last step (reset) %.0f, the two daily rows %.0f" %
      ((f - base).sum() / P, (f[-1] - base) / P, (np.sort(f[:-1])[-2:] - base).sum() / P))
PY
tail -5 $OUT/bench.log | cut -c1-300; rm -rf $OUT
