# usage (GPU box): bash tools/r04_split.sh -- the batch as 1 / 2 / 3 / 4 handles on streams of their own (bench.py --split) at 4096,
# 8192, 16384 and 32768 envs x 50 EVs: does overlapping one part's loads with another part's arithmetic and stores pay?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for R in 1 2; do
for E in 4096 8192 16384 32768; do
for s in 1 2 3 4; do
  python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-host-path --split $s --envs-per-gpu $E 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('E $E split $s  ms/step %.4f value %.3e kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done; done 2>&1 | tee gpurun_out/r04/split.log
