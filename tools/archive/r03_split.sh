# usage (GPU box): bash tools/r03_split.sh -- the default workload as 1 / 2 / 4 handles whose launches overlap (bench.py --split)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for shape in "" "--envs-per-gpu 16384" "--config c5"; do
for s in 1 2 4; do
  python3 bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-host-path --split $s $shape 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('split $s $shape  ms/step %.4f value %.3e kernel_ms %.4f' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"
done; done 2>&1 | tee gpurun_out/r03/split.log
