# usage (GPU box): bash tools/r03_shapes.sh <tag> -- kernel time per step of the product library at every BASELINE shape
cd $GRAFT_REPO_ROOT
TAG=${1:-shapes}
mkdir -p gpurun_out/r03
run() { python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-host-path "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s ms/step %.4f kernel_ms %.4f frac %.3f many %.3e' % (sys.argv[1], d['ms_per_step'], r['kernel_ms'], r['frac'], d['step_many']['env_steps_per_s']))" "$*"; }
{
run --config c3
run --config c2
run --config c4
run --config c5
run --envs-per-gpu 16384
run --envs-per-gpu 8192
run --envs-per-gpu 1024
} | tee gpurun_out/r03/${TAG}.log
