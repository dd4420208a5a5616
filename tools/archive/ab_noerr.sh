# usage (GPU box): bash tools/ab_noerr.sh <tag> [rounds] [bench args] -- like r03_ab.sh for ablation builds that break the
# bookkeeping on purpose: bench.py runs with FLEET_BENCH_NO_ERRCHECK=1 (no device error check), nothing is edited in place
cd $GRAFT_REPO_ROOT
TAG=${1:-abl}; shift
mkdir -p gpurun_out/r03
FLEET_BENCH_NO_ERRCHECK=1 bash tools/ab_run.sh "$@" > gpurun_out/r03/${TAG}.log 2>&1
grep -v "Traceback\|File\|json\|raise\|^ *\^" gpurun_out/r03/${TAG}.log | tail -60
