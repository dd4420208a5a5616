# usage (GPU box): bash tools/r03_ab.sh <tag> [rounds] [bench args] -- every ab_variants/*.so back to back on this device
cd $GRAFT_REPO_ROOT
TAG=${1:-ab}; shift
mkdir -p gpurun_out/r03
bash tools/ab_run.sh "$@" > gpurun_out/r03/${TAG}.log 2>&1
grep -v "Traceback\|File\|json\|raise\|^ *\^" gpurun_out/r03/${TAG}.log | tail -40
