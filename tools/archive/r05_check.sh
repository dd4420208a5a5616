# usage (GPU box): bash tools/r05_check.sh <tag> [pytest args] -- the full -m gpu suite + one default bench line of the library in the tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
TAG=${1:-check}; shift
python3 -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15 | tee gpurun_out/r05/pytest_${TAG}.log
python3 bench.py --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | tee gpurun_out/r05/bench_${TAG}.json
