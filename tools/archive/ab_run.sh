# usage (GPU box): bash tools/ab_run.sh [rounds] [bench args]  -- see tools/ab_build.sh
cd $GRAFT_REPO_ROOT
ROUNDS=${1:-2}; shift
cp fleetrl_amd/libfleet_hip.so /tmp/libfleet_hip.keep.so
trap "cp /tmp/libfleet_hip.keep.so fleetrl_amd/libfleet_hip.so" EXIT
run() { python3 bench.py --steps 2000 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.4f kernel_ms %.4f many %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['step_many']['env_steps_per_s']))"; }
for r in $(seq $ROUNDS); do
  for f in ab_variants/*.so; do
    cp $f fleetrl_amd/libfleet_hip.so; echo "== $(basename $f .so)"; run "$@"
  done
done
cp /tmp/libfleet_hip.keep.so fleetrl_amd/libfleet_hip.so
