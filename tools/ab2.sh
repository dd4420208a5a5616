# usage (GPU box): bash tools/ab2.sh "<E list>" [extra bench args] -- every ab_variants/*.so at each batch size, twice
cd $GRAFT_REPO_ROOT
ES=${1:-"4096 16384"}; shift
cp fleetrl_amd/libfleet_hip.so /tmp/libfleet_hip.keep.so
run() { python3 bench.py --steps 1500 --warmup 100 --no-cpu-baseline "$@" 2>&1 | tail -1 | python3 -c "import sys,json
s=sys.stdin.read()
try:
    d=json.loads(s); print('   ms/step %.4f kernel_ms %.4f many %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['step_many']['env_steps_per_s']))
except Exception: print('   FAILED', s[-300:])"; }
for E in $ES; do for r in 1 2; do for f in ab_variants/*.so; do
  cp $f fleetrl_amd/libfleet_hip.so; echo "== E=$E $(basename $f .so)"; run --envs-per-gpu $E "$@"
done; done; done
cp /tmp/libfleet_hip.keep.so fleetrl_amd/libfleet_hip.so
