# usage: bash tools/ablate.sh "<flags1>" "<flags2>" ...  (on the GPU box): rebuild with each flag set and time the step
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.4f kernel_ms %.4f many %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['step_many']['env_steps_per_s']))"; }
for f in "$@"; do
  echo "== flags: [$f]"; FLEET_EXTRA_HIPCC_FLAGS="$f" python3 -c "from fleetrl_amd import build; build.build(force=True)"; run
done
FLEET_EXTRA_HIPCC_FLAGS="" python3 -c "from fleetrl_amd import build; build.build(force=True)"
