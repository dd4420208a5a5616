# usage (GPU box): bash tools/r02_attrib.sh  -- stall attribution of the per-step launch (round 2): ablation builds from
# tools/ab_build.sh timed back to back on one device at 4096x50 and 16384x50, stamps, rare-row profile.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LOG=gpurun_out/r02_attrib.log
: > $LOG
cp fleetrl_amd/libfleet_hip.so /tmp/libfleet_hip.keep.so
run() { python3 bench.py --steps 1500 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.4f kernel_ms %.4f pair_ms %.4f many %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_event_pair_per_launch'], d['step_many']['env_steps_per_s']))"; }
for E in 4096 16384; do
  for f in base empty noobs nopush tablocal norare noreduce base; do
    cp ab_variants/$f.so fleetrl_amd/libfleet_hip.so; echo "== E=$E $f" >> $LOG; run --envs-per-gpu $E >> $LOG 2>&1
  done
  cp ab_variants/base.so fleetrl_amd/libfleet_hip.so
  for deg in none linear; do echo "== E=$E base deg=$deg" >> $LOG; run --envs-per-gpu $E --deg $deg >> $LOG 2>&1; done
done
cp ab_variants/stamps.so fleetrl_amd/libfleet_hip.so
for E in 4096 16384; do echo "== stamps E=$E" >> $LOG; E=$E python3 tools/stamps.py >> $LOG 2>&1; done
cp ab_variants/base.so fleetrl_amd/libfleet_hip.so
for E in 4096 16384; do echo "== step_profile E=$E" >> $LOG; E=$E python3 tools/step_profile.py >> $LOG 2>&1; done
cp /tmp/libfleet_hip.keep.so fleetrl_amd/libfleet_hip.so
cat $LOG
