# usage (GPU box): bash tools/r03_valu_abl.sh -- instructions per wavefront and step of every ab_variants/*.so (SQ counters):
# how many instructions each part of the step executes, by what the ablation build leaves out
cd $GRAFT_REPO_ROOT
export FLEET_BENCH_NO_ERRCHECK=1
cp fleetrl_amd/libfleet_hip.so /tmp/libfleet_hip.keep.so
trap "cp /tmp/libfleet_hip.keep.so fleetrl_amd/libfleet_hip.so" EXIT
mkdir -p gpurun_out/r03
for f in ab_variants/*.so; do
  tag=$(basename $f .so)
  cp $f fleetrl_amd/libfleet_hip.so
  bash tools/prof_counters.sh valu_$tag "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "$@" | python3 -c "
import sys
v={}
for l in sys.stdin:
    p=l.split()
    if len(p)==2:
        try: v[p[0]]=float(p[1])
        except ValueError: pass
w=v.get('SQ_WAVES',1)
print('$tag  per wave: VALU %.0f  SALU %.0f  branch %.0f  VALU-active cycles %.0f  wave cycles %.0f' % (v.get('SQ_INSTS_VALU',0)/w, v.get('SQ_INSTS_SALU',0)/w, v.get('SQ_INSTS_BRANCH',0)/w, 4*v.get('SQ_ACTIVE_INST_VALU',0)/w, 4*v.get('SQ_WAVE_CYCLES',0)/w))"
done 2>&1 | tee gpurun_out/r03/valu_abl.log
rm -rf gpurun_out/prof
