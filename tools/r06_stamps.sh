# usage (GPU box): [LAUNCHES="0 2"] [DEGS="rainflow none"] bash tools/r06_stamps.sh <tag> [E ...] -- in-kernel timeline of every ab_stamps/*.so
# (-DFLEET_STAMPS builds, never the product library): tools/stamps.py
cd $GRAFT_REPO_ROOT
TAG=$1; shift
SIZES=${@:-4096}
mkdir -p gpurun_out/r06
cp fleetrl_amd/libfleet_hip.so /tmp/keep2.so; cp fleetrl_amd/libfleet_hip.gfx950.hsaco /tmp/keep2.hsaco
trap "cp /tmp/keep2.so fleetrl_amd/libfleet_hip.so; cp /tmp/keep2.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco" EXIT
for f in ab_stamps/*.so; do
  cp $f fleetrl_amd/libfleet_hip.so; cp ${f%.so}.gfx950.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco
  for E in $SIZES; do
    for D in ${DEGS:-rainflow}; do
    for LAUNCH in ${LAUNCHES:-0 2}; do  # 0: hipLaunchKernel per step, 2: the library's own queue
      echo "==== $(basename $f .so) E=$E deg $D launch mode $LAUNCH" >> gpurun_out/r06/${TAG}_stamps.log
      DEG=$D LAUNCH=$LAUNCH E=$E STEPS=${STEPS:-20011} timeout 300 python3 tools/stamps.py 2>&1 | grep -v amdgpu.ids | head -30 >> gpurun_out/r06/${TAG}_stamps.log
    done
    done
  done
done
cat gpurun_out/r06/${TAG}_stamps.log | cut -c1-400
