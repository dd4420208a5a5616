# usage (GPU box): bash tools/r05_stamps.sh <tag> [E ...] -- in-kernel timeline of every ab_stamps/*.so (-DFLEET_STAMPS builds, never the
# product library) incl. WHERE each wavefront ran (die / CU / SIMD): tools/stamps.py
cd $GRAFT_REPO_ROOT
TAG=$1; shift
SIZES=${@:-4096}
mkdir -p gpurun_out/r05
cp fleetrl_amd/libfleet_hip.so /tmp/keep2.so; cp fleetrl_amd/libfleet_hip.gfx950.hsaco /tmp/keep2.hsaco
trap "cp /tmp/keep2.so fleetrl_amd/libfleet_hip.so; cp /tmp/keep2.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco" EXIT
for f in ab_stamps/*.so; do
  cp $f fleetrl_amd/libfleet_hip.so; cp ${f%.so}.gfx950.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco
  for E in $SIZES; do
    for LAUNCH in ${LAUNCHES:-0 2}; do  # 0: hipLaunchKernel per step, 2: the library's own queue
      echo "==== $(basename $f .so) E=$E launch mode $LAUNCH" >> gpurun_out/r05/${TAG}_stamps.log
      LAUNCH=$LAUNCH E=$E STEPS=20011 timeout 300 python3 tools/stamps.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05/${TAG}_stamps.log
    done
  done
done
cp /tmp/keep2.so fleetrl_amd/libfleet_hip.so; cp /tmp/keep2.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco
cat gpurun_out/r05/${TAG}_stamps.log | cut -c1-600
