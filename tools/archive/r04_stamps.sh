# usage (GPU box): bash tools/r04_stamps.sh <tag> [E ...] -- in-kernel timeline of every ab_stamps/*.so (-DFLEET_STAMPS builds, never the
# product library) at the given batch sizes (default 4096): where a wavefront spends its cycles, who ends the launch, by class
cd $GRAFT_REPO_ROOT
TAG=$1; shift
SIZES=${@:-4096}
mkdir -p gpurun_out/r04
cp fleetrl_amd/libfleet_hip.so /tmp/keep2.so
trap "cp /tmp/keep2.so fleetrl_amd/libfleet_hip.so" EXIT
for f in ab_stamps/*.so; do
  cp $f fleetrl_amd/libfleet_hip.so
  for E in $SIZES; do
    echo "==== $(basename $f .so) E=$E" >> gpurun_out/r04/${TAG}_stamps.log
    E=$E STEPS=20011 timeout 300 python3 tools/stamps.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/${TAG}_stamps.log
  done
done
cp /tmp/keep2.so fleetrl_amd/libfleet_hip.so
cat gpurun_out/r04/${TAG}_stamps.log | cut -c1-400
