# usage (GPU box): CONFIGS="E:deg:launch:tape ..." bash tools/r06_burst.sh <tag> -- tools/stamps_burst.py for every ab_stamps/*.so
cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out/r06
cp fleetrl_amd/libfleet_hip.so /tmp/keep3.so; cp fleetrl_amd/libfleet_hip.gfx950.hsaco /tmp/keep3.hsaco
trap "cp /tmp/keep3.so fleetrl_amd/libfleet_hip.so; cp /tmp/keep3.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco" EXIT
for f in ab_stamps/*.so; do
  cp $f fleetrl_amd/libfleet_hip.so; cp ${f%.so}.gfx950.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco
  TAG=$(basename $f .so) timeout 900 python3 tools/stamps_burst.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/${TAG}_burst.log
done
cat gpurun_out/r06/${TAG}_burst.log
