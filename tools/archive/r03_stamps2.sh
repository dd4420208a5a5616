# usage (GPU box): bash tools/r03_stamps2.sh <tag> <variant.so> ... -- stamps timeline of each listed ab_variants/<v>.so
cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/r03
cp fleetrl_amd/libfleet_hip.so /tmp/keep2.so
trap "cp /tmp/keep2.so fleetrl_amd/libfleet_hip.so" EXIT
for v in "$@"; do
  cp ab_variants/$v.so fleetrl_amd/libfleet_hip.so
  echo "==== $v" >> gpurun_out/r03/${TAG}_stamps.log
  STEPS=20011 timeout 120 python3 tools/stamps.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03/${TAG}_stamps.log
done
cp /tmp/keep2.so fleetrl_amd/libfleet_hip.so
grep -A13 "====" gpurun_out/r03/${TAG}_stamps.log | grep -v "^--"
grep "launch timeline" gpurun_out/r03/${TAG}_stamps.log
