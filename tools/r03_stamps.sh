# usage (GPU box): bash tools/r03_stamps.sh <tag> -- A/B of ab_variants/*.so (except stamps.so), then the stamps timeline
cd $GRAFT_REPO_ROOT
TAG=${1:-s}
mkdir -p gpurun_out/r03
bash tools/ab_run.sh 2 --no-host-path > gpurun_out/r03/${TAG}_ab.log 2>&1
cp fleetrl_amd/libfleet_hip.so /tmp/keep2.so
cp ab_variants/stamps.so fleetrl_amd/libfleet_hip.so
STEPS=20011 python3 tools/stamps.py > gpurun_out/r03/${TAG}_stamps.log 2>&1
cp /tmp/keep2.so fleetrl_amd/libfleet_hip.so
grep -v Traceback gpurun_out/r03/${TAG}_ab.log | tail -30; tail -40 gpurun_out/r03/${TAG}_stamps.log
