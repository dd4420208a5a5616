# round 3, first call: baseline + store-policy variants back to back, then the stamps timeline of the product kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash tools/ab_run.sh 2 --no-host-path > gpurun_out/r03/ab_first.log 2>&1
cp fleetrl_amd/libfleet_hip.so /tmp/keep2.so
cp ab_variants/stamps.so fleetrl_amd/libfleet_hip.so
STEPS=20011 python3 tools/stamps.py > gpurun_out/r03/stamps_base.log 2>&1
cp /tmp/keep2.so fleetrl_amd/libfleet_hip.so
tail -30 gpurun_out/r03/ab_first.log; tail -40 gpurun_out/r03/stamps_base.log
