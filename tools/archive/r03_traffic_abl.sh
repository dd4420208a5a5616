# usage (GPU box): bash tools/r03_traffic_abl.sh [bench args] -- HBM traffic (tools/prof_traffic.sh) of every ab_variants/*.so:
# a measured breakdown of the step kernel's reads and writes by what the ablation build leaves out
cd $GRAFT_REPO_ROOT
export FLEET_BENCH_NO_ERRCHECK=1
cp fleetrl_amd/libfleet_hip.so /tmp/libfleet_hip.keep.so
trap "cp /tmp/libfleet_hip.keep.so fleetrl_amd/libfleet_hip.so" EXIT
mkdir -p gpurun_out/r03
for f in ab_variants/*.so; do
  tag=$(basename $f .so)
  cp $f fleetrl_amd/libfleet_hip.so
  bash tools/prof_traffic.sh abl_$tag --no-host-path "$@" | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['envs']
print('$tag  read %.0f B/env-step  write %.0f B/env-step  l2 hit %.3f' % (d['hbm_read_bytes_per_launch']/e, d['hbm_write_bytes_per_launch']/e, d['l2_hit_rate']))"
done 2>&1 | tee gpurun_out/r03/traffic_abl.log
rm -rf gpurun_out/prof
