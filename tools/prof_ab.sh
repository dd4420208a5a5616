# usage (GPU box): bash tools/prof_ab.sh "<variants>" [bench args] -- SQ counter medians + kernel duration per ab_variants/<v>.so
cd $GRAFT_REPO_ROOT
VS=$1; shift
cp fleetrl_amd/libfleet_hip.so /tmp/libfleet_hip.keep.so
for v in $VS; do
  cp ab_variants/$v.so fleetrl_amd/libfleet_hip.so
  echo "=== $v $@"
  bash tools/prof_step.sh ab_$v "$@" 2>&1 | grep -v "amdgpu.ids" | grep -E "fleet_step|calls=|SQ_|GRBM" | grep -v "Li1ELb1\|true" 
done
cp /tmp/libfleet_hip.keep.so fleetrl_amd/libfleet_hip.so
