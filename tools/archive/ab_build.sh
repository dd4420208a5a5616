# usage (build container): bash tools/ab_build.sh tag1="<flags>" tag2="<flags>" ...
# Cross-compiles one libfleet_hip.so per flag set into ab_variants/ (git-ignored, travels with gpurun); then on the GPU
# box `bash tools/ab_run.sh [bench args]` times each of them back to back on the same device (box-to-box variance is
# larger than most effects worth measuring).
cd "$(dirname "$0")/.." && mkdir -p ab_variants && rm -f ab_variants/*.so
for spec in "$@"; do
  tag="${spec%%=*}"; flags="${spec#*=}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=12 -shared $flags \
    fleetrl_amd/csrc/fleet_kernels.hip fleetrl_amd/csrc/fleet_capi.hip -o ab_variants/$tag.so &
done
wait; ls -la ab_variants
