# usage (build container): bash tools/build_variants.sh tag1="<src>|<flags>" ...   src = "." (working tree) or a git revision
# Cross-compiles one libfleet_hip.so per entry into ab_variants/ (git-ignored, travels with gpurun); a git revision is exported to
# a temporary directory first, so that the round-4 kernel can run beside the tree's on the same box (tools/r05_ab.sh).
cd "$(dirname "$0")/.." && mkdir -p ab_variants && rm -f ab_variants/*.so ab_variants/*.hsaco
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=12 -mllvm -amdgpu-sched-strategy=max-memory-clause"
for spec in "$@"; do
  tag="${spec%%=*}"; rest="${spec#*=}"; src="${rest%%|*}"; flags="${rest#*|}"; [ "$flags" = "$rest" ] && flags=""
  if [ "$src" = "." ]; then dir=.; else dir=$(mktemp -d /tmp/absrc.XXXX); git archive "$src" fleetrl_amd/csrc include | tar -x -C "$dir"; sed -i "s/#define FLEET_ABI_VERSION .*/$(grep '#define FLEET_ABI_VERSION' include/fleet_hip.h)/" "$dir/include/fleet_hip.h"; fi  # (an older revision answers to the tree's ABI number: the public structures have not changed since version 4)
  sha="-DFLEET_SRC_SHA=\"v_$(cat $dir/fleetrl_amd/csrc/* | sha256sum | cut -c1-12)_$(echo "$flags" | sha256sum | cut -c1-8)\""; flags="$flags $sha"
  if [ -f "$dir/fleetrl_amd/csrc/fleet_direct.hip" ]; then  # sources with the library's own launch queue: the kernels' code object beside the library
    ( /opt/rocm/bin/hipcc $FLAGS -shared $flags "$dir/fleetrl_amd/csrc/fleet_kernels.hip" "$dir/fleetrl_amd/csrc/fleet_capi.hip" "$dir/fleetrl_amd/csrc/fleet_direct.hip" -L/opt/rocm/lib -lhsa-runtime64 -o ab_variants/$tag.so || echo "BUILD FAILED: $tag" ) &
    ( /opt/rocm/bin/hipcc --genco --no-gpu-bundle-output ${FLAGS/-fPIC/} $flags "$dir/fleetrl_amd/csrc/fleet_kernels.hip" -o ab_variants/$tag.gfx950.hsaco || echo "BUILD FAILED: $tag (code object)" ) &
  else
    ( /opt/rocm/bin/hipcc $FLAGS -shared $flags "$dir/fleetrl_amd/csrc/fleet_kernels.hip" "$dir/fleetrl_amd/csrc/fleet_capi.hip" -o ab_variants/$tag.so || echo "BUILD FAILED: $tag" ) &
  fi
done
wait; ls -la ab_variants
