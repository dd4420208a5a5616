# usage (GPU box): bash tools/ubench/xcc_map.sh -- see xcc_map.cpp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
/opt/rocm/bin/hipcc --genco --no-gpu-bundle-output --offload-arch=gfx950 -O3 tools/ubench/xcc_map_kernels.hip -o /tmp/xcc_map_kernels.hsaco || exit 1
/opt/rocm/bin/hipcc -O2 -std=c++17 tools/ubench/xcc_map.cpp -o /tmp/xcc_map -L/opt/rocm/lib -lhsa-runtime64 || exit 1
mkdir -p gpurun_out/r06
timeout 120 /tmp/xcc_map /tmp/xcc_map_kernels.hsaco 2>&1 | tee gpurun_out/r06/xcc_map.log
