# usage (GPU box): bash tools/ubench/aql_capacity.sh -- see aql_capacity.cpp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
/opt/rocm/bin/hipcc --genco --no-gpu-bundle-output --offload-arch=gfx950 -O3 tools/ubench/aql_fence_kernels.hip -o /tmp/aql_fence_kernels.hsaco || exit 1
/opt/rocm/bin/hipcc -O2 -std=c++17 tools/ubench/aql_capacity.cpp -o /tmp/aql_capacity -L/opt/rocm/lib -lhsa-runtime64 || exit 1
mkdir -p gpurun_out/r05
timeout 200 /tmp/aql_capacity /tmp/aql_fence_kernels.hsaco 2>&1 | tee gpurun_out/r05/aql_capacity.log
