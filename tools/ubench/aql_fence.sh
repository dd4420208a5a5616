# usage (GPU box): bash tools/ubench/aql_fence.sh [workgroups words launches]  -- see aql_fence.cpp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
/opt/rocm/bin/hipcc --genco --no-gpu-bundle-output --offload-arch=gfx950 -O3 tools/ubench/aql_fence_kernels.hip -o /tmp/aql_fence_kernels.hsaco || exit 1
/opt/rocm/bin/hipcc -O2 -std=c++17 tools/ubench/aql_fence.cpp -o /tmp/aql_fence -L/opt/rocm/lib -lhsa-runtime64 || exit 1
mkdir -p gpurun_out/r05
for shape in "1024 512" "4096 512"; do
  timeout 120 /tmp/aql_fence /tmp/aql_fence_kernels.hsaco $shape ${3:-2000}
done 2>&1 | tee gpurun_out/r05/aql_fence.log
