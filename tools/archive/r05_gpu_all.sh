cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r05/gpu_all.txt; tail -5 gpurun_out/r05/gpu_all.txt
