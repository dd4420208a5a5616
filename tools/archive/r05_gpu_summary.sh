# usage (GPU box): bash tools/r05_gpu_summary.sh -- the GPU test summary as tools/r05_final.sh writes it (gpurun_out/r05/gpu_tests.txt), alone
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
python3 -m pytest tests -m gpu -q -p no:cacheprovider > /tmp/pytest_all.log 2>&1
{ echo "kernel sources sha256[:16] $(python3 -c 'import bench; print(bench.kernel_source_sha())')"; grep -E "passed|failed|error" /tmp/pytest_all.log | tail -3; grep -E "^(FAILED|ERROR)" /tmp/pytest_all.log | head -20; } > gpurun_out/r05/gpu_tests.txt
cat gpurun_out/r05/gpu_tests.txt
