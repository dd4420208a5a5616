# usage (GPU box): bash tools/r05_final.sh -- what the round's committed evidence is made of, in one lease: the GPU test summary, every
# profile of tools/r05_profiles.sh, and -- once the traffic profiles of THIS kernel source sit in profiles/ -- the bench lines that
# carry roofline.traffic (tools/r05_benchlines.sh).  Outputs: gpurun_out/r05/gpu_tests.txt, gpurun_out/r05p/*.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
python3 -m pytest tests -m gpu -q -p no:cacheprovider > /tmp/pytest_all.log 2>&1
{ echo "kernel sources sha256[:16] $(python3 -c 'import bench; print(bench.kernel_source_sha())')"; grep -E "passed|failed|error" /tmp/pytest_all.log | tail -3; grep -E "^(FAILED|ERROR)" /tmp/pytest_all.log | head -20; } > gpurun_out/r05/gpu_tests.txt
python3 -m pytest tests/test_hip_parity.py -m gpu -q -s -p no:cacheprovider 2>/dev/null | grep "identical" > gpurun_out/r05/obs_words_identical.txt
bash tools/r05_profiles.sh > gpurun_out/r05/profiles.log 2>&1
cp gpurun_out/r05p/r05_traffic_*.json profiles/
bash tools/r05_benchlines.sh > gpurun_out/r05/benchlines.log 2>&1
cat gpurun_out/r05/gpu_tests.txt; tail -5 gpurun_out/r05/benchlines.log; head -3 gpurun_out/r05/obs_words_identical.txt
