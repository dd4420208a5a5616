# usage (GPU box): bash tools/r06_final.sh -- what the round's committed evidence is made of, in one lease: the GPU test summary, every
# profile of tools/r06_profiles.sh, and -- once the traffic profiles of THIS kernel source sit in profiles/ -- the bench lines that
# carry roofline.traffic.  Outputs: gpurun_out/r06/gpu_tests.txt, gpurun_out/r06p/*.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1200 python3 -m pytest tests -m gpu -q -p no:cacheprovider > /tmp/pytest_all.log 2>&1
{ echo "kernel sources sha256[:16] $(python3 -c 'import bench; print(bench.kernel_source_sha())')"; grep -E "passed|failed|error" /tmp/pytest_all.log | tail -3; grep -E "^(FAILED|ERROR)" /tmp/pytest_all.log | head -20; grep -A3 "warnings summary" /tmp/pytest_all.log | cut -c1-600; } > gpurun_out/r06/gpu_tests.txt
timeout 300 python3 -m pytest tests/test_hip_parity.py -m gpu -q -s -p no:cacheprovider 2>/dev/null | grep "identical" > gpurun_out/r06/obs_words_identical.txt
bash tools/r06_profiles.sh > gpurun_out/r06/profiles.log 2>&1
cp gpurun_out/r06p/r06_traffic_*.json profiles/
OUT=gpurun_out/r06p
timeout 300 python3 bench.py > $OUT/r06_bench_default.json 2> $OUT/bench2.err
timeout 300 python3 bench.py --steps 20 --warmup 5 > $OUT/r06_bench_default_20steps.json 2>> $OUT/bench2.err
timeout 300 python3 bench.py --config c5 --no-cpu-baseline > $OUT/r06_bench_c5.json 2>> $OUT/bench2.err
timeout 300 python3 bench.py --envs-per-gpu 16384 --no-cpu-baseline --no-host-path > $OUT/r06_bench_16384x50.json 2>> $OUT/bench2.err
for f in default default_20steps c5 16384x50; do python3 -c "
import json; d=json.loads(open('$OUT/r06_bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f frac %.3f'%(r['kernel_ms'],r['frac']), 'traffic', r['traffic'], 'many %.3e'%d['step_many']['env_steps_per_s'])"; done
cat gpurun_out/r06/gpu_tests.txt; head -3 gpurun_out/r06/obs_words_identical.txt
bash tools/launch_by_episode_step.sh > /dev/null 2>&1; cp gpurun_out/r06/launch_by_episode_step.log $OUT/launch_by_episode_step.log; head -3 $OUT/launch_by_episode_step.log
