# usage (GPU box): bash tools/r05_direct.sh -- FLEET_LAUNCH_DIRECT against the hipGraph replay: parity tests, then bench lines of both
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_direct_gpu.py -q -m gpu -x 2>&1 | tail -15 > $OUT/direct_tests.txt
cat $OUT/direct_tests.txt
line() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('%-34s'%'$2', 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f frac %.3f'%(r['kernel_ms'],r['frac']), 'value %.3e'%d['value'])" 2>&1 | tail -1; }
for rep in 1 2; do
for L in graph direct; do
  for A in "" "--steps 20 --warmup 5" "--envs-per-gpu 2048" "--envs-per-gpu 8192" "--envs-per-gpu 16384" "--config c4" "--config c5"; do
    timeout 300 python3 bench.py --launch $L $A --no-cpu-baseline --no-host-path > $OUT/direct_tmp.json 2> $OUT/direct_tmp.err || tail -5 $OUT/direct_tmp.err
    line $OUT/direct_tmp.json "$L $A"
  done
done
done 2>&1 | tee $OUT/direct_vs_graph.log
