# usage (GPU box): bash tools/r05_benchlines.sh -- the bench lines that carry roofline.traffic, re-run AFTER the traffic
# profiles of this kernel source have been copied into profiles/ (tools/r05_profiles.sh makes those)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05p; mkdir -p $OUT
python3 bench.py > $OUT/r05_bench_default.json 2> $OUT/bench2.err
python3 bench.py --steps 20 --warmup 5 > $OUT/r05_bench_default_20steps.json 2>> $OUT/bench2.err
python3 bench.py --config c5 --no-cpu-baseline > $OUT/r05_bench_c5.json 2>> $OUT/bench2.err
python3 bench.py --envs-per-gpu 16384 --no-cpu-baseline --no-host-path > $OUT/r05_bench_16384x50.json 2>> $OUT/bench2.err
for f in default default_20steps c5 16384x50; do python3 -c "
import json; d=json.loads(open('$OUT/r05_bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f frac %.3f'%(r['kernel_ms'],r['frac']), 'traffic', r['traffic'], 'many %.3e'%d['step_many']['env_steps_per_s'])"; done
