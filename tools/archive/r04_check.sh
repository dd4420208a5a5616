# usage (GPU box): bash tools/r04_check.sh <tag>  -- full GPU parity suite + the default bench line of the product library
cd $GRAFT_REPO_ROOT
TAG=${1:-check}
mkdir -p gpurun_out/r04
timeout 1800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/${TAG}_pytest.log
tail -15 gpurun_out/r04/${TAG}_pytest.log
python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-host-path > gpurun_out/r04/${TAG}_bench.json 2> gpurun_out/r04/${TAG}_bench.err
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r04/${TAG}_bench.json').read().strip().splitlines()[-1])
print('ms/step %.4f kernel_ms %.4f frac %.3f many %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['step_many']['env_steps_per_s']))
" || tail -5 gpurun_out/r04/${TAG}_bench.err
