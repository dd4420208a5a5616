cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_direct_gpu.py -q -m gpu -x 2>&1 | tail -4
for rep in 1 2; do for L in direct direct1; do for A in "" "--envs-per-gpu 6144" "--envs-per-gpu 8192" "--envs-per-gpu 16384"; do
python3 bench.py --launch $L $A --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L $A ms/step %.4f kernel_ms %.4f frac %.3f value %.3e'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['value']), d['config']['launch'][:60])"
done; done; done | tee gpurun_out/r05/direct_two_queues_in_library.log
