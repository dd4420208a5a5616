# usage (GPU box): bash tools/r05_tape_sweep.sh -- c3 through the library's queue with 1 ... 40 rows of action tape: where do the rows live?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2; do for L in 1 2 3 4 6 8 12 16 24 40; do
python3 bench.py --tape-len $L --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tape $L ms/step %.4f kernel_ms %.4f frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done | tee gpurun_out/r05/tape_sweep_c3.log
