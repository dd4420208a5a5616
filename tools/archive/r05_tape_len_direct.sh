# usage (GPU box): bash tools/r05_tape_len_direct.sh -- does the length of the action tape matter once a die's L2 can keep state between launches?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2; do for A in "--config c4" "--envs-per-gpu 2048" "--envs-per-gpu 1024" "" ; do for L in 2 8 32 64; do
python3 bench.py --tape-len $L $A --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tape $L $A ms/step %.4f kernel_ms %.4f frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done; done | tee gpurun_out/r05/tape_len_direct.log
