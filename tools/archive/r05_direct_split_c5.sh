cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2; do for S in 1 2; do for A in "--config c5" "--config c4" "--envs-per-gpu 6144" "--envs-per-gpu 8192"; do
python3 bench.py --launch direct --split $S $A --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split $S $A ms/step %.4f frac %.3f value %.3e'%(d['ms_per_step'], d['roofline']['frac'], d['value']))"
done; done; done | tee gpurun_out/r05/direct_split_c5.log
