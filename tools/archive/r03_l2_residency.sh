# usage (GPU box): bash tools/r03_l2_residency.sh -- HBM-side traffic per env-step and L2 hit rate of the step kernel at batch sizes
# whose per-XCD footprint is a fraction of the 4 MB L2: does state survive in the L2 from one launch to the next?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for e in 512 1024 2048 4096; do
  bash tools/prof_traffic.sh l2res_$e --no-host-path --envs-per-gpu $e | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['envs']
print('envs %5d  read %.0f B/env-step  write %.0f B/env-step  l2 hit %.3f' % (e, d['hbm_read_bytes_per_launch']/e, d['hbm_write_bytes_per_launch']/e, d['l2_hit_rate']))"
done 2>&1 | tee gpurun_out/r03/l2_residency.log
rm -rf gpurun_out/prof
