# usage (GPU box): bash tools/r05_c5_ab.sh -- the c5 shard through both launch paths, interleaved (the shape drifts within a lease)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2 3; do for L in direct graph; do
  python3 bench.py --launch $L --config c5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L c5 ms/step %.4f frac %.3f'%(d['ms_per_step'], d['roofline']['frac']))"
done; done | tee gpurun_out/r05/c5_direct_vs_graph_interleaved.log
