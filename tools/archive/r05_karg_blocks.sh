cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
cp fleetrl_amd/libfleet_hip.so /tmp/k.so; cp fleetrl_amd/libfleet_hip.gfx950.hsaco /tmp/k.hsaco
for rep in 1 2; do for V in a_tree b_karg40; do cp ab_variants/$V.so fleetrl_amd/libfleet_hip.so; cp ab_variants/$V.gfx950.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco
for L in 1 2 40; do
python3 bench.py --tape-len $L --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V tape $L ms/step %.4f kernel_ms %.4f frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done; done | tee gpurun_out/r05/karg_blocks.log
cp /tmp/k.so fleetrl_amd/libfleet_hip.so; cp /tmp/k.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco
