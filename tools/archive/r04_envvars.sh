# usage (GPU box): bash tools/r04_envvars.sh -- does a runtime switch change the launch floor? the default bench command (c3, graph replay
# of 2000 steps) and the eager 20-step run under a few HIP / ROCr environment settings, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
run() { env "$@" python3 bench.py --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-44s graph  ms/step %.4f kernel_ms %.4f' % (sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms']))" "$*"
env "$@" python3 bench.py --no-cpu-baseline --no-host-path --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-44s eager  ms/step %.4f kernel_ms %.4f' % (sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms']))" "$*"; }
{
run FLEET_NOP=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=1
run HSA_ENABLE_SDMA=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run FLEET_NOP=1
} 2>&1 | tee gpurun_out/r04/envvars.log
