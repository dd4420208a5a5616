# usage (GPU box): [NOERR=1] [STEPS=1000] bash tools/r06_ab.sh <tag> <rounds> "<bench args>" ["<bench args>" ...]
# Every ab_variants/*.so (tools/build_variants.sh) at every argument set, back to back on this device, `rounds` times
# (box-to-box variance is larger than most effects).  One line per run: kernel time from the dispatch timestamps / HIP events.
cd $GRAFT_REPO_ROOT
TAG=${1:-ab}; ROUNDS=${2:-1}; shift; shift
mkdir -p gpurun_out/r06
cp fleetrl_amd/libfleet_hip.so /tmp/keep6.so; cp fleetrl_amd/libfleet_hip.gfx950.hsaco /tmp/keep6.hsaco
trap "cp /tmp/keep6.so fleetrl_amd/libfleet_hip.so; cp /tmp/keep6.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco" EXIT
run() { FLEET_BENCH_NO_ERRCHECK=${NOERR:-0} timeout 90 python3 bench.py --steps ${STEPS:-1000} --warmup 100 --no-cpu-baseline --no-host-path $2 2>/tmp/r06_err.log | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
    print('%-16s %-44s ms/step %.4f kernel_ms %.5f frac %.3f many %.3e %s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['kernel_ms'], r['frac'], d['step_many']['env_steps_per_s'], c['launch_mode']))
except Exception as e:
    print('%-16s %-44s FAILED %s' % (sys.argv[1], sys.argv[2], e)); print(open('/tmp/r06_err.log').read()[-600:])
" "$1" "$2"; }
{
for R in $(seq $ROUNDS); do
for V in $(ls ab_variants | grep '\.so$' | sed 's/.so//'); do
  cp ab_variants/$V.so fleetrl_amd/libfleet_hip.so
  rm -f fleetrl_amd/libfleet_hip.gfx950.hsaco; [ -f ab_variants/$V.gfx950.hsaco ] && cp ab_variants/$V.gfx950.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco
  for A in "$@"; do run $V "$A"; done
done
done
} | tee gpurun_out/r06/${TAG}.log
