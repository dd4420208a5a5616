# usage (GPU box): [NOERR=1] [LAUNCH=direct|graph] bash tools/r05_ab.sh <tag> [rounds] [shape ...] -- every ab_variants/*.so at several batch shapes, back to
# back on this device (box-to-box variance is larger than most effects).  Shapes: c2 c3 c4 c5 (bench configs), <envs> (c3 with
# that many envs), <envs>x<evs> (c3's fleet with another geometry), c5:<envs>x<evs> (the mixed fleet with another geometry).
cd $GRAFT_REPO_ROOT
TAG=${1:-ab}; ROUNDS=${2:-1}; shift; shift
SHAPES=${@:-c3 16384 c5 c4}
mkdir -p gpurun_out/r05
cp fleetrl_amd/libfleet_hip.so /tmp/keep5.so; cp fleetrl_amd/libfleet_hip.gfx950.hsaco /tmp/keep5.hsaco
trap "cp /tmp/keep5.so fleetrl_amd/libfleet_hip.so; cp /tmp/keep5.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco" EXIT
run() { FLEET_BENCH_NO_ERRCHECK=${NOERR:-0} python3 bench.py --launch ${LAUNCH:-direct} --steps ${STEPS:-1000} --warmup 100 --no-cpu-baseline --no-host-path "$@" 2>/tmp/r05_err.log | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
    evs=c.get('evs_per_env'); envs=c.get('envs_per_gpu')
    print('%-18s %-26s ms/step %.4f kernel_ms %.4f frac %.3f evsteps/s %.3e many %.3e' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']*(evs or 0), d['step_many']['env_steps_per_s']))
except Exception as e:
    print('%-18s %-26s FAILED %s' % (sys.argv[1], sys.argv[2], e)); print(open('/tmp/r05_err.log').read()[-600:])
" "$V" "$*"; }
{
for R in $(seq $ROUNDS); do
for V in $(ls ab_variants | grep '\.so$' | sed 's/.so//'); do
  cp ab_variants/$V.so fleetrl_amd/libfleet_hip.so
  rm -f fleetrl_amd/libfleet_hip.gfx950.hsaco; [ -f ab_variants/$V.gfx950.hsaco ] && cp ab_variants/$V.gfx950.hsaco fleetrl_amd/libfleet_hip.gfx950.hsaco  # (none: bench.py falls back to the hipGraph and says so)
  for S in $SHAPES; do
    case $S in
      c5:*) g=${S#c5:}; run --config c5 --envs-per-gpu ${g%x*} --evs ${g#*x};;
      c*) run --config $S;;
      *x*) run --envs-per-gpu ${S%x*} --evs ${S#*x};;
      *) run --envs-per-gpu $S;;
    esac
  done
done
done
} | tee gpurun_out/r05/${TAG}.log
