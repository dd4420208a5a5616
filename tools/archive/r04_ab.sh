# usage (GPU box): [CADS="2 8"] bash tools/r04_ab.sh <tag> [rounds] [shape ...] -- every ab_variants/*.so at several batch shapes, back to
# back on this device (box-to-box variance is larger than most effects); shapes: c3 8192 16384 c5 c4 c2 (default: c3 16384 c5 c4);
# CADS: values of FLEET_RF_CADENCE to run each library with (default: the library's own choice)
cd $GRAFT_REPO_ROOT
TAG=${1:-ab}; ROUNDS=${2:-1}; shift; shift
SHAPES=${@:-c3 16384 c5 c4}
mkdir -p gpurun_out/r04
cp fleetrl_amd/libfleet_hip.so /tmp/keep4.so
trap "cp /tmp/keep4.so fleetrl_amd/libfleet_hip.so" EXIT
run() { FLEET_BENCH_NO_ERRCHECK=${NOERR:-0} python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-host-path "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-16s %-24s ms/step %.4f kernel_ms %.4f frac %.3f many %.3e' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['kernel_ms'], r['frac'], d['step_many']['env_steps_per_s']))" "$V$CTAG" "$*"; }
{
for R in $(seq $ROUNDS); do
for V in $(ls ab_variants | sed 's/.so//'); do
  cp ab_variants/$V.so fleetrl_amd/libfleet_hip.so
  for C in ${CADS:-own}; do
    if [ "$C" = own ]; then unset FLEET_RF_CADENCE; CTAG=""; else export FLEET_RF_CADENCE=$C; CTAG="/cad$C"; fi
    case $V in a_r3*) if [ "$C" != own ] && [ "$C" != "${CADS%% *}" ]; then continue; fi;; esac
    for S in $SHAPES; do
      case $S in c*) run --config $S;; *) run --envs-per-gpu $S;; esac
    done
  done
done
done
} | tee gpurun_out/r04/${TAG}.log
