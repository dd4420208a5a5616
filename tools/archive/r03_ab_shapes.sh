# usage (GPU box): bash tools/r03_ab_shapes.sh <tag> -- every ab_variants/*.so at several batch shapes, same device
cd $GRAFT_REPO_ROOT
TAG=${1:-abshapes}
mkdir -p gpurun_out/r03
cp fleetrl_amd/libfleet_hip.so /tmp/keep4.so
trap "cp /tmp/keep4.so fleetrl_amd/libfleet_hip.so" EXIT
run() { python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-host-path "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-12s %-28s kernel_ms %.4f frac %.3f' % (sys.argv[1], sys.argv[2], r['kernel_ms'], r['frac']))" "$V" "$*"; }
{
for V in $(ls ab_variants | sed 's/.so//'); do
  cp ab_variants/$V.so fleetrl_amd/libfleet_hip.so
  run --config c3
  run --envs-per-gpu 8192
  run --envs-per-gpu 16384
  run --config c5
  run --config c4
done
} | tee gpurun_out/r03/${TAG}.log
