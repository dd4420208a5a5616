cd $GRAFT_REPO_ROOT
cp fleetrl_amd/libfleet_hip.so /tmp/keep.so
for E in 1024 2048 3072 4096 6144 8192; do for f in base nopush; do cp ab_variants/$f.so fleetrl_amd/libfleet_hip.so; echo "E=$E $f $(python3 bench.py --steps 1500 --warmup 100 --no-cpu-baseline --no-host-path --envs-per-gpu $E 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['kernel_ms'])")"; done; echo "E=$E none $(python3 bench.py --steps 1500 --warmup 100 --no-cpu-baseline --no-host-path --envs-per-gpu $E --deg none 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['kernel_ms'])")"; done
cp /tmp/keep.so fleetrl_amd/libfleet_hip.so
