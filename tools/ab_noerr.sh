cd $GRAFT_REPO_ROOT
cp fleetrl_amd/libfleet_hip.so /tmp/keep.so
sed -i 's/        g.batch.check_errors()/        pass/; s/        g0.batch.check_errors()/        pass/' bench.py
for E in 4096 16384; do for f in base noeval nostk noacc nopop noall nopush; do cp ab_variants/$f.so fleetrl_amd/libfleet_hip.so; echo "E=$E $f $(python3 bench.py --steps 1500 --warmup 100 --no-cpu-baseline --no-host-path --envs-per-gpu $E 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['kernel_ms'])")"; done; done
cp /tmp/keep.so fleetrl_amd/libfleet_hip.so
