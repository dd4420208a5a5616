cd $GRAFT_REPO_ROOT
for T in 64 8 2; do for S in "--envs-per-gpu 16384" "--config c5" "--config c3"; do
python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-host-path --tape-len $T $S 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tape %s %-24s kernel_ms %.4f frac %.3f' % (sys.argv[1], sys.argv[2], r['kernel_ms'], r['frac']))" $T "$S"
done; done
