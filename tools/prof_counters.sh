# usage (GPU box): bash tools/prof_counters.sh <tag> "<counter list>" [bench args...] -- one rocprofv3 --pmc pass of bench.py,
# per-launch medians of the single-step kernel's counters.  Keep a pass to about four counters of one hardware block: a
# request the hardware cannot schedule aborts rocprofv3, which then hangs in its signal handler (hence the timeout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; CNT=$2; shift; shift
OUT=$R/gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 240 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/pmc -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-host-path "$@" > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import glob, os, sys
import pandas as pd
f = glob.glob(f"{sys.argv[1]}/pmc/*/*counter_collection.csv")
if not f:
    print("no counter output:"); print(open(f"{sys.argv[1]}/bench.log").read()[-1500:]); sys.exit(0)
c = pd.read_csv(f[0])
c = c[c.Kernel_Name.str.contains(r"fleet_step_kernel<\d+, \d+, %s," % os.environ.get("FLEET_PROF_MULTI", "false"), regex=True)]
print(c.groupby("Counter_Name").Counter_Value.median().to_string())
PY
