# usage: bash tools/prof_traffic.sh <tag> [bench args...]   (on the GPU box through gpurun)
# HBM traffic of the step kernel from the L2's memory-side counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench_l2.log 2>&1
python3 - "$OUT" <<'PY'
import glob, json, sys
import pandas as pd
out = sys.argv[1]
res = {}
for name in ("fetch", "write", "l2"):
    f = glob.glob(f"{out}/{name}/*/*counter_collection.csv")
    if not f:
        continue
    c = pd.read_csv(f[0])
    c = c[c.Kernel_Name.str.contains("fleet_step_kernel") & ~c.Kernel_Name.str.contains("true>")]
    for cn, g in c.groupby("Counter_Name"):
        res[cn] = float(g.Counter_Value.median())
# FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
# streams (MI355X_MICROARCH.md "HBM"), so the read side is doubled.  Our accesses are 16-B-per-lane records for the
# state/tables and 4-B-per-lane observation stores: the doubling is exact for the former, approximate overall.
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["hbm_read_bytes_per_launch"] = res["FETCH_SIZE"] * 1024 * 2
    res["hbm_write_bytes_per_launch"] = res["WRITE_SIZE"] * 1024
    res["hbm_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
if "TCC_HIT_sum" in res:
    res["l2_hit_rate"] = res["TCC_HIT_sum"] / max(res["TCC_HIT_sum"] + res["TCC_MISS_sum"], 1)
print(json.dumps(res))
open(f"{out}/traffic.json", "w").write(json.dumps(res, indent=1))
PY
