# usage: bash tools/prof_traffic.sh <tag> [bench args...]   (on the GPU box through gpurun); tag = r06_traffic_<config>, the
# result (gpurun_out/prof/<tag>/traffic.json) is what gets committed as profiles/<tag>.json
# HBM traffic of the step kernel from the L2's memory-side counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --lean "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --lean "$@" > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --lean "$@" > $OUT/bench_l2.log 2>&1
python3 - "$OUT" "$@" <<'PY'
import glob, json, os, sys
import pandas as pd
out = sys.argv[1]
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
res = {"kernel_src_sha": bench.kernel_source_sha(), "bench_args": sys.argv[2:]}
n_groups = 1
lines = [l for l in open(f"{out}/bench_fetch.log") if l.startswith("{")]
if lines:
    j = json.loads(lines[-1])
    n_groups = j["config"].get("launches_per_step", len(j["config"]["groups"]))  # kernels of the single-step instance per step
    res.update(groups=n_groups, envs=j["config"]["envs_per_gpu"], evs=j["config"]["evs_per_env"], config=j["config"]["name"],
               algorithmic_bytes_per_launch=j["roofline"]["bytes_per_launch"], kernel=j["roofline"]["kernel"],
               launch_mode=j["config"].get("launch_mode", "graph"))
for name in ("fetch", "write", "l2"):
    f = glob.glob(f"{out}/{name}/*/*counter_collection.csv")
    if not f:
        continue
    c = pd.read_csv(f[0])
    c = c[c.Kernel_Name.str.contains(r"fleet_step_kernel<\d+, \d+, false,", regex=True)]  # the single-step instances
    # per STEP: with several fleet types (c5) a step is one launch per type, all the same kernel instance
    for cn, g in c.groupby("Counter_Name"):
        res[cn] = float(g.Counter_Value.mean()) * n_groups
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
