# usage (GPU box): bash tools/ubench/sector_fetch.sh -- request sizes between the L2 and the fabric for loads / stores that touch
# one half of a 128-byte line (see sector_fetch.hip); per kernel: median over its launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/sector; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/r04; cd $R
for P in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $P | cut -c1-18 | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/$T -- $R/tools/ubench/sector_fetch > $OUT/$T.log 2>&1
done
python3 - $OUT <<'PY' | tee $R/gpurun_out/r04/sector_fetch.log
import glob, sys, pandas as pd
rows = {}
for f in glob.glob(f"{sys.argv[1]}/*/*/*counter_collection.csv"):
    c = pd.read_csv(f)
    for (k, n), g in c.groupby(["Kernel_Name", "Counter_Name"]):
        rows.setdefault(k, {})[n] = g.Counter_Value.median()
names = {"0, false": "lo16", "1, false": "hi16", "2, false": "lo48", "3, false": "both", "4, false": "pair64",
         "0, true": "lo16_w", "2, true": "lo48_w", "3, true": "both_w", "4, true": "pair64_w"}
print("1 Mi lanes, one row per lane; counters per launch / 2^20 (i.e. per row)")
for k, v in sorted(rows.items()):
    tag = next((t for s, t in names.items() if f"<{s}>" in k), k[:40])
    print("%-9s" % tag, "  ".join("%s %.3f" % (n.replace("TCC_EA0_", "").replace("_sum", ""), x / 2**20) for n, x in sorted(v.items())))
PY
rm -rf $OUT
