cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/calib; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/calib_traffic.py > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/calib_traffic.py > $OUT/w.log 2>&1
python3 - $OUT <<'PY'
import glob, sys, pandas as pd
out=sys.argv[1]
for name in ("fetch","write"):
    c=pd.read_csv(glob.glob(f"{out}/{name}/*/*counter_collection.csv")[0])
    for k,g in c.groupby("Kernel_Name"):
        print(name, k[:70], g.Counter_Name.iloc[0], g.Counter_Value.median(), "KiB-units; grid", g.Grid_Size.iloc[0] if "Grid_Size" in g else "")
PY
