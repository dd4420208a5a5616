# usage: bash tools/prof_mix.sh <tag> [bench args]  -- VALU instruction mix of the step kernel (per-wave medians)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift; OUT=$R/gpurun_out/prof/$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/a -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_SALU --output-format csv -d $OUT/b -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import glob, sys, pandas as pd
out = sys.argv[1]
for p in ("a", "b"):
    c = pd.read_csv(glob.glob(f"{out}/{p}/*/*counter_collection.csv")[0])
    c = c[c.Kernel_Name.str.contains(r"fleet_step_kernel<\d+, \d+, false,", regex=True)]
    med = c.groupby("Counter_Name").Counter_Value.median()
    w = med["SQ_WAVES"]
    print((med / w).round(1).to_string())
PY
