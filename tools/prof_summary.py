"""Summarise a tools/prof_step.sh output directory: per-kernel duration stats + per-wave counter medians."""
import glob
import sys

import numpy as np
import pandas as pd

out = sys.argv[1]
kt = pd.read_csv(glob.glob(f"{out}/kt/*/*kernel_trace.csv")[0])
kt["dur"] = kt.End_Timestamp - kt.Start_Timestamp
for name, k in kt.groupby("Kernel_Name"):
    if "fleet" not in name:
        continue
    short = name.replace("void (anonymous namespace)::", "").split("(")[0][-70:]
    print(f"{short}: calls={len(k)} avg={k.dur.mean():.0f}ns p10={k.dur.quantile(.1):.0f} p50={k.dur.median():.0f} "
          f"p90={k.dur.quantile(.9):.0f} max={k.dur.max():.0f} vgpr={k.VGPR_Count.iloc[0]} sgpr={k.SGPR_Count.iloc[0]} "
          f"scratch={k.Scratch_Size.iloc[0]} grid={k.Grid_Size_X.iloc[0]} wg={k.Workgroup_Size_X.iloc[0]}")
for f in ("pmc1", "pmc2"):
    files = glob.glob(f"{out}/{f}/*/*counter_collection.csv")
    if not files:
        continue
    c = pd.read_csv(files[0])
    c = c[c.Kernel_Name.str.contains(r"fleet_step_kernel<\d+, \d+, false,", regex=True)]  # the single-step instances
    med = c.groupby("Counter_Name").Counter_Value.median()
    print(med.to_string())
