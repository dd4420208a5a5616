#!/usr/bin/env python3
"""Summary statistics of the REFERENCE's schedule generator (build container only: imports /root/reference).

    python oracle/gen_schedule_stats.py      ->  tests/golden/schedule_stats.json

Runs `fleetrl.utils.schedule.schedule_generator.ScheduleGenerator` (unmodified) for each use case over a few 12-week windows
with different seeds -- it is O(rows^2), so a year per vehicle would take hours -- and stores what
`fleetrl_amd.schedule_gen.summarize_schedule` makes of its output: data only, no reference source text.  The fixture pins the
DISTRIBUTIONS our own generator (fleetrl_amd/schedule_gen.py) has to reproduce (tests/test_schedule_gen.py).
"""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_harness import _prepare_imports, base_config  # noqa: E402

WINDOWS = [("2020-01-06 00:00", "2020-03-29 23:45"), ("2020-03-30 00:00", "2020-06-21 23:45"), ("2020-06-22 00:00", "2020-09-13 23:45")]
SEEDS = [11, 23, 37]


def main():
    import pandas as pd

    _prepare_imports()
    from fleetrl.utils.schedule.schedule_config import ScheduleType
    from fleetrl.utils.schedule.schedule_generator import ScheduleGenerator

    from fleetrl_amd.schedule_gen import summarize_schedule

    out = {}
    for uc, st in (("lmd", ScheduleType.Delivery), ("ct", ScheduleType.Caretaker), ("ut", ScheduleType.Utility)):
        frames = []
        t0 = time.time()
        for k, ((a, b), seed) in enumerate(zip(WINDOWS, SEEDS)):
            cfg = base_config()
            cfg.update(use_case=uc, seed=seed, gen_start_date=a, gen_end_date=b, freq="15T")
            with contextlib.redirect_stdout(io.StringIO()):
                f = ScheduleGenerator(env_config=cfg, schedule_type=st, vehicle_id=str(k)).generate_schedule()
            f["ID"] = k
            frames.append(f)
            print(f"  {uc}: window {k + 1}/{len(WINDOWS)} done ({time.time() - t0:.0f}s)", flush=True)
        out[uc] = summarize_schedule(pd.concat(frames))
        out[uc]["columns"] = list(frames[0].columns)
    path = os.path.join(ROOT, "tests", "golden", "schedule_stats.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
