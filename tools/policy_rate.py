import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from bench import bench_config
from fleetrl_amd import _capi
from fleetrl_amd.batch import FleetBatch
from fleetrl_amd.config import resolve_config
from fleetrl_amd.params import make_params, time_features
from fleetrl_amd.policies import night_schedule, run_policy
from fleetrl_amd.synth import synth_tables
E, N = 4096, 50
rc = resolve_config(bench_config(E, N, "ct"))
tb = synth_tables("ct", N)
p = make_params(rc, tb, E, seed=0)
for pol in ("uncontrolled", "distributed", "night"):
    b = FleetBatch(p, tb, time_features(tb))
    b.reset()
    night = night_schedule(tb, target_soc=p.target_soc, init_battery_cap=p.init_battery_cap, charging_eff=p.charging_eff, evse_power=p.evse_power) if pol == "night" else None
    run_policy(b, pol, 192, chunk=96, night=night)
    t0 = time.perf_counter()
    o, r, d = run_policy(b, pol, 96 * 20, chunk=96)
    dt = time.perf_counter() - t0
    print(pol, "%.3e env-steps/s" % (E * 96 * 20 / dt), "mean reward/step %.3f" % (r.mean() / (96 * 20)), "episodes", int(d.sum()))
    b.close()
