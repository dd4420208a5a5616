"""Debug helper (GPU box): step HIP and the CPU oracle in lock-step on the headline slice and report the first EV whose
degradation state differs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_hip_shapes import _cfg, _tables
from fleetrl_amd.batch import FleetBatch
from fleetrl_amd.config import resolve_config
from fleetrl_amd.params import make_params, time_features
from oracle.fleet_oracle import OracleBatch

uc, n_evs, num_envs, steps = "ct", 50, 96, 220
tb = _tables(uc, n_evs)
rc = resolve_config(_cfg(uc, "rainflow", False))
p = make_params(rc, tb, num_envs, seed=1)
tf = time_features(tb)
hip, cpu = FleetBatch(p, tb, tf), OracleBatch(p, tb, tf, threads=16)
rng = np.random.default_rng(0)
hip.reset(); cpu.reset()
hist = []
for s in range(steps):
    mode = (s // 40) % 3
    a = rng.uniform(-1, 1, size=(num_envs, n_evs)) if mode == 0 else rng.uniform(-0.2, 1, size=(num_envs, n_evs)) \
        if mode == 1 else np.full((num_envs, n_evs), 1.0)
    a[rng.random(a.shape) < 0.15] = 0.0
    a = a.astype(np.float32)
    hip.step(a); cpu.step(a)
    hist.append(cpu.get("soc_deg").copy())
    bad = None
    for f in ("soh", "rf_len", "fd_cyc", "sei_l"):
        h, c = hip.get(f), cpu.get(f)
        if not np.allclose(h, c, rtol=1e-9, atol=0):
            bad = np.argwhere(~np.isclose(h, c, rtol=1e-9, atol=0))[0]
            print("step", s, "field", f, "env,ev", bad, "hip", h[tuple(bad)], "cpu", c[tuple(bad)])
    if bad is not None:
        e, c = bad
        print("time_idx", hip.get("time_idx")[e], "start", hip.get("start_idx")[e], "ep_len", hip.get("ep_len")[e], "episodes", hip.get("episodes")[e])
        for f in ("soh", "rf_len", "fd_cyc", "fd_cal", "sei_l", "soc_deg"):
            print(f, hip.get(f)[e, c], cpu.get(f)[e, c])
        n = int(hip.get("ep_len")[e])
        ser = np.array([h[e, c] for h in hist[-n:]])
        print("series of this episode (soc_deg after each step):", repr(ser))
        break
else:
    print("no mismatch")
