"""Diagnostic (GPU box): lock-step HIP vs oracle on the 90-day episode of tests/test_rainflow_adversarial_gpu.py, state compared
after EVERY step; prints the first field / EV that differs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_rainflow_adversarial_gpu import _cfg  # noqa: E402

from fleetrl_amd.batch import FleetBatch  # noqa: E402
from fleetrl_amd.config import resolve_config  # noqa: E402
from fleetrl_amd.params import make_params, time_features  # noqa: E402
from fleetrl_amd.synth import synth_tables  # noqa: E402
from oracle.fleet_oracle import OracleBatch  # noqa: E402

N, E = 6, 4
tb = synth_tables("ut", N, seed=31)
rc = resolve_config(_cfg(24 * 90, uc="ut"))
p = make_params(rc, tb, E, seed=3, start_range=(0, 96 * 30))
tf = time_features(tb)
hip, cpu = FleetBatch(p, tb, tf), OracleBatch(p, tb, tf)
hip.reset(); cpu.reset()
rng = np.random.default_rng(4)
hist = []
for s in range(8640):
    soc = hip.get("soc")
    a = np.clip(2.0 * (0.45 - soc) + 0.25 * rng.uniform(-1, 1, size=(E, N)), -0.9, 0.9).astype(np.float32)
    a[rng.random(a.shape) < 0.15] = 0.0
    hip.step(a); cpu.step(a)
    hist.append(cpu.get("soc_deg").copy())
    bad = None
    for f, tol in (("soc", 1e-9), ("soc_deg", 1e-9), ("hours_left", 0), ("soh", 1e-11), ("rf_len", 0), ("fd_cyc", 1e-8), ("fd_cal", 1e-9), ("sei_l", 1e-9)):
        x, y = hip.get(f).astype(np.float64), cpu.get(f).astype(np.float64)
        err = np.abs(x - y) / np.maximum(np.abs(y), 1e-300)
        err[(x == y)] = 0
        if (err > tol).any():
            e, c = np.unravel_index(np.argmax(err), err.shape)
            print(f"step {s}: {f} differs at env {e} ev {c}: hip {x[e, c]!r} cpu {y[e, c]!r} (rel {err[e, c]:.3e}); time_idx {cpu.get('time_idx')[e]} "
                  f"rf_len hip {hip.get('rf_len')[e, c]} cpu {cpu.get('rf_len')[e, c]}")
            bad = (e, c)
            break
    if bad:
        e, c = bad
        ser = np.array([h[e, c] for h in hist])
        print("samples logged so far for that EV:", len(ser), " last 12:", repr(ser[-12:]))
        for f in ("soh", "rf_len", "fd_cyc", "fd_cal", "sei_l", "soc", "soc_deg"):
            print(f, "hip", hip.get(f)[e], "cpu", cpu.get(f)[e])
        break
else:
    print("no difference in 8640 steps")
