"""Diagnostic (FLEET_STAMPS builds only): how long a wavefront of the step kernel waits for its FIRST burst of loads -- the env head
(scalar load) and the state records -- by batch size, degradation model and launch path, one line per configuration.  The question
behind it: are the state records served by the die's L2 when the launches go through the library's own queue (no release between
them)?  A burst from the L2 returns in ~750 + ~950 cycles (1024 envs x 50 EVs), one from beyond it in ~1150 + ~1550 and slower with
every further wavefront per SIMD.
usage (GPU box, the stamped library copied over the product library): CONFIGS="E:deg:launch:tape ..." python3 tools/stamps_burst.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import bench_config  # noqa: E402
from fleetrl_amd import _capi  # noqa: E402
from fleetrl_amd.batch import FleetBatch  # noqa: E402
from fleetrl_amd.config import resolve_config  # noqa: E402
from fleetrl_amd.params import make_params, time_features  # noqa: E402
from fleetrl_amd.synth import synth_tables  # noqa: E402

N = 50
tb = synth_tables("ct", N)
tf = time_features(tb)
dev = torch.device("cuda", 0)
lib = _capi.load_library()
SLOTS = 32
tag = os.environ.get("TAG", "")
for cfg in os.environ.get("CONFIGS", "4096:rainflow:2:16").split():
    E, deg, launch, tape_len = cfg.split(":")
    E, launch, tape_len = int(E), int(launch), int(tape_len)
    rc = resolve_config(bench_config(E, N, "ct", deg=deg))
    b = FleetBatch(make_params(rc, tb, E, seed=0), tb, tf)
    tape = torch.rand((tape_len, E, N), device=dev) * 2 - 1
    obs = torch.empty((E, b.obs_dim), device=dev)
    rew = torch.empty(E, device=dev, dtype=torch.float64)
    done = torch.empty(E, device=dev, dtype=torch.uint8)
    b.reset_dev(obs.data_ptr())
    import time
    args = (tape.data_ptr(), tape_len, obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    b.run_tape_dev(6000, *args, use_graph=launch)
    b.synchronize()
    t0 = time.perf_counter()
    b.run_tape_dev(2011, *args, use_graph=launch)
    b.synchronize()
    period = (time.perf_counter() - t0) * 1e6 / 2011
    buf = np.zeros(4096 * SLOTS, dtype=np.uint64)
    if launch >= 2:
        lib.fleet_debug_read_stamps_direct.argtypes = [C.c_void_p, C.c_size_t]
        assert lib.fleet_debug_read_stamps_direct(buf.ctypes.data, buf.nbytes) == 0
    else:
        lib.fleet_debug_read_stamps.argtypes = [C.c_void_p]
        assert lib.fleet_debug_read_stamps(buf.ctypes.data) == 0
    s = buf.reshape(4096, SLOTS)[: min(E, 4096), :9].astype(np.int64)
    s = s[(s > 0).all(axis=1)]
    d = np.diff(s, axis=1)
    q = lambda v: "%5d %5d %5d %5d" % tuple(np.percentile(v, p) for p in (1, 10, 50, 90))  # noqa: E731
    print(f"{tag:10s} E={E:5d} {deg:8s} launch {launch} tape {tape_len:3d}  period {period:6.2f} us | head ready p1/p10/p50/p90 {q(d[:, 0])} | "
          f"state ready {q(d[:, 1])} | both {q(d[:, 0] + d[:, 1])} | rest {q(s[:, 8] - s[:, 2])} | total {q(s[:, 8] - s[:, 0])}", flush=True)
    b.close()
    del b, tape, obs, rew, done
