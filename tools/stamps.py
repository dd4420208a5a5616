"""Diagnostic: where a wavefront of the step kernel spends its cycles (s_memtime stamps, FLEET_STAMPS build only).
Run on the GPU box:  FLEET_EXTRA_HIPCC_FLAGS=-DFLEET_STAMPS python3 -c "from fleetrl_amd import build; build.build(force=True)"; python3 tools/stamps.py
Read the SHARES, not the absolute run time (the stamps serialise the schedule)."""
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

E, N = int(os.environ.get("E", 4096)), 50
rc = resolve_config(bench_config(E, N, "ct"))
tb = synth_tables("ct", N)
b = FleetBatch(make_params(rc, tb, E, seed=0), tb, time_features(tb))
dev = torch.device("cuda", 0)
tape = torch.rand((16, E, N), device=dev) * 2 - 1
obs = torch.empty((E, b.obs_dim), device=dev)
rew = torch.empty(E, device=dev, dtype=torch.float64)
done = torch.empty(E, device=dev, dtype=torch.uint8)
b.reset_dev(obs.data_ptr())
b.run_tape_dev(300, tape.data_ptr(), 16, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), use_graph=False)
b.synchronize()
lib = _capi.load_library()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.fleet_debug_read_stamps.argtypes = [C.c_void_p]
assert lib.fleet_debug_read_stamps(buf.ctypes.data) == 0
s = buf.reshape(4096, 16)[: min(E, 4096), :9].astype(np.int64)  # one row per wavefront (G = 64: one env each)
names = ["entry->env head ready", "stage-2 issue + hot loads ready", "charge + state machine", "observation stores",
         "rainflow update", "state stores", "reductions + leader", "SEI pass / reset", ]
d = np.diff(s, axis=1)
print("cycles per segment, median over the wavefronts (last step of the run):")
for k, n in enumerate(names):
    print(f"  {n:34s} {np.median(d[:, k]):8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}")
tot = s[:, 8] - s[:, 0]
print(f"  {'total':34s} {np.median(tot):8.0f}   p90 {np.percentile(tot, 90):8.0f}   max {tot.max():8.0f}")
long = d[:, 7] > 4 * np.median(d[:, 7])
print(f"wavefronts with a long last segment (burst / daily evaluation / reset): {long.mean() * 100:.1f} %")
for name, sel in (("ordinary", ~long), ("long", long)):
    if sel.any():
        print(f"  {name:9s} last segment median {np.median(d[sel, 7]):8.0f}  p90 {np.percentile(d[sel, 7], 90):8.0f}   total median "
              f"{np.median(tot[sel]):8.0f}  p90 {np.percentile(tot[sel], 90):8.0f}")
t0 = s[:, 0].min()
print("kernel timeline [cycles from the first wave's entry]: last entry", int(s[:, 0].max() - t0), " median exit", int(np.median(s[:, 8]) - t0),
      " p90 exit", int(np.percentile(s[:, 8], 90) - t0), " last exit", int(s[:, 8].max() - t0))
print("first wave start spread [cycles]:", int(s[:, 0].max() - s[:, 0].min()), " kernel span:", int(s[:, 8].max() - s[:, 0].min()))
