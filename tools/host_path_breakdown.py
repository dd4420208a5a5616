import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import bench_config
from fleetrl_amd import FleetVecEnv
from fleetrl_amd.synth import synth_tables
E, N = 4096, 50
venv = FleetVecEnv(bench_config(E, N, "ct"), E, tables=synth_tables("ct", N))
venv.reset()
b = venv.core.batch
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, size=(8, E, N)).astype(np.float32)
def t(f, n=200):
    for i in range(10): f(i)
    t0=time.perf_counter()
    for i in range(n): f(i)
    return (time.perf_counter()-t0)/n*1e6
print("venv.step      us", t(lambda i: venv.step(acts[i%8])))
print("batch.step     us", t(lambda i: b.step(acts[i%8])))
obs = np.empty((E,b.obs_dim),np.float32); term=np.empty_like(obs); rew=np.empty(E); done=np.empty(E,np.uint8)
import ctypes as C
def raw(i):
    a=acts[i%8]
    b.lib.fleet_step_host(b.h, a.ctypes.data, 0, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, term.ctypes.data)
print("raw step_host (prealloc, with terminal) us", t(raw))
def raw2(i):
    a=acts[i%8]
    b.lib.fleet_step_host(b.h, a.ctypes.data, 0, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, None)
print("raw step_host (prealloc, no terminal) us", t(raw2))
print("infos list us", t(lambda i: [{} for _ in range(E)]))
