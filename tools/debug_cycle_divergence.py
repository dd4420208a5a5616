"""Diagnostic (CPU only): the ONE EV of 204 800 whose rainflow cycle bookkeeping differs between the HIP kernels and the CPU oracle after 400
steps of bench.py's saturating action tape (tests/test_direct_guard_gpu.py::test_direct_run_at_the_headline_shape_against_the_oracle reports
env 180, EV 18).  Question: is that the reference algorithm's own sensitivity to last-bit differences of the SOC samples (its reversal
extraction compares samples exactly) -- or do the kernels' STREAMING rainflow and the reference's batch recount disagree on that series?
This script replays the oracle for the envs up to 180, takes the logged SOC samples of that EV, and runs a Python model of the kernels'
streaming bookkeeping (fleet_kernels.hip rf_begin / rf_finish / sei_evaluate, library pow / exp) on the ORACLE's samples.  If the model
lands on the oracle's fd_cyc, the two algorithms agree on the series and the GPU's other value comes from its samples' last bits.
usage: python tools/debug_cycle_divergence.py [env] [ev]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import bench_config  # noqa: E402
from fleetrl_amd.config import resolve_config  # noqa: E402
from fleetrl_amd.params import make_params, time_features  # noqa: E402
from fleetrl_amd.synth import synth_tables  # noqa: E402
from oracle.fleet_oracle import OracleBatch  # noqa: E402

ENV, EV = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (180, 18)
E_full, N, L, chunks = 4096, 50, 32, (1, 63, 200, 136)
tb = synth_tables("ct", N)
tf = time_features(tb)
rng = np.random.default_rng(21)
acts = rng.uniform(-1, 1, size=(L, E_full, N)).astype(np.float32)
acts[rng.random(acts.shape) < 0.15] = 0.0
E = ENV + 1
acts = np.ascontiguousarray(acts[:, :E])
p = make_params(resolve_config(bench_config(E, N, "ct")), tb, E, auto_reset=True, seed=0)
cpu = OracleBatch(p, tb, tf, threads=min(8, os.cpu_count() or 1))
cpu.reset()
samples, rows, episodes = [cpu.get("soc_deg")[ENV, EV]], [int(cpu.get("time_idx")[ENV])], [int(cpu.get("episodes")[ENV])]
for chunk in chunks:
    for k in range(chunk):
        cpu.step(acts[k % L])
        samples.append(cpu.get("soc_deg")[ENV, EV])
        rows.append(int(cpu.get("time_idx")[ENV]))
        episodes.append(int(cpu.get("episodes")[ENV]))
want = dict(fd_cyc=cpu.get("fd_cyc")[ENV, EV], rf_len=int(cpu.get("rf_len")[ENV, EV]), sei_l=cpu.get("sei_l")[ENV, EV])

# ---- the kernels' streaming bookkeeping on the oracle's samples -----------------------------------------------------------------
st = math.exp(6.93e-2 * (p.temperature - 25.0) * ((25.0 + 273.15) / (p.temperature + 273.15)))


def stress(r, mean, count):
    eff = min(r * count, 1.0)
    if not eff > 0.0:
        return 0.0
    return 1.0 / (1.4e5 * eff ** -0.501 - 1.23e5) * math.exp(1.04 * (mean - 0.5)) * st


ties = []       # (relative margin of a three-point test |X - Y| / Y, sample index): a margin of ~1e-16 flips with the samples' last bit
plateaus = []   # sample indices where a sample EQUALS its predecessor exactly while the EV is plugged in (the reversal extraction skips it)
rf_len, fd_cyc, sei_l = 1, 0.0, 1.0 - p.init_soh
stack, sgn, nc, mean_sum, csum, nsamp, prev = [samples[0]], 0, 0, 0.0, 0.0, 1, samples[0]
hour, minute = np.asarray(tb.hour), np.asarray(tb.minute)
for i in range(1, len(samples)):
    x = samples[i]
    if episodes[i] != episodes[i - 1]:
        # the step ended the episode: its sample was logged and (on a degradation row) evaluated before the reset; the oracle's getter
        # already shows the new episode's first sample -- the ended episode's last sample is not observable here, which is fine as long
        # as the episode does not end on a degradation row with a push (reported below)
        stack, sgn, nc, mean_sum, csum, nsamp, prev = [x], 0, 0, 0.0, 0.0, 1, x
        continue
    nsamp += 1
    if x == prev and x not in (0.0,) and i > 1 and samples[i - 2] != x:
        plateaus.append(i)
    if x != prev:
        s_next = 1 if x > prev else 2
        if sgn != 0 and sgn != s_next:  # `prev` is a reversal point: push it, close what the three-point rule allows
            stack.append(prev)
            if len(stack) >= 3:
                X, Y = abs(stack[-1] - stack[-2]), abs(stack[-2] - stack[-3])
                ties.append((abs(X - Y) / max(Y, 1e-300), i))
            while len(stack) >= 3 and not abs(stack[-1] - stack[-2]) < abs(stack[-2] - stack[-3]):
                half = len(stack) == 3
                a, b = (stack[0], stack[1]) if half else (stack[-3], stack[-2])
                if nc >= rf_len - 1:
                    csum += stress(abs(a - b), 0.5 * (a + b), 0.5 if half else 1.0)
                mean_sum += 0.5 * (a + b)
                nc += 1
                if half:
                    stack.pop(0)
                else:
                    last = stack.pop(); stack.pop(); stack.pop(); stack.append(last)
        sgn = s_next
        prev = x
    if hour[rows[i]] == 14 and minute[rows[i]] == 45 and nsamp >= 3:  # the daily evaluation: forced last point on a virtual stack
        v, vs = x, list(stack)
        cyc = []
        while len(vs) + 1 >= 3:
            a, b = vs[-2], vs[-1]
            if abs(v - b) < abs(b - a):
                break
            if len(vs) + 1 == 3:
                cyc.append((a, b, 0.5)); vs.pop(0)
            else:
                cyc.append((a, b, 1.0)); vs.pop(); vs.pop()
        for j in range(len(vs) - 1):
            cyc.append((vs[j], vs[j + 1], 0.5))
        cyc.append((vs[-1], v, 0.5))
        ln = nc + len(cyc)
        if ln > 0 and ln > rf_len:
            vsum = sum(stress(abs(a - b), 0.5 * (a + b), c) for k, (a, b, c) in enumerate(cyc[:-1]) if nc + k >= rf_len - 1)
            fd_cyc = fd_cyc + (csum + vsum)
            mean_cal = (mean_sum + sum(0.5 * (a + b) for a, b, _ in cyc)) / ln
            fd_cal = 4.14e-10 * ((nsamp - 1) * p.dt * 3600.0) * math.exp(1.04 * (mean_cal - 0.5)) * st
            fd = fd_cyc + fd_cal
            sei_l = 1.0 - 5.75e-2 * math.exp(-121.0 * fd) - (1.0 - 5.75e-2) * math.exp(-fd)
            rf_len, csum = ln, 0.0
print(f"env {ENV} EV {EV}: oracle fd_cyc {want['fd_cyc']!r} rainflow_length {want['rf_len']} sei_l {want['sei_l']!r}")
print(f"streaming model on the oracle's samples: fd_cyc {fd_cyc!r} rainflow_length {rf_len} sei_l {sei_l!r}")
print("relative difference of fd_cyc:", abs(fd_cyc - want["fd_cyc"]) / abs(want["fd_cyc"]))
ties.sort()
print("closest three-point tests |X - Y| / Y (sample index):", [(f"{m:.2e}", i) for m, i in ties[:5]])
print(f"{len(plateaus)} samples equal their predecessor exactly (saturated SOC: the reversal extraction skips them); first:", plateaus[:8])
