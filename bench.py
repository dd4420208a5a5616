#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched FleetEnv.step() hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: every env of the batch advances one 15-minute slot
(charge integration, grid balance, arrival/departure state machine, observation, SOC log, daily rainflow/SEI
degradation, auto-reset), ONE kernel launch per step, inputs (action tape, tables, state) resident in HBM.
Workload at N=1 = BASELINE.json configs[2]: 4096 envs x 50 EVs, caretaker fleet, load+pv observations, rainflow
degradation, 48 h episodes, random start rows; synthetic seeded inputs (fleetrl_amd/synth.py).  N>1: every rank runs
its own 4096-env shard (weak scaling), no data-path collective; one RCCL all-gather of episode returns for logging.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel vs the 8 TB/s HBM roof, algorithmic bytes
per SURVEY.md section 8d), `cpu_baseline` (the CPU oracle -- a port of the reference's algorithm -- timed on this box's
host cores on a bounded sample of the same workload), `step_many` (K-steps-per-launch open-loop entry, reported aside).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)


def bench_config(num_envs: int, n_evs: int, use_case: str):
    """The reference's config dict for the benchmark workload (same keys as /root/reference/config.json)."""
    return {
        "data_path": "<synthetic>", "use_case": use_case, "building_name": None, "price_name": None, "tariff_name": None,
        "schedule_name": None, "pv_name": None, "seed": 0, "include_building": True, "include_pv": True,
        "include_price": True, "time_picker": "random", "max_batt_cap_in_all_use_cases": 60, "init_soh": 1.0,
        "log_data": False, "deg_emp": False, "calculate_degradation": True, "verbose": 0, "normalize_in_env": False,
        "aux": True, "ignore_price_reward": False, "ignore_overloading_penalty": False, "ignore_invalid_penalty": False,
        "ignore_overcharging_penalty": False, "gen_schedule": False, "gen_start_date": None, "gen_end_date": None,
        "gen_name": None, "gen_n_evs": n_evs, "spot_markup": None, "spot_mul": None, "feed_in_ded": None,
        "real_time": False, "episode_length": 48, "target_soc": 0.85, "obc_max_power": 100, "min_laxity": 2,
    }


def algorithmic_bytes_per_env_step(n_evs: int, obs_dim: int, tail_a: int, rainflow: bool, episode_steps: int) -> float:
    """bytes = N*B_ev + B_env + B_deg with
    B_ev = 82 (action 4 R, there[t],there[t+1] 2 R, time_left/soc_on_return[t+1] 8 R, soc 16 RW, hours_left 8 RW,
    soh 8 R, soc history append 8 W, 7 per-EV observation slots 28 W);
    B_env = 4*(obs_dim - 7N) W + 17 (time index RW, reward W, done W) + 4*(look-ahead scalars + 6 + 6) R;
    B_deg = 8*N*(mean history length)/96 amortised re-read of the SOC history on the daily degradation step."""
    b_ev = 82.0
    b_env = 4.0 * (obs_dim - 7 * n_evs) + 17.0 + 4.0 * (tail_a + 12)
    b_deg = 8.0 * n_evs * ((episode_steps + 1) / 2.0) / 96.0 if rainflow else 0.0
    return n_evs * b_ev + b_env + b_deg


def cpu_baseline(params, tables, time_feat, n_evs: int, budget_s: float = 8.0):
    """Time the CPU oracle (a port of the reference's algorithm; checker code, never the product) on a bounded sample
    of the same workload: same tables / params / action distribution.  Primary figure: ONE host core (scalar port);
    an OpenMP-over-envs figure on up to 32 threads is reported beside it."""
    import ctypes as C

    from oracle.fleet_oracle import OracleBatch

    def run(num_envs, threads, budget):
        p = type(params)()
        C.memmove(C.byref(p), C.byref(params), C.sizeof(p))
        p.num_envs = num_envs
        eng = OracleBatch(p, tables, time_feat, threads=threads)
        rng = np.random.default_rng(7)
        acts = rng.uniform(-1, 1, size=(8, num_envs, n_evs)).astype(np.float32)
        acts[rng.random(acts.shape) < 0.15] = 0.0
        eng.reset()
        for i in range(2):
            eng.step(acts[i])
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < budget:
            for i in range(8):
                eng.step(acts[i])
            steps += 8
        dt = time.perf_counter() - t0
        eng.close()
        return num_envs * steps / dt, steps, dt

    v1, steps1, dt1 = run(128, 1, budget_s)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthr = max(1, min(avail, 32))
    vm, stepsm, dtm = run(2048, nthr, budget_s)
    return {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"128 envs x {n_evs} EVs x {steps1} steps in {dt1:.1f} s on one core, same tables/params/action "
                      f"distribution as the GPU run",
            "all_cores": {"value": vm, "cores": nthr, "host_cpus": avail,
                          "sample": f"2048 envs x {n_evs} EVs x {stepsm} steps in {dtm:.1f} s, OpenMP over envs"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--evs", type=int, default=50)
    ap.add_argument("--use-case", default="ct")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tape-len", type=int, default=64)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (with --device-index) only exists to smoke-test the multi-rank control "
                         "flow with several ranks on ONE GPU, which RCCL refuses")
    ap.add_argument("--device-index", type=int, default=None, help="GPU to use instead of LOCAL_RANK (test hook, see --backend)")
    ap.add_argument("--prime-ms", type=float, default=300.0, help="untimed clock-ramp replay before the warmup steps")
    ap.add_argument("--deg", default="rainflow", choices=["none", "linear", "rainflow"],
                    help="degradation model (default rainflow = the BASELINE workload; others are diagnostics)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from fleetrl_amd.batch import FleetBatch
    from fleetrl_amd.config import resolve_config
    from fleetrl_amd.distributed import dist_env, gather_episode_stats
    from fleetrl_amd.params import make_params, time_features
    from fleetrl_amd.synth import synth_tables

    rank, local_rank, world = dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.device_index is not None:
        local_rank = args.device_index
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    E, N = args.envs_per_gpu, args.evs
    cfg = bench_config(E, N, args.use_case)
    cfg["calculate_degradation"] = args.deg != "none"
    cfg["deg_emp"] = args.deg == "linear"
    rc = resolve_config(cfg)
    tables = synth_tables(args.use_case, N, seed=1234)
    tf = time_features(tables)
    params = make_params(rc, tables, E, auto_reset=True, env_id_offset=rank * E, seed=0)
    batch = FleetBatch(params, tables, tf, device=local_rank)

    # device-resident action tape: uniform(-1,1) float32, 15 % forced to 0, seeded per rank
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    L = args.tape_len
    tape = torch.rand((L, E, N), device=dev, generator=gen, dtype=torch.float32) * 2 - 1
    tape[torch.rand((L, E, N), device=dev, generator=gen) < 0.15] = 0.0
    obs = torch.empty((E, batch.obs_dim), device=dev, dtype=torch.float32)
    reward = torch.empty(E, device=dev, dtype=torch.float64)
    done = torch.empty(E, device=dev, dtype=torch.uint8)
    torch.cuda.synchronize()
    use_graph = not args.no_graph

    def run(k):
        batch.run_tape_dev(k, tape.data_ptr(), L, obs.data_ptr(), reward.data_ptr(), done.data_ptr(), use_graph=use_graph)

    batch.reset_dev(obs.data_ptr())
    # clock ramp: replay the same step (untimed) for a fixed wall time before the W warmup steps, so that a short
    # --steps/--warmup run measures the same steady state as a long one
    t_prime = time.perf_counter()
    while (time.perf_counter() - t_prime) * 1e3 < args.prime_ms:
        run(256)
        batch.synchronize()
    run(args.warmup)
    batch.synchronize()
    ret = torch.zeros(E, device=dev, dtype=torch.float64)
    ln = torch.zeros(E, device=dev, dtype=torch.int32)
    gather_episode_stats(ret.to(cdev), ln.to(cdev), equal_shards=True)  # warmup of the logging collective too (RCCL channel setup is not a per-step cost)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    batch.timer_start()
    run(args.steps)
    ev_ms = batch.timer_stop()  # HIP events on the stream the kernels run on
    # logging collective: one all-gather of finished-episode returns / lengths (RCCL over xGMI when N > 1)
    batch.get_dev("last_ep_return", ret.data_ptr())  # device-side unpack, no host round trip
    batch.get_dev("last_ep_len", ln.data_ptr())
    batch.synchronize()
    r_all, n_all = gather_episode_stats(ret.to(cdev), ln.to(cdev), equal_shards=True)
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        w = torch.tensor([wall], device=cdev, dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    batch.check_errors()

    out = None
    if rank == 0:
        S = rc.episode_length * (60 // rc.minutes)
        tail_a = 2 * (rc.price_lookahead + 1) + 2 * (rc.bl_pv_lookahead + 1)
        bytes_step = algorithmic_bytes_per_env_step(N, batch.obs_dim, tail_a, True, S)
        # dominant kernel: fleet_step_kernel, one launch per step; per-launch duration from one HIP event pair per launch
        # average launch duration: HIP events on the kernels' stream around the timed region (K graph-replayed launches,
        # ~0.5 us of inter-kernel gap per launch included -> slightly conservative; rocprofv3's per-kernel average is in
        # profiles/).  A second figure brackets every launch with its own event pair (adds ~1.5 us of event overhead).
        k_ms = ev_ms / args.steps
        per = batch.time_steps_dev(min(args.steps, 512), tape.data_ptr(), L, obs.data_ptr(), reward.data_ptr(), done.data_ptr())
        achieved = bytes_step * E / (k_ms * 1e-3) / 1e9
        # K-steps-per-launch entry (open-loop rollouts), reported aside
        K = 64
        rsum = torch.empty(E, device=dev, dtype=torch.float64)
        batch.step_many_dev(K, tape.data_ptr(), obs.data_ptr(), rsum.data_ptr())
        batch.synchronize()
        batch.timer_start()
        reps = max(1, min(args.steps, 2048) // K)
        for _ in range(reps):
            batch.step_many_dev(K, tape.data_ptr(), obs.data_ptr(), rsum.data_ptr())
        many_ms = batch.timer_stop()
        batch.check_errors()
        # HBM traffic of the step kernel from the committed rocprofv3 PMC passes of this same command
        # (tools/prof_traffic.sh -> profiles/r01_traffic_step_kernel.json); null when the workload differs from the profiled one
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic_step_kernel.json")
        if os.path.isfile(tpath) and (E, N, args.use_case, args.deg) == (4096, 50, "ct", "rainflow"):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        out = {
            "metric": "env-steps/sec (num_envs x EVs batch, 1 launch per step)",
            "value": world * E * args.steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{E} envs x {N} EVs per GPU, {args.use_case} fleet, load+pv obs, rainflow/SEI degradation, "
                                   f"48 h episodes, random start rows, auto-reset (BASELINE.json configs[2])",
                       "envs_per_gpu": E, "evs_per_env": N, "obs_dim": batch.obs_dim, "launch": "hipGraph" if use_graph else "eager", "prime_ms": args.prime_ms,
                       "ev_steps_per_s": world * E * N * args.steps / wall},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "fleet_step_kernel<G=64,DEG=rainflow,MULTI=false,WIDE=false>", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_env_step": bytes_step, "bytes_per_launch": bytes_step * E,
                         "kernel_ms_event_pair_per_launch": float(np.mean(per))},
            # K steps per launch: only the last step's observation is part of the result (and written), so the
            # algorithmic bytes per env-step are smaller by the observation row for K-1 of the K steps
            "step_many": {"K": K, "env_steps_per_s": E * K * reps / (many_ms * 1e-3),
                          "algorithmic_bytes_per_env_step": bytes_step - 4 * batch.obs_dim * (K - 1) / K,
                          "GBps": (bytes_step - 4 * batch.obs_dim * (K - 1) / K) * E * K * reps / (many_ms * 1e-3) / 1e9},
            "episodes_gathered": int((n_all > 0).sum().item()),
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is a single-GPU-run item (the other ranks would idle)
            out["cpu_baseline"] = cpu_baseline(params, tables, tf, N)
    batch.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
