#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched FleetEnv.step() hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: every env of the batch advances one 15-minute slot (charge
integration, grid balance, arrival/departure state machine, observation, SOC log, daily degradation, auto-reset), ONE kernel
launch per step and fleet type, inputs (action tape, tables, state) resident in HBM.

Workloads (`--config`, per GPU; synthetic seeded inputs from fleetrl_amd/synth.py):
  c3 (default)  BASELINE.json configs[2]: 4096 envs x 50 EVs, caretaker fleet, load+pv observations, rainflow/SEI degradation
  c2            configs[1]: 256 envs x 5 EVs, last-mile delivery, price-only observations, linear degradation
  c4            configs[3], one GPU's shard (16384 / 8): 2048 envs x 50 EVs, utility fleet, load+pv, rainflow
  c5            configs[4], one GPU's shard (65536 / 8): 8192 envs x 200 EVs, one third each lmd / ct / ut (own tables and
                parameters per fleet type = three handles on three HIP streams), spot_2021-like prices, fixed feed-in tariff
N > 1: every rank runs its own shard of that size (weak scaling), no data-path collective; one RCCL all-gather of episode
returns for logging, outside the timed region and reported as `log_gather_ms`.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel vs the 8 TB/s HBM roof, algorithmic bytes per
SURVEY.md section 8d), `cpu_baseline` (the CPU oracle -- a port of the reference's algorithm -- timed on this box's host
cores on a bounded sample of the same workload), `step_many` (K-steps-per-launch open-loop entry) and `host_path` (the
host-pointer entry points incl. PCIe, what an SB3 loop sees), both reported aside.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)

CONFIGS = {
    "c2": dict(envs=256, evs=5, groups=("lmd",), building=False, pv=False, deg="linear", price_year="2020", feed_in="spot",
               what="BASELINE.json configs[1]"),
    "c3": dict(envs=4096, evs=50, groups=("ct",), building=True, pv=True, deg="rainflow", price_year="2020", feed_in="spot",
               what="BASELINE.json configs[2]"),
    "c4": dict(envs=2048, evs=50, groups=("ut",), building=True, pv=True, deg="rainflow", price_year="2020", feed_in="spot",
               what="BASELINE.json configs[3], one GPU's shard of 16384 envs"),
    "c5": dict(envs=8192, evs=200, groups=("lmd", "ct", "ut"), building=True, pv=True, deg="rainflow", price_year="2021",
               feed_in="fixed", what="BASELINE.json configs[4], one GPU's shard of 65536 envs"),
}


def bench_config(num_envs: int, n_evs: int, use_case: str, building: bool = True, pv: bool = True, deg: str = "rainflow"):
    """The reference's config dict for a benchmark workload (same keys as /root/reference/config.json)."""
    return {
        "data_path": "<synthetic>", "use_case": use_case, "building_name": None, "price_name": None, "tariff_name": None,
        "schedule_name": None, "pv_name": None, "seed": 0, "include_building": building, "include_pv": pv,
        "include_price": True, "time_picker": "random", "max_batt_cap_in_all_use_cases": 60, "init_soh": 1.0,
        "log_data": False, "deg_emp": deg == "linear", "calculate_degradation": deg != "none", "verbose": 0,
        "normalize_in_env": False, "aux": True, "ignore_price_reward": False, "ignore_overloading_penalty": False,
        "ignore_invalid_penalty": False, "ignore_overcharging_penalty": False, "gen_schedule": False, "gen_start_date": None,
        "gen_end_date": None, "gen_name": None, "gen_n_evs": n_evs, "spot_markup": None, "spot_mul": None, "feed_in_ded": None,
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


def kernel_name(n_evs: int, deg: str) -> str:
    """The step-kernel instance a launch of this geometry runs (fleet_kernels.hip launch_step_gd)."""
    g = 1
    while g < n_evs and g < 64:
        g <<= 1
    if 64 < n_evs <= 256:  # groups of two / four wavefronts per env, one EV per lane (kMaxGroup)
        g = 128 if n_evs <= 128 else 256
    return (f"fleet_step_kernel<G={g},DEG={deg},MULTI=false,WIDE={'true' if n_evs > 256 else 'false'}>")


def kernel_source_sha() -> str:
    """Identifies the kernel a committed traffic profile was taken on: hash of the kernel sources."""
    h = hashlib.sha256()
    for f in ("fleet_kernels.hip", "fleet_device.h"):
        h.update(open(os.path.join(ROOT, "fleetrl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(config: str, envs: int, evs: int, launch_mode: str = "direct"):
    """(HBM bytes per launch, source) from the committed rocprofv3 PMC passes of this same command (tools/prof_traffic.sh ->
    profiles/r06_traffic_<config>.json); (None, why) when the profile is absent, was taken on another kernel source or shape.
    The counters need rocprofv3, so they cannot be read inside this run: the figure is looked up, and `traffic_source` says so."""
    for name in (f"r06_traffic_{config}.json", f"r06_traffic_{envs}x{evs}.json"):  # the config's own shape, or an override's
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.isfile(path):
            continue
        t = json.load(open(path))
        if t.get("kernel_src_sha") == kernel_source_sha() and (t.get("envs"), t.get("evs"), t.get("config")) == (envs, evs, config) \
                and t.get("launch_mode", "graph") == launch_mode:
            return t.get("hbm_bytes_per_launch"), (f"profiles/{name}: FETCH_SIZE x 2 + WRITE_SIZE from separate rocprofv3 --pmc passes of "
                                                   "this command on this kernel source (tools/prof_traffic.sh)")
    return None, "no committed rocprofv3 PMC profile for this kernel source, shape and launch mode"


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

    e1 = max(1, min(128, 6400 // n_evs))
    v1, steps1, dt1 = run(e1, 1, budget_s)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthr = max(1, min(avail, 32))
    em = max(nthr, min(2048, 102400 // n_evs))
    vm, stepsm, dtm = run(em, nthr, budget_s)
    return {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{e1} envs x {n_evs} EVs x {steps1} steps in {dt1:.1f} s on one core, same tables/params/action "
                      f"distribution as the GPU run",
            "all_cores": {"value": vm, "cores": nthr, "host_cpus": avail,
                          "sample": f"{em} envs x {n_evs} EVs x {stepsm} steps in {dtm:.1f} s, OpenMP over envs"}}


class Group:
    """One fleet type of the workload: a handle of the C ABI (own HIP stream) with its device-resident action tape and
    output slices."""

    def __init__(self, torch, dev, use_case, envs, evs, spec, rank, env_id_offset, tape_len, seed):
        from fleetrl_amd.batch import FleetBatch
        from fleetrl_amd.config import resolve_config
        from fleetrl_amd.params import make_params, time_features
        from fleetrl_amd.synth import synth_tables

        self.use_case, self.E, self.N = use_case, envs, evs
        cfg = bench_config(envs, evs, use_case, spec["building"], spec["pv"], spec["deg"])
        self.rc = resolve_config(cfg)
        self.tables = synth_tables(use_case, evs, seed=1234, include_building=spec["building"], include_pv=spec["pv"],
                                   price_year=spec["price_year"], feed_in=spec["feed_in"])
        self.tf = time_features(self.tables)
        self.params = make_params(self.rc, self.tables, envs, auto_reset=True, env_id_offset=env_id_offset, seed=0)
        self.batch = FleetBatch(self.params, self.tables, self.tf, device=dev.index)
        # device-resident action tape: uniform(-1,1) float32, 15 % forced to 0, seeded per rank / group
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        self.L = tape_len
        self.tape = torch.rand((tape_len, envs, evs), device=dev, generator=gen, dtype=torch.float32) * 2 - 1
        self.tape[torch.rand((tape_len, envs, evs), device=dev, generator=gen) < 0.15] = 0.0
        self.obs = torch.empty((envs, self.batch.obs_dim), device=dev, dtype=torch.float32)
        self.reward = torch.empty(envs, device=dev, dtype=torch.float64)
        self.done = torch.empty(envs, device=dev, dtype=torch.uint8)
        S = self.rc.episode_length * (60 // self.rc.minutes)
        tail_a = 2 * (self.rc.price_lookahead + 1) + (self.rc.bl_pv_lookahead + 1) * (int(spec["building"]) + int(spec["pv"]))
        self.rainflow = spec["deg"] == "rainflow"
        self.bytes_step = algorithmic_bytes_per_env_step(evs, self.batch.obs_dim, tail_a, self.rainflow, S)

    def run(self, k, use_graph):
        self.batch.run_tape_dev(k, self.tape.data_ptr(), self.L, self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                                use_graph=use_graph)


def self_launch(n: int) -> int:
    """Run this command under `torch.distributed.run` with one rank per GPU; returns the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif res.stdout:
        sys.stderr.write(res.stdout[-4000:])
    return res.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS), help="BASELINE.json workload (per GPU), see the module docstring")
    ap.add_argument("--envs-per-gpu", type=int, default=None, help="override the config's batch size")
    ap.add_argument("--evs", type=int, default=None, help="override the config's EVs per env")
    ap.add_argument("--use-case", default=None, help="override the config's fleet type(s) with one type")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--launch", choices=("graph", "eager", "direct", "direct1"), default="direct",
                    help="how the K launches of a region reach the GPU.  direct (default): AQL packets written by the library into a queue "
                         "of its own without the L2 write-back HIP attaches to every kernel boundary (fleet_hip.h FLEET_LAUNCH_DIRECT; a "
                         "batch of >= 6144 wavefronts goes to two queues, direct1 keeps it on one) -- an OPEN-LOOP replay: a step's outputs "
                         "are visible after the region only; graph: a replayed hipGraph (eager below 64 steps), eager: one hipLaunchKernel "
                         "each -- CLOSED LOOP: every step's outputs are visible before the next step starts, what fleet_step_dev and every "
                         "Gym / SB3 loop get.  The kernel time of the other path is reported beside the headline's "
                         "(roofline.other_launch_paths)")
    ap.add_argument("--phase", choices=("locked", "staggered"), default="locked",
                    help="locked (default): all envs start their episodes together and stay in lock step -- what the reference's vec env "
                         "does by construction (fixed-length episodes, every env auto-resets on the same step); a launch's duration then "
                         "depends on the episode step it is and one launch in 192 resets every env.  staggered: the envs' episodes are "
                         "de-phased first (masked resets spread over one episode length), so that every launch sees the stationary mixture "
                         "of episode ages and ends a few episodes.  Whichever is timed, the other's kernel time is reported beside it "
                         "(roofline.other_phase)")
    ap.add_argument("--lean", action="store_true",
                    help="only the headline's launches (profiling runs: rocprofv3 aggregates by kernel name, and the other launch path, the "
                         "other episode phase, the K-step entry and the host path all run instances of the same kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--tape-len", type=int, default=32,
                    help="independent rows of the device-resident action tape (replayed cyclically; a run of fewer steps uses as many rows "
                         "as it has steps).  32 at every shape: a much shorter tape is another workload -- the same few action vectors "
                         "repeated forever drive every battery to its limit and the rainflow push drops out of the step "
                         "(config.workload_invariants says what the run did)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (with --device-index) only exists to smoke-test the multi-rank control "
                         "flow with several ranks on ONE GPU, which RCCL refuses")
    ap.add_argument("--gather", default="torch", choices=["torch", "capi"],
                    help="the logging all-gather: through torch.distributed (default), or as ONE ncclAllGather issued by the C ABI "
                         "(fleet_gather_episode_stats_rccl; single fleet type, --backend nccl; falls back to torch on any error)")
    ap.add_argument("--device-index", type=int, default=None, help="GPU to use instead of LOCAL_RANK (test hook, see --backend)")
    ap.add_argument("--prime-ms", type=float, default=300.0, help="untimed clock-ramp replay before the warmup steps")
    ap.add_argument("--deg", default=None, choices=["none", "linear", "rainflow"],
                    help="override the config's degradation model (diagnostics)")
    ap.add_argument("--split", type=int, default=1,
                    help="diagnostic: run every fleet type's envs as this many handles (own stream each), whose launches overlap")
    ap.add_argument("--reps", type=int, default=None,
                    help="timed regions of exactly --steps launches each, run back to back; the median region is reported "
                         "(default: 31 for runs of up to 256 steps, else 7)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process (nothing in this process
    # has touched the GPU yet -- a process that has must never exec another program on this pool), relay its JSON line
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    from fleetrl_amd.distributed import dist_env, gather_episode_stats, shard_range

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
    launched = "WORLD_SIZE" in os.environ  # under torch.distributed.run: the collectives run even with one rank (RCCL smoke test)
    if launched:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    spec = dict(CONFIGS[args.config])
    if args.deg:
        spec["deg"] = args.deg
    if args.use_case:
        spec["groups"] = (args.use_case,)
    if args.split > 1:
        spec["groups"] = tuple(uc for uc in spec["groups"] for _ in range(args.split))
    E = args.envs_per_gpu or spec["envs"]
    N = args.evs or spec["evs"]
    # action tape: L steps of [E, N] float32 actions, replayed cyclically
    L = max(1, min(args.tape_len, args.steps))
    graph_len = L * ((64 + L - 1) // L)  # launches per captured graph: whole tape cycles, at least 64 (fleet_run_tape_dev)
    # launches go through a captured hipGraph of L steps; a run shorter than 64 steps launches eagerly (one graph launch costs
    # about as much as six kernel launches on the host: 9.9 vs 9.65 us per step measured for the driver's 20-step regions)
    launch_mode = "eager" if args.no_graph else args.launch
    if launch_mode == "graph" and args.steps < 64:
        launch_mode = "eager"
    use_graph = {"eager": 0, "graph": 1, "direct": 2, "direct1": 3}[launch_mode]  # _capi.LAUNCH_*
    launch_note = ""
    groups, off = [], 0
    for k, uc in enumerate(spec["groups"]):
        lo, hi = shard_range(E, len(spec["groups"]), k)  # env groups in order: the first E % n groups hold one env more
        groups.append(Group(torch, dev, uc, hi - lo, N, spec, rank, rank * E + off, L, 1 + rank * 16 + k))
        off += hi - lo
    torch.cuda.synchronize()

    def run(k):
        for g in groups:  # asynchronous, one stream per group: the groups' kernels overlap
            g.run(k, use_graph)

    def sync():
        for g in groups:
            g.batch.synchronize()

    for g in groups:
        g.batch.reset_dev(g.obs.data_ptr())
    if use_graph >= 2:
        # the library's own AQL queue needs its code object beside the library and an HSA agent for the device: where that is not
        # to be had (an older build tree, a profiler that does not pass foreign queues) the run says so and replays a hipGraph
        try:
            for g in groups:
                g.run(1, use_graph)
            sync()
        except Exception as ex:  # every rank decides alike only if the cause is in the tree; a lone failing rank raises below
            launch_note = f"direct submission unavailable ({ex}); "
            launch_mode = "graph" if args.steps >= 64 else "eager"
            use_graph = {"eager": 0, "graph": 1}[launch_mode]
        for g in groups:
            g.batch.reset_dev(g.obs.data_ptr())
    def barrier():
        torch.cuda.synchronize()
        if launched:  # one rank: the synchronize IS the barrier (a second one would only add host time to every timed region)
            dist.barrier()
            torch.cuda.synchronize()

    reps = args.reps if args.reps else (31 if args.steps <= 256 else 7)
    n_ev = max(3, min(reps, 15))

    def kernel_regions(mode):
        """The kernels' own time for `n_ev` regions of exactly K launches through launch path `mode`, [region][group] in ms: HIP events
        on the streams the kernels run on -- or, for the library's own queue, which no HIP event can bracket, the dispatch timestamps
        of each region's first and last packet (start of the first launch to end of the last).  The regions are enqueued back to
        back and read afterwards: no host gap between them, so a short region measures the same running kernel as a long one."""
        barrier()
        for g in groups:
            g.batch.time_regions_begin(n_ev, args.steps, g.tape.data_ptr(), g.L, g.obs.data_ptr(), g.reward.data_ptr(), g.done.data_ptr(),
                                       use_graph=mode)
        per_group = [g.batch.time_regions_read() for g in groups]
        return [[float(pg[i]) for pg in per_group] for i in range(n_ev)]

    def kernel_ms(mode):
        regs = kernel_regions(mode)
        return max(sorted(regs, key=max)[len(regs) // 2]) / args.steps

    # clock ramp: replay the same step (untimed) for a fixed wall time before the W warmup steps, so that a short
    # --steps/--warmup run measures the same steady state as a long one
    t_prime = time.perf_counter()
    while (time.perf_counter() - t_prime) * 1e3 < args.prime_ms:
        run(4 * L)
        sync()
    # ---- episode phase -------------------------------------------------------------------------------------------------------
    # After a plain reset all envs run their 192-step episodes in lock step -- the reference's vec env does by construction (every env
    # has the same episode_length and is auto-reset on the step it ends: complete_pipeline.ipynb cell 13, fleet_environment.py:627-628).
    # A launch's duration then depends on WHICH episode step it is (nothing is pushed at step 0, closures carry stress towards the
    # end) and one launch in 192 resets every env.  Staggered: env e is reset once more after (e * 2654435761 mod 2^32) mod S further
    # steps, so the episode ages are uniform over 0 .. S-1 and every launch is the stationary mixture (1/S of the envs end their
    # episode in it -- and the few wavefronts that reset their env then set every launch's duration).
    S_ep = groups[0].rc.episode_length * (60 // groups[0].rc.minutes)

    def stagger():
        for g in groups:
            g.batch.reset_dev(g.obs.data_ptr())
            ph = (torch.arange(g.E, device=dev, dtype=torch.int64) * 2654435761 % (1 << 32)) % S_ep
            g.masks = (ph[None, :] == torch.arange(S_ep, device=dev)[:, None]).to(torch.uint8).contiguous()
        torch.cuda.synchronize()
        for k in range(1, S_ep):  # (phase 0 keeps the reset it has just had)
            for g in groups:
                g.batch.step_dev(g.tape[k % g.L].data_ptr(), g.obs.data_ptr(), g.reward.data_ptr(), g.done.data_ptr())
                g.batch.reset_dev(g.obs.data_ptr(), g.masks[k].data_ptr())
        sync()
        for g in groups:
            del g.masks

    if args.phase == "staggered":
        stagger()
    run(args.warmup)
    sync()
    ret = torch.zeros(E, device=dev, dtype=torch.float64)
    ln = torch.zeros(E, device=dev, dtype=torch.int32)
    gather_episode_stats(ret.to(cdev), ln.to(cdev), equal_shards=True)  # warmup of the logging collective too (RCCL channel setup is not a per-step cost)

    # ---- the timed regions: each is exactly K steps between two barriers ------------------------------------------------------
    # One region is what the contract describes (barrier + synchronize, K launches, synchronize + barrier).  A single region of
    # a short run mostly measures the host's wake-up after the synchronize (tens of microseconds against 20 launches of 9 us), so
    # R regions are run back to back and the MEDIAN one is reported (`reps`, with the fastest and slowest beside it); inside a
    # region the wait for the device is a spin on hipStreamQuery instead of a blocking synchronize.
    # N > 1: a rank's span ends when ITS device is idle after the K launches; the closing barrier follows, and the line reports the
    # MAX over the ranks of those spans (with the closing RCCL barrier inside the span every rank would report the same figure, the
    # slowest rank's PLUS a collective's latency, and the MAX over ranks would be moot).  `ms_per_step_incl_closing_barrier` keeps the
    # other reading beside it.  N = 1: the synchronize IS the barrier, one figure.
    walls, walls_closed = [], []
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        for g in groups:
            g.batch.spin_wait()
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        if launched:
            dist.barrier()
            torch.cuda.synchronize()
        walls_closed.append(time.perf_counter() - t0)
    # the kernels' own time (roofline), kept out of the wall-clock regions (kernel_regions above) -- through the headline's launch
    # path, then through the other paths of the same kernel on the same state (reported beside it, never as `value`)
    ev_regions = kernel_regions(use_graph)
    other_paths = {}
    for name, mode, what in (() if args.lean else (
            ("open_loop_direct_queue", 2, "OPEN LOOP: the library's own queue, no L2 write-back between the launches of a region; a step's "
                                          "outputs are visible after the region only (FLEET_LAUNCH_DIRECT: a recorded tape, an open-loop rollout)"),
            ("closed_loop_hip_graph" if args.steps >= graph_len else "closed_loop_hip_eager", 1 if args.steps >= graph_len else 0,
             "CLOSED LOOP: HIP's launches, an agent-scope release (L2 write-back) at every kernel boundary -- every step's outputs are "
             "visible before the next step starts: what fleet_step_dev and every Gym / SB3 loop get"))):
        if (mode >= 2) != (use_graph >= 2):
            try:
                other_paths[name] = {"kernel_ms": kernel_ms(mode), "what": what}
            except Exception as ex:  # (no queue to be had: see launch_note)
                other_paths[name] = {"kernel_ms": None, "what": f"unavailable: {ex}"}
    # ---- what the run did: the workload's invariants, and the same launches in the OTHER episode phase ---------------------------
    # The invariants are taken in the STAGGERED state (the stationary mixture of episode ages: the same whatever the shape, the step
    # count and the prime time of the run were), over a window of W launches, envs that end an episode inside it left out.
    def workload_invariants(W=16):
        try:
            before = [(g.batch.get("episodes"), g.batch.get("rf_cycles"), g.batch.get("rf_stack")) for g in groups]
            run(W)
            sync()
            pushes = closures = evsteps = 0
            for g, (e0, c0, s0) in zip(groups, before):
                keep = g.batch.get("episodes") == e0
                dc = (g.batch.get("rf_cycles").astype(np.int64) - c0)[keep].sum()
                ds = (g.batch.get("rf_stack").astype(np.int64) - s0)[keep].sum()
                closures += int(dc)
                pushes += int(ds + 2 * dc)  # a push adds a point, a full cycle removes two (a half cycle one: an upper bound, exact without them)
                evsteps += int(keep.sum()) * g.N * W
            return {"push_fraction": pushes / max(evsteps, 1), "closure_fraction": closures / max(evsteps, 1), "window_steps": W,
                    "ev_steps": evsteps, "episode_phase": "staggered",
                    "what": "share of the EV-steps that push a rainflow reversal point / close a rainflow cycle"}
        except Exception as ex:  # (an older library run beside the tree by the A/B scripts: no such fields)
            return {"push_fraction": None, "closure_fraction": None, "what": f"unavailable: {ex}"}

    if args.lean:
        invariants, other_phase_ms = {"push_fraction": None, "closure_fraction": None, "what": "not taken (--lean)"}, None
    elif args.phase == "locked":
        stagger()
        run(args.warmup + 32)
        sync()
        invariants = workload_invariants()
        other_phase_ms = kernel_ms(use_graph)
    else:
        invariants = workload_invariants()
        for g in groups:
            g.batch.reset_dev(g.obs.data_ptr())
        run(args.warmup + 7)
        sync()
        other_phase_ms = kernel_ms(use_graph)
    w = torch.tensor([walls, walls_closed], device=cdev, dtype=torch.float64)
    if launched:
        dist.all_reduce(w, op=dist.ReduceOp.MAX)  # every region: the slowest rank's time
    walls, walls_closed = ([float(x) for x in row] for row in w.cpu())
    order = sorted(range(reps), key=lambda i: walls[i])
    wall = walls[order[reps // 2]]
    ev_ms = sorted(ev_regions, key=max)[len(ev_regions) // 2]
    # every rank's own kernel time per step (N > 1: the line reports rank 0's roofline; the spread over the ranks beside it)
    k_rank = torch.tensor([max(ev_ms) / args.steps], device=cdev, dtype=torch.float64)
    k_all = [k_rank.clone() for _ in range(world)] if launched else [k_rank]
    if launched:
        dist.all_gather(k_all, k_rank)
    k_ranks = [float(x.item()) for x in k_all]
    # ---- logging collective: one all-gather of finished-episode returns / lengths (RCCL over xGMI when N > 1) ------------------
    t1 = time.perf_counter()
    off = 0
    for g in groups:
        g.batch.get_dev("last_ep_return", ret[off:off + g.E].data_ptr())  # device-side unpack, no host round trip
        g.batch.get_dev("last_ep_len", ln[off:off + g.E].data_ptr())
        off += g.E
    sync()
    gather_how = "torch.distributed all_gather_into_tensor" if launched else None
    r_all = n_all = None
    if args.gather == "capi" and launched and args.backend == "nccl" and len(groups) == 1:
        # the same gather without PyTorch's collectives: rank 0 makes the RCCL id, the launcher's store hands it round, every rank
        # creates its communicator and the library issues one ncclAllGather on the handle's stream
        # Every stage is agreed on by ALL ranks before anybody acts on it (an all_reduce(MIN) of an ok flag): a rank that failed
        # alone would otherwise sit in the torch gather while the others sit in ncclAllGather.
        import ctypes as C

        from fleetrl_amd import _capi

        def all_ok(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], device=cdev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        lib = groups[0].batch.lib
        comm = C.c_void_p()
        why = None
        try:
            uid = (C.c_char * 128)()
            ok = True
            if rank == 0 and lib.fleet_rccl_unique_id(uid) != _capi.OK:
                ok, why = False, lib.fleet_last_error(None).decode()
            box = [bytes(uid) if ok else None]
            dist.broadcast_object_list(box, src=0)
            ok = box[0] is not None
            if ok and lib.fleet_rccl_comm_create(local_rank, world, rank, box[0], C.byref(comm)) != _capi.OK:
                ok, why = False, lib.fleet_last_error(None).decode()
            if all_ok(ok):  # every rank holds a communicator: the collective can be entered
                out = torch.zeros((world, 2, E), device=dev, dtype=torch.float64)
                ok = lib.fleet_gather_episode_stats_rccl(groups[0].batch.h, comm, world, C.c_void_p(out.data_ptr())) == _capi.OK
                if not ok:
                    why = lib.fleet_last_error(groups[0].batch.h).decode()
                sync()
                if all_ok(ok):
                    r_all, n_all = out[:, 0, :].reshape(-1), out[:, 1, :].reshape(-1).to(torch.int32)
                    gather_how = "fleet_gather_episode_stats_rccl: one ncclAllGather issued by the C ABI"
            if r_all is None:
                gather_how = f"torch.distributed all_gather_into_tensor (the C-ABI gather failed on some rank: {why or 'another rank'})"
        finally:
            if comm:
                lib.fleet_rccl_comm_destroy(comm)
    if r_all is None:
        r_all, n_all = gather_episode_stats(ret.to(cdev), ln.to(cdev), equal_shards=True)
    barrier()
    gather_ms = (time.perf_counter() - t1) * 1e3
    check = os.environ.get("FLEET_BENCH_NO_ERRCHECK") != "1"  # diagnostics only (tools/ab_noerr.sh: ablation builds)
    for g in groups:
        if check:
            g.batch.check_errors()

    out = None
    if rank == 0:
        # dominant kernel: fleet_step_kernel, one launch per step and group.  Average launch duration: HIP events on the
        # kernels' streams around the timed region -- or, for the library's own queue, the dispatch timestamps of the region's
        # first and last packet (start of the first launch to end of the last) -- divided by the launches (the gaps between
        # the launches are included -> slightly conservative; rocprofv3's per-kernel average is in profiles/).  With several groups their streams run
        # concurrently: the launch "duration" is the slowest stream's time per step, and the bytes are all groups' bytes.
        k_ms = max(ev_ms) / args.steps
        bytes_launch = sum(g.bytes_step * g.E for g in groups)
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9
        g0 = max(groups, key=lambda g: g.E * g.N)
        per = [float("nan")] if args.lean else g0.batch.time_steps_dev(min(args.steps, 512), g0.tape.data_ptr(), g0.L, g0.obs.data_ptr(),
                                                                        g0.reward.data_ptr(), g0.done.data_ptr())
        # K-steps-per-launch entry (open-loop rollouts), reported aside
        # (an open-loop rollout of K steps consumes K steps of actions: a tape of its own where the single-step tape is shorter)
        K = min(64, max(args.steps, 1))
        ktape = g0.tape
        if g0.L < K:
            gen = torch.Generator(device=dev).manual_seed(7)
            ktape = torch.rand((K, g0.E, g0.N), device=dev, generator=gen, dtype=torch.float32) * 2 - 1
            ktape[torch.rand((K, g0.E, g0.N), device=dev, generator=gen) < 0.15] = 0.0
            torch.cuda.synchronize()
        rsum = torch.empty(g0.E, device=dev, dtype=torch.float64)
        g0.batch.step_many_dev(K, ktape.data_ptr(), g0.obs.data_ptr(), rsum.data_ptr())
        g0.batch.synchronize()
        g0.batch.timer_start()
        many_reps = max(1, min(args.steps, 2048) // K)
        for _ in range(many_reps):
            g0.batch.step_many_dev(K, ktape.data_ptr(), g0.obs.data_ptr(), rsum.data_ptr())
        many_ms = g0.batch.timer_stop()
        del ktape
        if check:
            g0.batch.check_errors()
        fleets = "+".join(g.use_case for g in groups)
        traffic, traffic_source = (committed_traffic(args.config, E, N, launch_mode) if not (args.deg or args.use_case)
                                   else (None, "diagnostic override of the workload"))
        graph_used = use_graph == 1 and args.steps >= graph_len
        n_queues = max(g.batch.direct_queues() for g in groups) if use_graph >= 2 else 0
        if use_graph < 2:
            launch_desc = launch_note + (f"hipGraph of {graph_len} launches" if graph_used else "eager") + \
                          ": HIP's launches, an agent-scope release (L2 write-back) at every kernel boundary; CLOSED LOOP: every step's " \
                          "outputs are visible before the next step starts"
        else:
            launch_desc = ("AQL packets written by the library into " + ("an HSA queue" if n_queues < 2 else "two HSA queues") + " of its own "
                           "(FLEET_LAUNCH_DIRECT): every launch invalidates the per-CU caches, only the last launch of a region writes the "
                           "dies' L2s back (the env state stays there from step to step; every region records the queue's workgroup -> die "
                           "placement and every launch checks the die it runs on against it); OPEN LOOP: a step's outputs are visible after "
                           "the region only (tape replay)" +
                           ("; the batch as two ranges of workgroups, one in-order chain per queue" if n_queues == 2 else ""))

        def path_line(kms):
            a = bytes_launch / (kms * 1e-3) / 1e9
            return {"kernel_ms": kms, "achieved": a, "frac": a / HBM_PEAK_GBS}

        paths = {}
        for name, v in other_paths.items():
            paths[name] = dict(path_line(v["kernel_ms"]), what=v["what"]) if v["kernel_ms"] else v
        out = {
            "metric": "env-steps/sec (num_envs x EVs batch, 1 launch per step)",
            "value": world * E * args.steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "reps": reps, "ms_per_step_min": min(walls) * 1e3 / args.steps, "ms_per_step_max": max(walls) * 1e3 / args.steps,
            # how `ms_per_step` / `value` were taken (not comparable with rounds 1-2, which timed ONE region with a blocking sync)
            "timing": f"median of {reps} back-to-back regions of exactly {args.steps} launches each, barrier + synchronize on both "
                      "sides, the wait inside a region spins on hipStreamQuery"
                      + ("; a rank's span ends when its own device is idle (MAX over ranks), the closing barrier follows" if launched else ""),
            **({"ms_per_step_incl_closing_barrier": sorted(walls_closed)[reps // 2] * 1e3 / args.steps} if launched else {}),
            "errcheck": check,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{E} envs x {N} EVs per GPU, {fleets} fleet{'s (one third each)' if len(groups) > 1 else ''}, "
                                   f"{'load+pv' if spec['building'] and spec['pv'] else 'price-only'} obs, {spec['deg']} degradation, "
                                   f"{'spot_2021-like prices + fixed feed-in tariff, ' if spec['price_year'] == '2021' else ''}"
                                   f"48 h episodes, random start rows, auto-reset ({spec['what']})"
                                   + ("; NOTE: 2048 envs per GPU are two wavefronts per SIMD -- the launch is latency-bound (0.17 of the "
                                      "roof), a multi-GPU run of this config scales while every GPU idles" if args.config == "c4" else ""),
                       "name": args.config, "envs_per_gpu": E, "evs_per_env": N, "obs_dim": g0.batch.obs_dim,
                       "groups": [{"use_case": g.use_case, "envs": g.E} for g in groups],
                       "launch": launch_desc, "launch_mode": launch_mode, "launches_per_step": len(groups) * max(1, n_queues), "prime_ms": args.prime_ms,
                       "closed_loop": use_graph < 2,  # every step's outputs are visible before the next step starts
                       "action_tape": f"{L} independent rows x {E * N * 4 / 2**20:.2f} MB of float32 actions resident in HBM, replayed cyclically",
                       "episode_phase": (f"staggered: episode ages uniform over the {S_ep} steps of an episode, every launch ends "
                                         f"~1/{S_ep} of the episodes (--phase staggered)" if args.phase == "staggered" else
                                         f"locked: all envs in the same episode step, one launch in {S_ep} resets every env -- the "
                                         "reference's vec env by construction (--phase locked)"),
                       "workload_invariants": invariants,
                       "ev_steps_per_s": world * E * N * args.steps / wall},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name(N, spec["deg"]), "kernel_ms": k_ms,
                         "kernel_ms_ranks_min": min(k_ranks), "kernel_ms_ranks_max": max(k_ranks),
                         "kernel_src_sha": kernel_source_sha(),
                         "algorithmic_bytes_per_env_step": bytes_launch / E, "bytes_per_launch": bytes_launch,
                         "kernel_ms_event_pair_per_launch": float(np.mean(per)),
                         "launch_mode": launch_mode,
                         "note": "frac = algorithmic bytes per launch / average launch duration / 8 TB/s: the state, the observation rows "
                                 "and the action tape of this shape sit in the 256 MiB Infinity Cache, so this is a rate of algorithmic "
                                 "bytes against the HBM peak, not measured HBM traffic (`traffic`: rocprofv3 counters)",
                         # the same kernel on the same state through the other launch paths (kernel time only; never `value`)
                         "other_launch_paths": paths,
                         "other_phase": (dict(path_line(other_phase_ms), phase="staggered" if args.phase == "locked" else "locked")
                                         if other_phase_ms else None)},
            # K steps per launch: only the last step's observation is part of the result (and written), so the
            # algorithmic bytes per env-step are smaller by the observation row for K-1 of the K steps
            "step_many": {"K": K, "group": g0.use_case, "env_steps_per_s": g0.E * K * many_reps / (many_ms * 1e-3),
                          "algorithmic_bytes_per_env_step": g0.bytes_step - 4 * g0.batch.obs_dim * (K - 1) / K,
                          "GBps": (g0.bytes_step - 4 * g0.batch.obs_dim * (K - 1) / K) * g0.E * K * many_reps / (many_ms * 1e-3) / 1e9},
            "log_gather_ms": gather_ms, "collective_backend": (args.backend if launched else None), "log_gather": gather_how,
            "episodes_gathered": int((n_all > 0).sum().item()),
        }
        if not args.no_host_path and world == 1:
            out["host_path"] = host_path(g0, groups)
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is a single-GPU-run item (the other ranks would idle)
            out["cpu_baseline"] = cpu_baseline(g0.params, g0.tables, g0.tf, N)
    for g in groups:
        g.batch.close()
    if launched:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


def host_path(g0, groups, budget_s: float = 1.0):
    """The host-pointer path, PCIe both ways, as an SB3 loop sees it (never the headline value): `fleet_step_host` through
    FleetBatch.step with NumPy buffers, bounded to about a second per variant.  Primary figure: the DEFAULT of FleetBatch.step /
    FleetVecEnv (`copy=True`: a private copy of the observations per step, like the reference's env returns); beside it
    `copy=False` (the pinned transfer buffer itself is handed out)."""
    acts = g0.tape[:8].cpu().numpy()

    def run(copy):
        g0.batch.step(acts[0], copy=copy)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < budget_s:
            g0.batch.step(acts[n % len(acts)], copy=copy)
            n += 1
        dt = time.perf_counter() - t0
        return g0.E * n / dt, dt * 1e3 / n

    v_copy, ms_copy = run(True)
    v_pin, ms_pin = run(False)
    return {"env_steps_per_s": v_copy, "ms_per_step": ms_copy, "envs": g0.E, "copy_obs": True,
            "no_copy": {"env_steps_per_s": v_pin, "ms_per_step": ms_pin, "copy_obs": False},
            "what": "fleet_step_host: actions host->device, one launch, observations/rewards/dones device->host, synchronous"}


if __name__ == "__main__":
    main()
