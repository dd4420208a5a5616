"""bench.py end to end on the GPU box: the JSON contract at N = 1, and the N = 2 control flow (sharding by rank, barriers,
max-over-ranks timing, the logging all-gather) with two ranks sharing ONE GPU over gloo -- RCCL refuses two ranks on one
device, and the pool has no multi-GPU box for the builder.  The RCCL branch itself (`--backend nccl`, the default) runs here
with ONE rank under the launcher: process-group set-up on the device, barriers, the max-reduce of the region times and the
all-gather of episode returns all go through RCCL.  `python bench.py --gpus N` without a launcher starts its own ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_json_contract_single_gpu():
    res = subprocess.run([sys.executable, "bench.py", "--steps", "300", "--warmup", "20", "--envs-per-gpu", "512", "--prime-ms", "50"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    d = _json_line(res.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["reps"] >= 3 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    assert "traffic_source" in d["roofline"]
    assert d["n_gpus"] == 1 and d["steps"] == 300 and d["unit"] == "env-steps/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel"] == "fleet_step_kernel<G=64,DEG=rainflow,MULTI=false,WIDE=false>"
    assert abs(d["value"] - 512 * 300 / (d["ms_per_step"] * 300 * 1e-3)) / d["value"] < 1e-9
    # the default: the library's own queue as an open-loop replay, labelled as such, with the closed-loop path (HIP's launches) of the
    # same kernel beside it, the other episode phase's figure, and what the run did
    assert d["config"]["launch_mode"] == "direct" and "FLEET_LAUNCH_DIRECT" in d["config"]["launch"] and "OPEN LOOP" in d["config"]["launch"]
    assert d["config"]["closed_loop"] is False and r["launch_mode"] == "direct"
    (name, cl), = r["other_launch_paths"].items()
    assert name == "closed_loop_hip_graph" and "CLOSED LOOP" in cl["what"] and cl["kernel_ms"] > r["kernel_ms"]
    assert abs(cl["frac"] - cl["achieved"] / 8000.0) < 1e-12
    assert r["other_phase"]["phase"] == "staggered" and r["other_phase"]["kernel_ms"] > 0 and "locked" in d["config"]["episode_phase"]
    inv = d["config"]["workload_invariants"]
    assert 0.10 < inv["push_fraction"] < 0.40 and 0.03 < inv["closure_fraction"] < inv["push_fraction"] and inv["ev_steps"] > 0
    assert d["config"]["action_tape"].startswith("32 independent rows")
    assert "timing" in d and d["errcheck"] is True and r["kernel_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert d["episodes_gathered"] == 512


def test_bench_through_a_replayed_hipgraph():
    res = subprocess.run([sys.executable, "bench.py", "--launch", "graph", "--steps", "300", "--warmup", "20", "--envs-per-gpu", "512",
                          "--prime-ms", "50", "--no-cpu-baseline", "--no-host-path"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    d = _json_line(res.stdout)
    assert d["config"]["launch"].startswith("hipGraph of 64 launches") and d["config"]["launch_mode"] == "graph" and d["roofline"]["kernel_ms"] > 0
    assert d["config"]["closed_loop"] is True and set(d["roofline"]["other_launch_paths"]) == {"open_loop_direct_queue"}


def test_bench_workload_invariants_agree_across_shapes():
    """The shapes the roofline claims are made on run the SAME workload: the share of EV-steps that push a rainflow reversal point /
    close a cycle agrees within 5 % between the headline shape, 16384 x 50 and the c5 shard (the short cyclic tapes of
    round 5 had made the large shapes another workload -- batteries saturated, the push dropped out)."""
    inv = {}
    for name, extra in (("c3", ["--config", "c3"]), ("16384x50", ["--config", "c3", "--envs-per-gpu", "16384"]), ("c5", ["--config", "c5"])):
        res = subprocess.run([sys.executable, "bench.py", *extra, "--steps", "64", "--warmup", "20", "--prime-ms", "50", "--reps", "3",
                              "--no-cpu-baseline", "--no-host-path"], cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-2000:]
        d = _json_line(res.stdout)
        inv[name] = d["config"]["workload_invariants"]
        assert d["config"]["action_tape"].startswith("32 independent rows")
    for k in ("push_fraction", "closure_fraction"):
        # (c5 mixes three fleet types with other schedules: its own level, compared at 15 %; the two caretaker shapes at 5 %)
        assert abs(inv["16384x50"][k] - inv["c3"][k]) <= 0.05 * inv["c3"][k], (k, inv)
        assert abs(inv["c5"][k] - inv["c3"][k]) <= 0.15 * inv["c3"][k], (k, inv)


def test_bench_two_ranks_on_one_gpu_over_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "400", "--warmup", "50", "--envs-per-gpu", "768",
           "--backend", "gloo", "--device-index", "0", "--prime-ms", "50"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)  # the launcher itself never touches the GPU
    assert res.returncode == 0, res.stderr[-3000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 768
    assert d["episodes_gathered"] == 2 * 768  # both ranks' finished episodes arrived through the gather
    assert abs(d["value"] - 2 * 768 * 400 / (d["ms_per_step"] * 400 * 1e-3)) / d["value"] < 1e-9  # whole-job aggregate
    assert "cpu_baseline" not in d and "host_path" not in d  # single-GPU-run items
    # N > 1: a rank's span ends when its own device is idle (MAX over ranks); the reading with the closing barrier inside is beside it
    assert d["ms_per_step_incl_closing_barrier"] >= d["ms_per_step"] and "closing barrier follows" in d["timing"]


def test_bench_plain_python_starts_its_own_ranks():
    """What the driver runs for N > 1 when it has no launcher of its own: `python bench.py --gpus 2`.  bench.py starts
    torch.distributed.run as a child process before anything touches the GPU and relays the single JSON line."""
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "200", "--warmup", "20", "--envs-per-gpu", "512",
           "--backend", "gloo", "--device-index", "0", "--prime-ms", "50"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 2 and d["collective_backend"] == "gloo" and d["episodes_gathered"] == 2 * 512
    assert d["log_gather_ms"] > 0 and d["ms_per_step"] > 0


def test_bench_rccl_branch_with_one_rank():
    """`--backend nccl` (RCCL) under the launcher with world size 1: init_process_group on the device, dist.barrier,
    all_reduce(MAX) of the region times and the all-gather of episode returns execute on RCCL."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", "bench.py", "--gpus", "1", "--steps", "200", "--warmup", "20", "--envs-per-gpu", "512",
           "--backend", "nccl", "--prime-ms", "50", "--no-cpu-baseline", "--no-host-path"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 1 and d["collective_backend"] == "nccl" and d["episodes_gathered"] == 512
    assert d["ms_per_step_incl_closing_barrier"] >= d["ms_per_step"]  # (the closing barrier of a region is an RCCL collective here)
    # ... and the same gather issued by the C ABI itself (one ncclAllGather on the handle's stream, --gather capi)
    res = subprocess.run(cmd + ["--gather", "capi"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _json_line(res.stdout)
    assert d["episodes_gathered"] == 512 and d["log_gather"].startswith("fleet_gather_episode_stats_rccl"), d["log_gather"]


def test_bench_eight_ranks_on_one_gpu_over_gloo():
    """The driver's 8-GPU run executes `shard_range`, env-id offsets, the build lock and the gather of eight shards for the first
    time; here the same control flow with eight ranks sharing the one GPU (gloo: RCCL refuses several ranks per device), as
    plain `python bench.py --gpus 8` (self-launch) on BASELINE configs[3]'s fleet type: every rank's finished episodes arrive."""
    cmd = [sys.executable, "bench.py", "--gpus", "8", "--config", "c4", "--steps", "260", "--warmup", "20", "--envs-per-gpu", "256",
           "--backend", "gloo", "--device-index", "0", "--prime-ms", "20", "--reps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 8 and d["collective_backend"] == "gloo" and d["config"]["envs_per_gpu"] == 256
    assert d["episodes_gathered"] == 8 * 256
    assert abs(d["value"] - 8 * 256 * 260 / (d["ms_per_step"] * 260 * 1e-3)) / d["value"] < 1e-9  # whole-job aggregate
