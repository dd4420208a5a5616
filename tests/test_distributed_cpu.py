"""N>1 path on CPU: world_size-2 `gloo` run of the sharding + logging gather, with the CPU oracle standing in for the
GPU engine (tests may use the oracle).  The 2-rank run must equal the 1-rank run env for env (envs are independent
and the start sampler is keyed by the GLOBAL env id)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fleetrl_amd.distributed import gather_episode_stats, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything_once():
    for total, world in ((16384, 8), (10, 3), (7, 8), (4096, 1)):
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _run_shard(rank, world, total_envs, steps, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from golden_util import load_trace
    from fleetrl_amd.params import make_params
    from oracle.fleet_oracle import OracleBatch

    g = load_trace("lmd5_price_linear")
    lo, hi = shard_range(total_envs, world, rank)
    cfg = dict(g.cfg)
    cfg["time_picker"] = "random"
    from fleetrl_amd.config import resolve_config

    p = make_params(resolve_config(cfg), g.tables, hi - lo, extrema=g.extrema, start_range=(0, g.tables.T - g.ep_steps - 50),
                    env_id_offset=lo, seed=11)
    eng = OracleBatch(p, g.tables, g.time_feat)
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, size=(steps, total_envs, g.N)).astype(np.float32)  # same tape on every rank, sliced by env id
    eng.reset()
    for s in range(steps):
        eng.step(acts[s, lo:hi])
    r_all, n_all = gather_episode_stats(eng.get("last_ep_return"), eng.get("last_ep_len"))
    if total_envs % world == 0:  # the one-collective path bench.py uses (equal shards) gives the same arrays
        r2, n2 = gather_episode_stats(eng.get("last_ep_return"), eng.get("last_ep_len"), equal_shards=True)
        np.testing.assert_array_equal(r2, r_all)
        np.testing.assert_array_equal(n2, n_all)
    if rank == 0:
        np.savez(os.path.join(out_dir, f"w{world}.npz"), r=r_all, n=n_all)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_run_equals_single_rank(tmp_path):
    total, steps = 10, 200  # 48 h episodes of 192 steps: every env finishes one episode
    mp.spawn(_run_shard, args=(1, total, steps, 29511, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run_shard, args=(2, total, steps, 29512, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "w1.npz"), np.load(tmp_path / "w2.npz")
    assert a["n"].shape == (total,) and (a["n"] == 192).all()
    np.testing.assert_array_equal(a["n"], b["n"])
    np.testing.assert_array_equal(a["r"], b["r"])
