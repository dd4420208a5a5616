"""bench.py helpers that carry documented numbers, checked on CPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_algorithmic_bytes_match_the_survey_figures():
    # C3/C4: 50 EVs, load+pv+aux (obs_dim 388), rainflow, 48 h episodes -> ~4.83 kB per env-step
    b = bench.algorithmic_bytes_per_env_step(50, 388, 2 * 9 + 2 * 5, True, 192)
    assert abs(b - (82 * 50 + 329 + 8 * 50 * 96.5 / 96)) < 1e-9 and 4800 < b < 4850
    # C5: 200 EVs -> ~18.3 kB
    b = bench.algorithmic_bytes_per_env_step(200, 1438, 28, True, 192)
    assert 18200 < b < 18400
    # C2: 5 EVs, price-only (obs_dim 60), linear degradation -> 647 B
    assert bench.algorithmic_bytes_per_env_step(5, 60, 18, False, 192) == 82 * 5 + 237


def test_bench_config_resolves_and_is_the_baseline_workload():
    from fleetrl_amd.config import resolve_config

    rc = resolve_config(bench.bench_config(4096, 50, "ct"))
    assert rc.is_caretaker and rc.include_building and rc.include_pv and rc.aux and not rc.normalize_in_env
    assert rc.deg_mode == 2 and rc.episode_length == 48 and rc.time_picker == "random"


def test_committed_traffic_profile_is_well_formed():
    path = os.path.join(ROOT, "profiles", "r01_traffic_step_kernel.json")
    t = json.load(open(path))
    assert t["hbm_bytes_per_launch"] == t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"]
    assert t["hbm_read_bytes_per_launch"] == t["FETCH_SIZE"] * 1024 * 2  # gfx950 read-side correction


def test_kernel_name_follows_the_launcher():
    assert bench.kernel_name(50, "rainflow") == "fleet_step_kernel<G=64,DEG=rainflow,MULTI=false,WIDE=false>"
    assert bench.kernel_name(200, "rainflow") == "fleet_step_kernel<G=256,DEG=rainflow,MULTI=false,WIDE=false>"  # four wavefronts per env
    assert bench.kernel_name(100, "rainflow") == "fleet_step_kernel<G=128,DEG=rainflow,MULTI=false,WIDE=false>"
    assert bench.kernel_name(300, "rainflow") == "fleet_step_kernel<G=64,DEG=rainflow,MULTI=false,WIDE=true>"
    assert bench.kernel_name(5, "linear") == "fleet_step_kernel<G=8,DEG=linear,MULTI=false,WIDE=false>"


def test_committed_traffic_only_counts_for_the_kernel_it_was_profiled_on(tmp_path, monkeypatch):
    """`roofline.traffic` is null unless the committed profile carries the hash of the current kernel sources and the
    shape and launch mode of the run."""
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "fleetrl_amd" / "csrc")
    for f in ("fleet_kernels.hip", "fleet_device.h"):
        (tmp_path / "fleetrl_amd" / "csrc" / f).write_text("v1 " + f)
    sha = bench.kernel_source_sha()
    (tmp_path / "profiles" / "r06_traffic_c3.json").write_text(json.dumps(
        {"kernel_src_sha": sha, "envs": 4096, "evs": 50, "config": "c3", "hbm_bytes_per_launch": 123.0, "launch_mode": "direct"}))
    (tmp_path / "profiles" / "r06_traffic_16384x50.json").write_text(json.dumps(
        {"kernel_src_sha": sha, "envs": 16384, "evs": 50, "config": "c3", "hbm_bytes_per_launch": 456.0, "launch_mode": "direct"}))
    val, src = bench.committed_traffic("c3", 4096, 50, "direct")
    assert val == 123.0 and "profiles/r06_traffic_c3.json" in src  # the JSON line says where the figure comes from
    assert bench.committed_traffic("c3", 16384, 50, "direct")[0] == 456.0  # an override's shape has a profile of its own
    assert bench.committed_traffic("c3", 8192, 50, "direct")[0] is None
    assert bench.committed_traffic("c3", 4096, 50, "publish")[0] is None  # the closed-loop launches write their outputs through: other traffic
    assert bench.committed_traffic("c3", 4096, 50, "graph")[0] is None  # ... and the way the launches reached the GPU
    assert bench.committed_traffic("c5", 8192, 200, "direct")[0] is None
    (tmp_path / "fleetrl_amd" / "csrc" / "fleet_kernels.hip").write_text("v2")
    val, src = bench.committed_traffic("c3", 4096, 50, "direct")
    assert val is None and "no committed" in src


def test_every_baseline_config_has_a_bench_mode():
    assert set(bench.CONFIGS) == {"c2", "c3", "c4", "c5"}
    assert bench.CONFIGS["c3"]["envs"] == 4096 and bench.CONFIGS["c3"]["evs"] == 50
    assert bench.CONFIGS["c5"]["groups"] == ("lmd", "ct", "ut") and bench.CONFIGS["c5"]["envs"] * 8 == 65536
    assert bench.CONFIGS["c4"]["envs"] * 8 == 16384 and bench.CONFIGS["c2"]["deg"] == "linear"
