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
