"""CPU-side host logic: config resolution, params assembly, spaces, the C ABI's shape (no GPU compute)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from golden_util import TRACE_NAMES, load_trace, params_for
from fleetrl_amd import _capi
from fleetrl_amd.config import MANDATORY_KEYS, resolve_config
from fleetrl_amd.params import obs_dim, picker_range, static_start_row, validate_supported
from fleetrl_amd.spaces import Box, observation_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    return dict(load_trace("ct5_both_rainflow").cfg)


def test_mandatory_keys_raise_keyerror_like_the_reference():
    for k in ("episode_length", "target_soc", "use_case", "init_soh"):
        cfg = _cfg()
        del cfg[k]
        with pytest.raises(KeyError):
            resolve_config(cfg)
    assert set(MANDATORY_KEYS) <= set(_cfg())


def test_use_case_sets_battery_and_grid_sizing():
    # LoadCalculation._import_company (load_calculation.py:15-60) and specify_company_and_battery_size (:1041-1060)
    for uc, cap, evse, batt in (("lmd", 60.0, 11, 60), ("ut", 50.0, 22, 50), ("ct", 16.7, 4.6, 16.7)):
        cfg = _cfg()
        cfg["use_case"] = uc
        rc = resolve_config(cfg)
        assert rc.init_battery_cap == cap
        grid, e, b = rc.company(1, 100.0)
        assert (e, b) == (evse, batt) and grid == max(100.0 * 1.1, 100.0 + 0.5 * evse)
    cfg = _cfg()
    cfg["use_case"] = "ut"
    assert resolve_config(cfg).company(5, 100.0)[0] == 1000  # quirk Q14
    cfg["use_case"] = "nope"
    with pytest.raises(TypeError):
        resolve_config(cfg)


def test_price_multiplier_scaling_and_ignore_flags():
    cfg = _cfg()
    rc = resolve_config(cfg)
    assert rc.price_multiplier == 3.33 * (60 / 16.7)
    cfg.update(ignore_price_reward=True, ignore_invalid_penalty=True, ignore_overcharging_penalty=True, ignore_overloading_penalty=True)
    rc = resolve_config(cfg)
    assert (rc.price_multiplier, rc.penalty_invalid_action, rc.penalty_overcharging, rc.penalty_overloading) == (0, 0, 0, 0)
    cfg.update(spot_markup=5, spot_mul=2.0, feed_in_ded=0.5)
    rc = resolve_config(cfg)
    assert (rc.fixed_markup, rc.variable_multiplier, rc.feed_in_deduction) == (5, 2.0, 0.5)


@pytest.mark.parametrize("name", TRACE_NAMES)
def test_obs_dim_matches_reference(name):
    g = load_trace(name)
    assert obs_dim(g.rc, g.N) == int(g.sc_obs_dim)


def test_obs_dim_formulas_of_the_survey():
    cfg = _cfg()
    rc = resolve_config(cfg)  # load + pv, aux
    assert obs_dim(rc, 5) == 73 and obs_dim(rc, 50) == 388 and obs_dim(rc, 200) == 1438
    cfg.update(include_building=False, include_pv=False)
    rc = resolve_config(cfg)
    assert obs_dim(rc, 1) == 32 and obs_dim(rc, 5) == 60


def test_unsupported_flag_combinations_are_rejected():
    for upd in (dict(include_price=False), dict(normalize_in_env=True, include_pv=True, include_building=False),
                dict(init_soh=0.95)):
        cfg = _cfg()
        cfg.update(upd)
        with pytest.raises(ValueError):
            validate_supported(resolve_config(cfg))


def test_picker_ranges():
    g = load_trace("lmd1_price_linear")
    # a full-year table starting 2020-01-01 00:00: build a tiny stand-in with the same date axis
    from fleetrl_amd.synth import synth_tables

    tb = synth_tables("lmd", 1, seed=3)
    assert static_start_row(tb) == 96 + 19 * 4  # "01/02/2021 19:00" re-based to 2020 -> Jan 2nd 19:00 (row 172)
    cfg = dict(g.cfg)
    for picker, want in (("static", (172, 172)), ("random", (0, tb.T - 1 - 60 * 96)),
                         ("eval", (tb.T - 1 - 60 * 96, tb.T - 1 - 2 * cfg["episode_length"] * 4))):
        cfg["time_picker"] = picker
        assert picker_range(resolve_config(cfg), tb) == want


def test_spaces():
    lo, hi = observation_bounds(7, True)
    assert lo.min() == 0 and hi.max() == 1
    lo, hi = observation_bounds(7, False)
    assert np.isinf(lo).all() and np.isinf(hi).all()
    b = Box(low=-1, high=1, shape=(5,), dtype=np.float32)
    assert b.shape == (5,) and b.dtype == np.float32 and b.contains(b.sample())


def test_c_abi_struct_layout_matches_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fleet_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu", '
                   'sizeof(FleetParams), offsetof(FleetParams, seed), offsetof(FleetParams, dt), offsetof(FleetParams, max_grid), '
                   'sizeof(FleetTables));return 0;}')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    P = _capi.FleetParams
    assert got == [ctypes.sizeof(P), P.seed.offset, P.dt.offset, P.max_grid.offset, ctypes.sizeof(_capi.FleetTablesC)]


def test_library_exports_every_symbol_the_header_declares():
    """No compute calls (no GPU here): the in-tree libfleet_hip.so must load and export exactly the header's entry points."""
    from fleetrl_amd import build

    path = build.build()
    hdr = open(os.path.join(ROOT, "include", "fleet_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(fleet_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_capi.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(path)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_no_cpu_fallback_without_a_gpu():
    """On a box without a GPU the product must fail loudly instead of computing on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fleetrl_amd.batch import FleetBatch, FleetHipError

    g = load_trace("ct2_pv_nodeg")
    with pytest.raises(FleetHipError) as ei:
        FleetBatch(params_for(g), g.tables, g.time_feat)
    assert ei.value.status in (_capi.ERR_NODEVICE, _capi.ERR_HIP)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fleetrl_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in text and "import oracle" not in text and "fleet_oracle" not in text, f


def _create_status(p, tables, time_feat):
    """fleet_create's argument check runs before the device is touched, so what it rejects can be tested without a GPU."""
    lib = _capi.load_library()
    tc, keep = _capi.pack_tables(tables, time_feat)
    h = ctypes.c_void_p()
    rc = lib.fleet_create(ctypes.byref(p), ctypes.byref(tc), 0, ctypes.byref(h))
    msg = lib.fleet_last_error(None).decode()
    if rc == _capi.OK:
        lib.fleet_destroy(h)
    del keep
    return rc, msg


def test_create_accepts_long_episodes_in_rainflow_mode():
    """The reference accepts any `episode_length` (time_config.py:1-24, fleet_environment.py:355).  The rainflow stack size
    travels in a 26-bit field of the hot record, so only episodes beyond 67 million steps are rejected (round 2 stopped at
    8188 steps = 85 days); the config layer has no limit of its own any more."""
    g = load_trace("ct5_both_rainflow")
    p = params_for(g)
    p.episode_steps = 24 * 4 * 365  # a whole year at 15 min: the status then only depends on the device
    rc, msg = _create_status(p, g.tables, g.time_feat)
    assert rc != _capi.ERR_INVALID, msg
    p.episode_steps = (1 << 26) - 2
    rc, msg = _create_status(p, g.tables, g.time_feat)
    assert rc == _capi.ERR_INVALID and "67 million" in msg
    p.deg_mode = _capi.DEG_LINEAR  # no stack: any length goes
    rc, msg = _create_status(p, g.tables, g.time_feat)
    assert rc != _capi.ERR_INVALID
    cfg = dict(g.cfg)
    cfg["episode_length"] = 24 * 90
    validate_supported(resolve_config(cfg))


def test_create_rejects_rainflow_rows_beyond_a_32_bit_offset():
    """The kernels address an EV's rainflow row as (the rows of its env) + a 32-bit byte offset: num_cars x episode length
    combinations whose rows would exceed 4 GiB per env are refused by name, not silently wrapped."""
    from fleetrl_amd.params import make_params, time_features
    from fleetrl_amd.synth import synth_tables

    cfg = dict(_cfg(), use_case="ut", episode_length=24)
    tb = synth_tables("ut", 16, seed=3)
    p = make_params(resolve_config(cfg), tb, 2, seed=0)
    tf = time_features(tb)
    p.episode_steps = 40_000_000  # 16 EVs x 40 M stack words x 8 B = 5.1 GB per env
    rc, msg = _create_status(p, tb, tf)
    assert rc == _capi.ERR_INVALID and "4 GiB" in msg
    p.episode_steps = 24 * 4 * 365
    rc, msg = _create_status(p, tb, tf)
    assert rc != _capi.ERR_INVALID, msg


def test_create_range_checks_the_irregular_grid_tables():
    """finish_row / lookahead_row entries index the tables on the host (tail rows) and on the device: out-of-range entries
    are rejected instead of read."""
    from golden_util import load_rt_trace

    g = load_rt_trace("lmd1_both_irregular")
    irr = g.tables.meta.get("irregular")
    p = params_for(g)
    assert g.tables.meta.get("irregular") is not None
    rc, _ = _create_status(p, g.tables, g.time_feat)
    assert rc != _capi.ERR_INVALID
    keep = g.tables.meta["irregular"]["finish_row"].copy()
    g.tables.meta["irregular"]["finish_row"][3] = g.tables.T + 7
    rc, msg = _create_status(p, g.tables, g.time_feat)
    assert rc == _capi.ERR_INVALID and "finish_row" in msg
    g.tables.meta["irregular"]["finish_row"][:] = keep
    g.tables.meta["irregular"]["lookahead_row"][5, 0] = g.tables.T
    rc, msg = _create_status(p, g.tables, g.time_feat)
    assert rc == _capi.ERR_INVALID and "lookahead_row" in msg
    assert irr is None or True


def test_create_rejects_a_negative_log_capacity():
    g = load_trace("ct2_pv_nodeg")
    p = params_for(g)
    p.log_data, p.log_capacity = 1, -5
    rc, msg = _create_status(p, g.tables, g.time_feat)
    assert rc == _capi.ERR_INVALID and "log_capacity" in msg


def test_product_build_takes_no_flags_from_the_environment(monkeypatch, tmp_path):
    """An environment variable must not be able to turn libfleet_hip.so into a diagnostic build with the product's file name
    (VERDICT r3): build() refuses FLEET_EXTRA_HIPCC_FLAGS, and build_variant() refuses the product's path."""
    from fleetrl_amd import build

    monkeypatch.setenv("FLEET_EXTRA_HIPCC_FLAGS", "-DFLEET_STAMPS")
    with pytest.raises(RuntimeError, match="no longer honoured"):
        build.build(force=True)
    monkeypatch.delenv("FLEET_EXTRA_HIPCC_FLAGS")
    with pytest.raises(ValueError, match="must not overwrite"):
        build.build_variant(build.lib_path(), ["-DFLEET_STAMPS"])


def test_direct_split_plan_never_overflows_the_packed_first_workgroup():
    """A run on the library's own queues may cover the grid with two ranges of workgroups; the second range's first workgroup travels in
    16 bits of a kernel argument (fleet_kernels.hip `p_N`).  ADVICE r5: nothing bounded it -- grids above ~131 000 workgroups would have
    wrapped the field, stepped the first half twice and the second half never.  The plan is a pure function of the grid."""
    lib = _capi.load_library()
    pg = (ctypes.c_uint32 * 2)()

    def plan(grid, split=1):
        parts = lib.fleet_direct_split_plan(grid, split, pg)
        return parts, pg[0], pg[1]

    assert plan(4096, 0) == (1, 4096, 0)                 # not asked to split
    assert plan(15) == (1, 15, 0)                        # too small to split
    assert plan(16) == (2, 8, 8)
    assert plan(4096) == (2, 2048, 2048)
    assert plan(1027) == (2, 520, 507)                   # the first range is a multiple of 8: env e keeps its die in the second
    for grid in (131040, 131056, 131057):                # the largest grids whose first range (65 528 workgroups) still fits 16 bits
        parts, a, b = plan(grid)
        assert parts == 2 and a % 8 == 0 and a <= 0xFFFF and a + b == grid
    for grid in (131058, 131072, 131080, 262144, 1 << 20, (1 << 31) - 1):  # beyond: ONE range, never a wrapped field
        assert plan(grid) == (1, grid, 0)


def test_library_and_code_object_carry_the_same_source_hash():
    """fleet_direct_open refuses a code object that was not compiled from the library's sources (both carry `FLEET_SRC_SHA`)."""
    from fleetrl_amd import build

    build.build()
    sha = build.source_sha().encode()
    assert sha in open(build.lib_path(), "rb").read()
    assert sha in open(build.code_object_path(), "rb").read()
