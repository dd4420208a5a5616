"""ctypes mirror of include/fleet_hip.h (struct layouts, constants) and the libfleet_hip.so loader.

There is deliberately NO CPU fallback: if the HIP library is missing or no GPU is visible, every
product entry point raises (`FleetHipError`).  The CPU restatement under oracle/ is test
infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ABI_VERSION = 10

OK, ERR_INVALID, ERR_HIP, ERR_STATE, ERR_NODEVICE, ERR_UNSUPPORTED = 0, 1, 2, 3, 4, 5
DEG_NONE, DEG_LINEAR, DEG_RAINFLOW = 0, 1, 2
PICK_STATIC, PICK_RANDOM, PICK_EVAL = 0, 1, 2
ACT_F32, ACT_F64 = 0, 1
# fleet_run_tape_dev / fleet_time_regions_begin: how the launches reach the GPU (include/fleet_hip.h FLEET_LAUNCH_*)
LAUNCH_EAGER, LAUNCH_GRAPH, LAUNCH_DIRECT, LAUNCH_DIRECT_ONE_QUEUE = 0, 1, 2, 3
POLICY_UNCONTROLLED, POLICY_DISTRIBUTED, POLICY_NIGHT = 2, 3, 4

DEVERR_OBS_FORMAT, DEVERR_NEG_LIFE, DEVERR_SOH_MISMATCH, DEVERR_DOD_RANGE, DEVERR_TABLE_END, DEVERR_INTERNAL, DEVERR_PLACEMENT = 1, 2, 4, 8, 16, 32, 64

# fleet_get fields: name -> (id, dtype, per_car)
FIELDS = {
    "soc": (0, np.float64, True),
    "hours_left": (1, np.float32, True),
    "soh": (2, np.float64, True),
    "soc_deg": (3, np.float64, True),
    "target_soc": (4, np.float64, True),
    "time_idx": (5, np.int32, False),
    "start_idx": (6, np.int32, False),
    "cashflow": (7, np.float64, False),
    "ep_return": (8, np.float64, False),
    "ep_len": (9, np.int32, False),
    "last_ep_return": (10, np.float64, False),
    "last_ep_len": (11, np.int32, False),
    "rf_len": (12, np.int32, True),
    "fd_cyc": (13, np.float64, True),
    "fd_cal": (14, np.float64, True),
    "sei_l": (15, np.float64, True),
    "error_bits": (16, np.uint32, False),
    "done": (17, np.uint8, False),
    "episodes": (18, np.int32, False),
    "penalty_record": (19, np.float64, False),
    "last_ep_len_f64": (20, np.float64, False),
    "rf_cycles": (21, np.int32, True),
    "rf_stack": (22, np.int32, True),
    "rf_until": (23, np.int32, False),
}

_I32_FIELDS = (
    "abi_version", "struct_bytes", "num_envs", "num_cars", "table_rows", "episode_steps", "price_lookahead",
    "bl_pv_lookahead", "steps_per_hour", "hour_phase", "include_building", "include_pv", "aux", "normalize",
    "is_caretaker", "deg_mode", "picker_mode", "start_lo", "start_hi", "auto_reset", "env_id_offset", "log_data",
    "real_time", "log_capacity",
)
_F64_FIELDS = (
    "dt", "evse_power", "obc_max_power", "batt_cap_nominal", "init_battery_cap", "grid_connection",
    "charging_eff", "discharging_eff", "fixed_markup", "variable_multiplier", "feed_in_deduction",
    "price_multiplier", "penalty_invalid_action", "penalty_overcharging", "clip_overcharging",
    "penalty_overloading", "fully_charged_reward", "target_soc", "target_soc_lunch", "eps", "def_soc",
    "min_laxity", "init_soh", "temperature", "max_time_left", "max_price", "min_price", "max_tariff",
    "min_tariff", "max_building", "max_pv", "max_soc", "max_hours_needed", "max_laxity", "max_evse", "max_grid",
)


class FleetParams(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in _I32_FIELDS] + [("seed", C.c_uint64)] + [(n, C.c_double) for n in _F64_FIELDS])

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


_TABLE_FIELDS = (
    ("there", C.POINTER(C.c_uint8)), ("time_left", C.POINTER(C.c_float)), ("soc_on_return", C.POINTER(C.c_double)),
    ("delu", C.POINTER(C.c_double)), ("tariff", C.POINTER(C.c_double)), ("prc", C.POINTER(C.c_double)),
    ("trc", C.POINTER(C.c_double)), ("load", C.POINTER(C.c_double)), ("pv", C.POINTER(C.c_double)),
    ("hour", C.POINTER(C.c_uint8)), ("minute", C.POINTER(C.c_uint8)), ("month", C.POINTER(C.c_uint8)),
    ("weekday", C.POINTER(C.c_uint8)), ("time_feat", C.POINTER(C.c_float)),
    # irregular time grids (real_time), all NULL / 0 otherwise
    ("dt_row", C.POINTER(C.c_double)), ("finish_row", C.POINTER(C.c_int32)), ("lookahead_row", C.POINTER(C.c_int32)),
    ("lookahead_cols", C.c_int32), ("reserved0", C.c_int32), ("second", C.POINTER(C.c_uint8)),
    ("pick_rows", C.POINTER(C.c_int32)), ("n_pick_rows", C.c_int32), ("reserved1", C.c_int32),
)


class FleetTablesC(C.Structure):
    _fields_ = list(_TABLE_FIELDS)


def pack_tables(tables, time_feat: np.ndarray | None):
    """Returns (FleetTablesC, keepalive list).  Arrays are made contiguous with the ABI's dtypes."""
    keep = []

    def ptr(a, dtype, ctype):
        a = np.ascontiguousarray(a, dtype=dtype)
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(ctype))

    t = FleetTablesC()
    t.there = ptr(tables.there, np.uint8, C.c_uint8)
    t.time_left = ptr(tables.time_left, np.float32, C.c_float)
    t.soc_on_return = ptr(tables.soc_on_return, np.float64, C.c_double)
    for name in ("delu", "tariff", "prc", "trc", "load", "pv"):
        setattr(t, name, ptr(np.nan_to_num(getattr(tables, name), nan=0.0), np.float64, C.c_double))
    for name in ("hour", "minute", "month", "weekday"):
        setattr(t, name, ptr(getattr(tables, name), np.uint8, C.c_uint8))
    if time_feat is not None:
        t.time_feat = ptr(time_feat, np.float32, C.c_float)
    else:
        t.time_feat = C.POINTER(C.c_float)()
    irr = getattr(tables, "meta", {}).get("irregular")
    if irr is not None:  # attached by fleetrl_amd.params.make_params for a real_time config on an irregular grid
        t.dt_row = ptr(irr["dt_row"], np.float64, C.c_double)
        t.finish_row = ptr(irr["finish_row"], np.int32, C.c_int32)
        t.lookahead_row = ptr(irr["lookahead_row"], np.int32, C.c_int32)
        t.lookahead_cols = int(irr["lookahead_row"].shape[1])
        t.second = ptr(irr["second"], np.uint8, C.c_uint8)
        t.pick_rows = ptr(irr["pick_rows"], np.int32, C.c_int32)
        t.n_pick_rows = int(irr["pick_rows"].size)
    return t, keep


class FleetHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"fleet_hip status {status}: {message}")
        self.status = status


_LIB = None
LIB_NAME = "libfleet_hip.so"


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


def load_library():
    """Load the in-tree HIP library; raises (never falls back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    # not built yet (fresh checkout) or older than its sources: compile it now if the ROCm toolchain is here -- still the HIP
    # library, never a substitute for it.  Without hipcc an existing library is used as it is.
    why = ""
    try:
        from . import build as _build

        have_hipcc = True
        try:
            _build.hipcc()
        except RuntimeError:
            have_hipcc = False
        if have_hipcc and _build.needs_build():
            _build.build()
    except Exception as exc:  # compile error: say so instead of reporting a merely "missing" library
        why = f"  Building it failed: {exc}"
        if os.path.isfile(path):
            raise FleetHipError(ERR_INVALID, f"{path} is older than its sources and rebuilding it failed: {exc}") from exc
    if not os.path.isfile(path):
        raise FleetHipError(ERR_NODEVICE, f"{path} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                                          f"(hipcc --offload-arch=gfx950).  There is no CPU fallback.{why}")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 and only finds the GPU if that copy is the one
    # that gets loaded; loading this library first would pull in /opt/rocm's copy instead ("No HIP GPUs are available"
    # on the first torch.cuda call afterwards).  Importing torch first makes the order deterministic.
    try:
        import torch  # noqa: F401
    except ImportError:  # pure C-ABI use without PyTorch: the system runtime is the only one
        pass
    lib = C.CDLL(path)
    vp, i32p, u8p, f32p, f64p = C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p
    lib.fleet_obs_dim.argtypes = [C.POINTER(FleetParams)]
    lib.fleet_obs_dim.restype = C.c_int
    lib.fleet_create.argtypes = [C.POINTER(FleetParams), C.POINTER(FleetTablesC), C.c_int, C.POINTER(vp)]
    lib.fleet_destroy.argtypes = [vp]
    lib.fleet_last_error.argtypes = [vp]
    lib.fleet_last_error.restype = C.c_char_p
    lib.fleet_set_stream.argtypes = [vp, vp]
    lib.fleet_get_stream.argtypes = [vp, C.POINTER(vp)]
    lib.fleet_use_own_stream.argtypes = [vp]
    lib.fleet_stream_query.argtypes = [vp]
    lib.fleet_stream_query.restype = C.c_int
    lib.fleet_log_dropped.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.fleet_log_capacity.argtypes = [vp]
    lib.fleet_log_read.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.fleet_log_clear.argtypes = [vp]
    lib.fleet_synchronize.argtypes = [vp]
    lib.fleet_set_start_schedule.argtypes = [vp, vp, C.c_int]
    lib.fleet_reset_dev.argtypes = [vp, u8p, f32p]
    lib.fleet_step_dev.argtypes = [vp, vp, C.c_int, f32p, f64p, u8p, f32p]
    lib.fleet_step_many_dev.argtypes = [vp, C.c_int, vp, C.c_int, f32p, f64p, vp]
    lib.fleet_rollout_policy_dev.argtypes = [vp, C.c_int, C.c_int, f32p, f64p, vp]
    lib.fleet_set_night_policy.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    if hasattr(lib, "fleet_set_rainflow_count_all"):  # (absent from the older libraries the A/B scripts run beside the tree's)
        lib.fleet_set_rainflow_count_all.argtypes = [vp, C.c_int]
        lib.fleet_set_rainflow_count_all.restype = C.c_int
    lib.fleet_reset_host.argtypes = [vp, u8p, f32p]
    lib.fleet_step_host.argtypes = [vp, vp, C.c_int, f32p, f64p, u8p, f32p]
    lib.fleet_get.argtypes = [vp, C.c_int, vp]
    lib.fleet_get_dev.argtypes = [vp, C.c_int, vp]
    lib.fleet_get_dist_factor.argtypes = [vp, vp]
    lib.fleet_check_errors.argtypes = [vp]
    lib.fleet_last_step_error_bits.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.fleet_timer_start.argtypes = [vp]
    lib.fleet_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.fleet_last_step_episodes.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_double)),
                                             C.POINTER(C.POINTER(C.c_int32))]
    lib.fleet_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.fleet_host_free.argtypes = [vp]
    lib.fleet_timer_mark.argtypes = [vp]
    lib.fleet_timer_read.argtypes = [vp, C.POINTER(C.c_float)]
    lib.fleet_run_tape_dev.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, f32p, f64p, u8p, C.c_int]
    lib.fleet_time_steps_dev.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, f32p, f64p, u8p, vp]
    lib.fleet_time_regions_begin.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, f32p, f64p, u8p, C.c_int]
    lib.fleet_time_regions_read.argtypes = [vp, vp]
    lib.fleet_rccl_unique_id.argtypes = [vp]
    lib.fleet_rccl_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    lib.fleet_rccl_comm_destroy.argtypes = [vp]
    lib.fleet_gather_episode_stats_rccl.argtypes = [vp, vp, C.c_int, vp]
    if hasattr(lib, "fleet_direct_queues"):  # (absent from older libraries the A/B scripts run beside the tree's)
        lib.fleet_direct_queues.argtypes = [vp]
        lib.fleet_direct_queues.restype = C.c_int
    if hasattr(lib, "fleet_direct_placement"):  # (ABI 9; absent from older libraries the A/B scripts run beside the tree's)
        lib.fleet_direct_placement.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.fleet_direct_split_plan.argtypes = [C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        lib.fleet_debug_direct_fault.argtypes = [vp, C.c_int, C.c_int]
        for name in ("fleet_direct_placement", "fleet_direct_split_plan", "fleet_debug_direct_fault"):
            getattr(lib, name).restype = C.c_int
    if hasattr(lib, "fleet_selftest_stress"):
        lib.fleet_selftest_stress.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_double)]
        lib.fleet_selftest_stress.restype = C.c_int
    if hasattr(lib, "fleet_selftest_division"):  # (absent from the round-4 library the A/B scripts run beside the tree's)
        lib.fleet_selftest_division.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.fleet_selftest_division.restype = C.c_int
    for name in ("fleet_create", "fleet_destroy", "fleet_set_stream", "fleet_get_stream", "fleet_use_own_stream", "fleet_log_dropped",
                 "fleet_log_capacity", "fleet_log_read",
                 "fleet_log_clear", "fleet_synchronize", "fleet_set_start_schedule",
                 "fleet_reset_dev", "fleet_step_dev", "fleet_step_many_dev", "fleet_rollout_policy_dev", "fleet_set_night_policy",
                 "fleet_reset_host", "fleet_step_host", "fleet_get", "fleet_get_dev", "fleet_get_dist_factor", "fleet_check_errors", "fleet_timer_start",
                 "fleet_timer_stop", "fleet_timer_mark", "fleet_timer_read", "fleet_run_tape_dev", "fleet_time_steps_dev",
                 "fleet_host_alloc", "fleet_host_free", "fleet_last_step_episodes", "fleet_last_step_error_bits",
                 "fleet_time_regions_begin", "fleet_time_regions_read", "fleet_rccl_unique_id", "fleet_rccl_comm_create",
                 "fleet_rccl_comm_destroy", "fleet_gather_episode_stats_rccl"):
        getattr(lib, name).restype = C.c_int
    _LIB = lib
    return lib


EXPORTED_SYMBOLS = (
    "fleet_obs_dim", "fleet_create", "fleet_destroy", "fleet_last_error", "fleet_set_stream", "fleet_get_stream", "fleet_use_own_stream",
    "fleet_synchronize", "fleet_stream_query", "fleet_log_capacity", "fleet_log_dropped", "fleet_log_read", "fleet_log_clear",
    "fleet_set_start_schedule", "fleet_reset_dev", "fleet_step_dev", "fleet_step_many_dev", "fleet_rollout_policy_dev",
    "fleet_set_night_policy", "fleet_reset_host",
    "fleet_step_host", "fleet_get", "fleet_get_dev", "fleet_get_dist_factor", "fleet_check_errors", "fleet_timer_start",
    "fleet_timer_stop", "fleet_timer_mark", "fleet_timer_read", "fleet_run_tape_dev", "fleet_time_steps_dev",
    "fleet_host_alloc", "fleet_host_free", "fleet_last_step_episodes", "fleet_last_step_error_bits",
    "fleet_time_regions_begin", "fleet_time_regions_read", "fleet_rccl_unique_id", "fleet_rccl_comm_create",
    "fleet_rccl_comm_destroy", "fleet_gather_episode_stats_rccl", "fleet_selftest_division", "fleet_direct_queues", "fleet_selftest_stress",
    "fleet_direct_placement", "fleet_direct_split_plan", "fleet_debug_direct_fault", "fleet_set_rainflow_count_all",
)
