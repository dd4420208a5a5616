"""ctypes front end of the CPU oracle (oracle/fleet_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported from tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg -- never
from the product package.  It re-uses the product's plain struct definitions (fleetrl_amd/_capi.py);
the dependency goes oracle -> product data layouts, never the other way round.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from fleetrl_amd import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfleet_oracle.so")
_LIB = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fleet_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "fleet_hip.h")
    stale = (not os.path.isfile(_SO)) or any(os.path.getmtime(f) > os.path.getmtime(_SO) for f in (src, hdr))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "all"], check=True, capture_output=True)
    return _SO


def load():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(build())
        vp = C.c_void_p
        lib.oracle_create.restype = vp
        lib.oracle_create.argtypes = [C.POINTER(_capi.FleetParams), C.POINTER(_capi.FleetTablesC)]
        lib.oracle_destroy.argtypes = [vp]
        lib.oracle_destroy.restype = None
        lib.oracle_set_threads.argtypes = [vp, C.c_int]
        lib.oracle_set_threads.restype = None
        lib.oracle_obs_dim.argtypes = [C.POINTER(_capi.FleetParams)]
        lib.oracle_set_start_schedule.argtypes = [vp, vp, C.c_int]
        lib.oracle_reset.argtypes = [vp, vp, vp]
        lib.oracle_step.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
        lib.oracle_get.argtypes = [vp, C.c_int, vp]
        lib.oracle_get_dist_factor.argtypes = [vp, vp]
        lib.oracle_rainflow.argtypes = [vp, C.c_int, vp]
        lib.oracle_soc_violation_penalty.argtypes = [C.c_double]
        lib.oracle_soc_violation_penalty.restype = C.c_double
        lib.oracle_overloading_penalty.argtypes = [C.c_double, C.c_double]
        lib.oracle_overloading_penalty.restype = C.c_double
        lib.oracle_philox_start.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        lib.oracle_philox_start.restype = C.c_uint32
        _LIB = lib
    return _LIB


def rainflow(series) -> np.ndarray:
    """Cycles of `series` as rows (range, mean, count, i_end) in emission order."""
    s = np.ascontiguousarray(series, dtype=np.float64)
    out = np.zeros((max(len(s), 1), 4))
    n = load().oracle_rainflow(s.ctypes.data, len(s), out.ctypes.data)
    return out[:n]


class OracleBatch:
    """Batch of reference-exact CPU envs sharing one table set (mirrors the product's FleetBatch)."""

    def __init__(self, params: _capi.FleetParams, tables, time_feat: np.ndarray | None = None, threads: int = 1):
        self.lib = load()
        self.params = params
        self._tc, self._keep = _capi.pack_tables(tables, time_feat)
        self.h = self.lib.oracle_create(C.byref(params), C.byref(self._tc))
        if not self.h:
            raise RuntimeError("oracle_create failed (ABI mismatch?)")
        self.E, self.N = params.num_envs, params.num_cars
        self.obs_dim = self.lib.oracle_obs_dim(C.byref(params))
        self.lib.oracle_set_threads(self.h, threads)

    def close(self):
        if self.h:
            self.lib.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_start_schedule(self, starts):
        if starts is None:
            self.lib.oracle_set_start_schedule(self.h, None, 0)
            return
        s = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1, self.E)
        self.lib.oracle_set_start_schedule(self.h, s.ctypes.data, s.shape[0])

    def reset(self, mask=None) -> np.ndarray:
        obs = np.zeros((self.E, self.obs_dim), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.lib.oracle_reset(self.h, None if m is None else m.ctypes.data, obs.ctypes.data)
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions)
        if a.dtype == np.float64:
            dt = _capi.ACT_F64
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            dt = _capi.ACT_F32
        assert a.shape == (self.E, self.N)
        obs = np.zeros((self.E, self.obs_dim), dtype=np.float32)
        term = np.zeros((self.E, self.obs_dim), dtype=np.float32)
        rew = np.zeros(self.E)
        done = np.zeros(self.E, dtype=np.uint8)
        self.lib.oracle_step(self.h, a.ctypes.data, dt, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, term.ctypes.data)
        return obs, rew, done, term

    def get(self, name: str) -> np.ndarray:
        fid, dtype, per_car = _capi.FIELDS[name]
        out = np.zeros((self.E, self.N) if per_car else (self.E,), dtype=dtype)
        if self.lib.oracle_get(self.h, fid, out.ctypes.data):
            raise KeyError(name)
        return out

    def dist_factor(self) -> np.ndarray:
        out = np.zeros((self.E, self.N))
        self.lib.oracle_get_dist_factor(self.h, out.ctypes.data)
        return out


class NightChargingRule:
    """The action rule of the reference's night-charging benchmark loop, one instance per env
    (benchmarking/night_charging.py:81-98): the `charging` / `charging_start` loop variables live here and, as in the
    reference, are NOT cleared when the episode resets.  `parameters()` restates :50-73 with pandas objects replaced by
    the (hour, minute) of the rows where a vehicle leaves home."""

    def __init__(self, charging_hour: int, charging_minute: int, max_time_needed: float, minutes_per_step: int, is_ct: bool):
        self.charging_hour, self.charging_minute = charging_hour, charging_minute
        self.max_time_needed, self.minutes, self.is_ct = max_time_needed, minutes_per_step, is_ct
        self.charging = False
        self.charging_start = None

    @staticmethod
    def parameters(leave_hours, leave_minutes, target_soc, cap, eff, evse):
        import math
        earliest = min(zip(leave_hours, leave_minutes))  # df_leaving_home['date'].dt.time.min() :54
        earliest_dep = earliest[0] + earliest[1] / 60  # :56
        max_time_needed = target_soc * cap / eff / evse  # :63
        starting_time = 24 + (earliest_dep - max_time_needed)  # :64-65
        if starting_time > 24:
            starting_time = 23.99  # :66-67
        charging_hour = int(math.modf(starting_time)[1])  # :69
        minutes = np.asarray([0, 15, 30, 45])
        closest_index = np.abs(minutes - int(math.modf(starting_time)[0] * 60)).argmin()  # :72
        return charging_hour, int(minutes[closest_index]), max_time_needed

    def action(self, row: int, hour: int, minute: int, n_evs: int, dist_factor: np.ndarray) -> np.ndarray:
        """Action for the env whose current table row is `row` (clock `hour`:`minute`); updates the loop state."""
        if 11 <= hour <= 14 and self.is_ct:  # :85-88 (`continue`: no bookkeeping on these rows)
            return np.clip(np.ones(n_evs) * dist_factor, 0, 1)
        if (self.charging_hour <= hour and self.charging_minute <= minute) or self.charging:  # :90
            if not self.charging:
                self.charging_start = row  # :91-92
            self.charging = True
            a = np.ones(n_evs)  # :94
        else:
            a = np.zeros(n_evs)  # :96
        # :97-98 -- timestamps of a regular grid: (time - charging_start) = (row - start_row) * minutes
        if self.charging and ((row - self.charging_start) * self.minutes * 60) / 3600 > int(self.max_time_needed):
            self.charging = False
        return a
