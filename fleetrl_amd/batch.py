"""FleetBatch: thin, typed Python handle over the C ABI (include/fleet_hip.h, libfleet_hip.so).

Two calling styles, same kernels:
  * NumPy / host pointers (`reset`, `step`): synchronous, what a Gymnasium/SB3 caller sees;
  * device pointers (`reset_dev`, `step_dev`, `step_many_dev`, `run_tape_dev`): torch-ROCm tensors stay in HBM,
    launches are asynchronous on the handle's stream.
No CPU fallback: constructing a FleetBatch without the HIP library or without a GPU raises FleetHipError.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import FleetHipError

__all__ = ["FleetBatch", "FleetHipError"]


class _PinnedBuffer:
    """Owner of one `fleet_host_alloc` block.  The NumPy arrays handed out are views of a ctypes array that holds a
    reference to this object, so the block outlives the batch for as long as any view of it lives (an observation returned by
    step() stays valid for as long as the caller keeps it).
    When the last view dies the block is only PARKED (`_dead`): `hipHostFree` synchronises the device, so it must not run from a
    finaliser -- which may fire on any thread, in the middle of a timed region, or at interpreter shutdown after the HIP runtime
    is gone.  Parked blocks are released by `drain()` on the caller's thread: at the next step() / pinned_array() / close() of
    any batch."""

    _dead: list = []   # (lib, address) of blocks whose views are all gone

    def __init__(self, lib, nbytes: int):
        out = C.c_void_p()
        rc = lib.fleet_host_alloc(int(nbytes), C.byref(out))
        if rc != _capi.OK or not out.value:
            raise FleetHipError(rc, "fleet_host_alloc failed")
        self._lib, self.address, self.nbytes = lib, int(out.value), int(nbytes)

    def array(self, shape, dtype) -> np.ndarray:
        raw = (C.c_char * self.nbytes).from_address(self.address)
        raw._fleet_owner = self  # the view -> ctypes array -> owner chain keeps the block alive
        return np.frombuffer(raw, dtype=dtype).reshape(shape)

    def __del__(self):
        try:
            if self.address:
                _PinnedBuffer._dead.append((self._lib, self.address))
                self.address = 0
        except Exception:  # interpreter shutdown: the process is going away with its pinned memory
            pass

    @classmethod
    def drain(cls):
        """Free the parked blocks (called on the caller's thread, never from a finaliser)."""
        while True:  # several vec envs may be stepped from several threads: `pop` is atomic, emptiness is not a promise
            try:
                lib, addr = cls._dead.pop()
            except IndexError:
                break
            lib.fleet_host_free(C.c_void_p(addr))


class FleetBatch:
    def __init__(self, params: _capi.FleetParams, tables, time_feat: np.ndarray | None = None, device: int = 0):
        self.lib = _capi.load_library()
        self.params = params
        tc, keep = _capi.pack_tables(tables, time_feat)
        h = C.c_void_p()
        rc = self.lib.fleet_create(C.byref(params), C.byref(tc), int(device), C.byref(h))
        del keep  # fleet_create copied everything to the device
        if rc != _capi.OK:
            raise FleetHipError(rc, self.lib.fleet_last_error(None).decode())
        self.h = h
        self.device = int(device)
        self.E, self.N = int(params.num_envs), int(params.num_cars)
        self.obs_dim = int(self.lib.fleet_obs_dim(C.byref(params)))
        self._term = None
        self._obs_ring = []    # pinned observation buffers step() lands its transfer in, used in turn (own list: never
        self._obs_next = 0     # mixed with the caller's pinned_array() buffers)

    # ------------------------------------------------------------------------------------------------------
    def _check(self, rc: int, device_errors: bool = False):
        """`device_errors`: the call is one that reports the kernels' FLEET_DEVERR_* bits through FLEET_ERR_STATE (the host step,
        fleet_check_errors): map them to the reference's exception types.  Any other entry's ERR_STATE (a timer read before its
        start, a log that is off, ...) is an error of the call itself, whatever sticky device bits an env may carry."""
        if rc == _capi.ERR_STATE and device_errors:
            self._raise_device_error()
        if rc != _capi.OK:
            raise FleetHipError(rc, self.lib.fleet_last_error(self.h).decode())

    def _raise_device_error(self):
        """Device error bits -> the exception the reference raises at that line inside step() (fleet_environment.py:610,
        rainflow_sei_degradation.py:164-167,179-180,209-210); running off the table -- a KeyError of the reference's
        `db.loc[...]` lookups -- becomes an IndexError that names the env and the row.  Every exception carries `.status`
        (= ERR_STATE) and `.error_bits` and `.env`."""
        msg = self.lib.fleet_last_error(self.h).decode()
        bits_all = np.zeros(self.E, dtype=np.uint32)
        rows = np.zeros(self.E, dtype=np.int32)
        # (plain fleet_get calls: they cannot fail with ERR_STATE themselves)
        if self.lib.fleet_get(self.h, _capi.FIELDS["error_bits"][0], bits_all.ctypes.data) != _capi.OK or \
           self.lib.fleet_get(self.h, _capi.FIELDS["time_idx"][0], rows.ctypes.data) != _capi.OK or not bits_all.any():
            raise FleetHipError(_capi.ERR_STATE, msg)
        e = int(np.flatnonzero(bits_all)[0])
        bits = int(bits_all[e])
        where = f"env {e} (global env {e + int(self.params.env_id_offset)}), table row {int(rows[e])} of {int(self.params.table_rows)}"
        if bits & _capi.DEVERR_PLACEMENT:
            exc = FleetHipError(_capi.ERR_STATE, "a launch on the library's own queue ran a workgroup on another die than the queue's placement "
                                                 f"probe says (FLEET_DEVERR_PLACEMENT): the run's results are void: {where}")
        elif bits & _capi.DEVERR_INTERNAL:
            exc = FleetHipError(_capi.ERR_STATE, f"internal error of the library (inconsistent launch arguments): {where}")
        elif bits & _capi.DEVERR_OBS_FORMAT:
            exc = TypeError("Observation format not recognized")
        elif bits & _capi.DEVERR_DOD_RANGE:
            exc = TypeError("DoD too large.")
        elif bits & _capi.DEVERR_NEG_LIFE:
            exc = TypeError("Life degradation is negative")
        elif bits & _capi.DEVERR_SOH_MISMATCH:
            exc = RuntimeError("Degradation calculation is not correct")
        else:
            exc = IndexError(f"the episode runs past the last table row: {where}")
        exc.status, exc.error_bits, exc.env, exc.detail = _capi.ERR_STATE, bits, e, f"{where}: {msg}"
        raise exc

    def close(self):
        if getattr(self, "h", None):
            self.lib.fleet_destroy(self.h)
            self.h = None
        self._obs_ring = []  # the blocks themselves live as long as any view of them (see _PinnedBuffer)
        _PinnedBuffer.drain()

    def pinned_array(self, shape, dtype=np.float32) -> np.ndarray:
        """A NumPy array in pinned (page-locked) host memory (`fleet_host_alloc`): the host entry points move such buffers
        over PCIe without staging.  The memory is freed when the last view of the array is gone (independent of this batch's
        lifetime)."""
        _PinnedBuffer.drain()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return _PinnedBuffer(self.lib, n).array(shape, dtype)

    def __del__(self):
        try:
            if getattr(self, "h", None):   # (a finaliser: destroy the handle, leave parked pinned blocks to the next drain())
                self.lib.fleet_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def synchronize(self):
        self._check(self.lib.fleet_synchronize(self.h))

    def spin_wait(self):
        """Wait for the handle's stream by polling (hipStreamQuery): returns within microseconds of the last launch's end,
        where a blocking synchronize pays the host's wake-up latency (bench.py's short timed regions)."""
        while True:
            rc = self.lib.fleet_stream_query(self.h)
            if rc == 0:
                return
            if rc != -1:
                self._check(rc)

    def set_stream(self, hip_stream: int):
        self._check(self.lib.fleet_set_stream(self.h, C.c_void_p(hip_stream)))

    def use_own_stream(self):
        """Back to the handle's own (non-blocking) stream after `set_stream` / `use_torch_stream`."""
        self._check(self.lib.fleet_use_own_stream(self.h))

    def stream_ptr(self) -> int:
        """The hipStream_t the handle launches on (an integer address)."""
        out = C.c_void_p()
        self._check(self.lib.fleet_get_stream(self.h, C.byref(out)))
        return int(out.value or 0)

    def use_torch_stream(self, device=None):
        """Launch on torch's current stream of the handle's device from now on: launches are then ordered with the torch ops
        that produce their inputs and consume their outputs (the handle's own stream is non-blocking, i.e. NOT ordered with
        torch's).  Call again after switching torch streams."""
        import torch

        dev = torch.device("cuda", self.device) if device is None else device
        self.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    # ---- device-side data log (FleetParams.log_data) -----------------------------------------------------------
    def log_capacity(self) -> int:
        return int(self.lib.fleet_log_capacity(self.h))

    def log_read(self, with_obs: bool = True):
        """-> dict(pos i32[E], row i32[cap,E], env f64[cap,E,4], ev f64[cap,E,4,N], obs f32[cap,E,obs_dim] | None): the whole
        ring in one transfer per array (layout: include/fleet_hip.h, fleet_log_read)."""
        cap = self.log_capacity()
        if cap <= 0:
            raise FleetHipError(_capi.ERR_INVALID, "the data log is off (log_data = 0)")
        pos = np.zeros(self.E, dtype=np.int32)
        row = np.zeros((cap, self.E), dtype=np.int32)
        env = np.zeros((cap, self.E, 4))
        ev = np.zeros((cap, self.E, 4, self.N))
        obs = np.zeros((cap, self.E, self.obs_dim), dtype=np.float32) if with_obs else None
        self._check(self.lib.fleet_log_read(self.h, pos.ctypes.data, row.ctypes.data, env.ctypes.data, ev.ctypes.data,
                                             None if obs is None else obs.ctypes.data))
        return {"pos": pos, "row": row, "env": env, "ev": ev, "obs": obs, "capacity": cap}

    def log_dropped(self) -> int:
        """Rows the ring has already overwritten, summed over the envs (0 = get_log() is complete)."""
        n = C.c_int64()
        self._check(self.lib.fleet_log_dropped(self.h, C.byref(n)))
        return int(n.value)

    def log_clear(self):
        self._check(self.lib.fleet_log_clear(self.h))

    def set_start_schedule(self, starts):
        if starts is None:
            self._check(self.lib.fleet_set_start_schedule(self.h, None, 0))
            return
        s = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1, self.E)
        self._check(self.lib.fleet_set_start_schedule(self.h, s.ctypes.data, s.shape[0]))

    # ---- host (NumPy) path ---------------------------------------------------------------------------------
    def reset(self, mask=None, out: np.ndarray | None = None) -> np.ndarray:
        obs = out if out is not None else np.zeros((self.E, self.obs_dim), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self.lib.fleet_reset_host(self.h, None if m is None else m.ctypes.data, obs.ctypes.data))
        return obs

    @staticmethod
    def _act(actions, shape):
        a = np.asarray(actions)
        if a.dtype == np.float64:
            a = np.ascontiguousarray(a)
            dt = _capi.ACT_F64
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            dt = _capi.ACT_F32
        if a.shape != shape:
            raise ValueError(f"actions must have shape {shape}, got {a.shape}")
        return a, dt

    OBS_RING = 4  # pinned observation buffers step() cycles through

    def step(self, actions, copy: bool = True):
        """-> (obs f32[E,obs_dim], reward f64[E], done u8[E], terminal_obs f32[E,obs_dim]); `terminal_obs` is a buffer
        reused between calls whose rows are valid only where `done` is set.
        `copy=True` (default) returns a fresh array, like the reference's env does every step (the transfer is pipelined with the
        host copy inside the library).  `copy=False` lands the transfer in one of OBS_RING pinned buffers used in turn and returns
        that buffer itself (no 4 * E * obs_dim byte copy on the host): it is overwritten OBS_RING calls later -- enough for an SB3 loop, which
        holds the previous observation while it steps and copies what it keeps; not for code that collects observations in a
        list.  Either way the memory stays valid for as long as the returned array is referenced -- with `copy=False` that is
        page-locked memory (E * obs_dim * 4 bytes per retained array), which stays pinned until the array is dropped."""
        a, dt = self._act(actions, (self.E, self.N))
        if _PinnedBuffer._dead:
            _PinnedBuffer.drain()
        if copy:
            # a fresh (pageable) array per step: the library lands the transfer in a pinned buffer of the handle piece by piece
            # and copies each piece here while the next ones are still on the link (fleet_step_host)
            obs = np.empty((self.E, self.obs_dim), dtype=np.float32)
        else:
            if len(self._obs_ring) < self.OBS_RING:
                self._obs_ring.append(self.pinned_array((self.E, self.obs_dim)))
            obs = self._obs_ring[self._obs_next % len(self._obs_ring)]
            self._obs_next += 1
        if obs.shape != (self.E, self.obs_dim) or obs.dtype != np.float32:  # what fleet_step_host will write
            raise FleetHipError(_capi.ERR_INVALID, "internal: observation buffer of the wrong shape")
        if self._term is None:  # reused across steps: only the rows of envs that just finished are meaningful
            self._term = np.zeros((self.E, self.obs_dim), dtype=np.float32)
        term = self._term
        rew = np.empty(self.E)
        done = np.empty(self.E, dtype=np.uint8)
        self._check(self.lib.fleet_step_host(self.h, a.ctypes.data, dt, obs.ctypes.data, rew.ctypes.data,
                                              done.ctypes.data, term.ctypes.data), device_errors=True)
        return obs, rew, done, term

    def last_step_episodes(self):
        """(env indices, returns, lengths) of the episodes that ended in the last `step()` -- already on the host, no launch."""
        n = C.c_int32()
        idx, ret, ln = C.POINTER(C.c_int32)(), C.POINTER(C.c_double)(), C.POINTER(C.c_int32)()
        self._check(self.lib.fleet_last_step_episodes(self.h, C.byref(n), C.byref(idx), C.byref(ret), C.byref(ln)))
        k = int(n.value)
        if k == 0:
            return np.zeros(0, np.int32), np.zeros(0), np.zeros(0, np.int32)
        return (np.ctypeslib.as_array(idx, (k,)).copy(), np.ctypeslib.as_array(ret, (k,)).copy(), np.ctypeslib.as_array(ln, (k,)).copy())

    # ---- device-pointer path (integers are raw device addresses, e.g. torch.Tensor.data_ptr()) -------------------
    def reset_dev(self, obs_ptr: int, mask_ptr: int | None = None):
        self._check(self.lib.fleet_reset_dev(self.h, mask_ptr, obs_ptr))

    def step_dev(self, actions_ptr: int, obs_ptr: int, reward_ptr: int, done_ptr: int, terminal_ptr: int | None = None,
                 act_dtype: int = _capi.ACT_F32):
        self._check(self.lib.fleet_step_dev(self.h, actions_ptr, act_dtype, obs_ptr, reward_ptr, done_ptr, terminal_ptr))

    def step_many_dev(self, K: int, actions_ptr: int, obs_ptr: int, reward_sum_ptr: int, done_count_ptr: int | None = None,
                      act_dtype: int = _capi.ACT_F32):
        self._check(self.lib.fleet_step_many_dev(self.h, int(K), actions_ptr, act_dtype, obs_ptr, reward_sum_ptr, done_count_ptr))

    def rollout_policy_dev(self, policy: int, K: int, obs_ptr: int, reward_sum_ptr: int, done_count_ptr: int | None = None):
        """K steps in one launch with a built-in policy (_capi.POLICY_UNCONTROLLED / POLICY_DISTRIBUTED / POLICY_NIGHT)."""
        self._check(self.lib.fleet_rollout_policy_dev(self.h, int(policy), int(K), obs_ptr, reward_sum_ptr, done_count_ptr))

    def set_night_policy(self, charging_hour: int, charging_minute: int, max_hours: int):
        """Parameters of _capi.POLICY_NIGHT (see fleetrl_amd.policies.night_schedule); clears the per-env window state."""
        self._check(self.lib.fleet_set_night_policy(self.h, int(charging_hour), int(charging_minute), int(max_hours)))

    def set_rainflow_count_all(self, on: bool = True):
        """Keep the rainflow count running to the end of every episode (from each env's next reset on) instead of stopping it at the
        episode's last degradation row, after which nobody reads it (include/fleet_hip.h).  Diagnostics / count tests only: state of
        health, observations and rewards do not depend on it."""
        self._check(self.lib.fleet_set_rainflow_count_all(self.h, 1 if on else 0))

    def run_tape_dev(self, steps: int, tape_ptr: int, tape_len: int, obs_ptr: int, reward_ptr: int, done_ptr: int,
                     use_graph: bool = True, act_dtype: int = _capi.ACT_F32):
        self._check(self.lib.fleet_run_tape_dev(self.h, int(steps), tape_ptr, int(tape_len), act_dtype, obs_ptr,
                                                 reward_ptr, done_ptr, int(use_graph)))

    def direct_placement(self):
        """(map8, num_xcc, any_grid): the workgroup -> die map the batch's own queue was probed to have (fleet_direct_placement)."""
        m = (C.c_int32 * 8)()
        nx, ag = C.c_int32(), C.c_int32()
        self._check(self.lib.fleet_direct_placement(self.h, m, C.byref(nx), C.byref(ag)))
        return [int(x) for x in m], int(nx.value), bool(ag.value)

    def debug_direct_fault(self, kind: int, tape_row: int):
        """TEST HOOK: corrupt the prepared argument block of one tape row (fleet_debug_direct_fault)."""
        self._check(self.lib.fleet_debug_direct_fault(self.h, int(kind), int(tape_row)))

    def direct_queues(self) -> int:
        """How the handle's last direct run (use_graph=_capi.LAUNCH_DIRECT) was laid out: 0 none yet, 1 one queue, 2 two queues."""
        return int(self.lib.fleet_direct_queues(self.h)) if hasattr(self.lib, "fleet_direct_queues") else 0

    def time_steps_dev(self, steps: int, tape_ptr: int, tape_len: int, obs_ptr: int, reward_ptr: int, done_ptr: int,
                       act_dtype: int = _capi.ACT_F32) -> np.ndarray:
        """Per-launch device durations [ms] measured with one HIP event pair per launch on the handle's stream."""
        ms = np.zeros(int(steps), dtype=np.float32)
        self._check(self.lib.fleet_time_steps_dev(self.h, int(steps), tape_ptr, int(tape_len), act_dtype, obs_ptr,
                                                   reward_ptr, done_ptr, ms.ctypes.data))
        return ms

    def time_regions_begin(self, regions: int, steps: int, tape_ptr: int, tape_len: int, obs_ptr: int, reward_ptr: int, done_ptr: int,
                           use_graph: bool = True, act_dtype: int = _capi.ACT_F32):
        """Enqueue `regions` event-bracketed regions of exactly `steps` launches each (asynchronous)."""
        self._n_regions = int(regions)
        self._check(self.lib.fleet_time_regions_begin(self.h, int(regions), int(steps), tape_ptr, int(tape_len), act_dtype, obs_ptr,
                                                       reward_ptr, done_ptr, int(use_graph)))

    def time_regions_read(self) -> np.ndarray:
        """Per-region device durations [ms] of the regions enqueued by time_regions_begin."""
        ms = np.zeros(self._n_regions, dtype=np.float32)
        self._check(self.lib.fleet_time_regions_read(self.h, ms.ctypes.data))
        return ms

    def timer_start(self):
        self._check(self.lib.fleet_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self.lib.fleet_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    def timer_mark(self):
        self._check(self.lib.fleet_timer_mark(self.h))

    def timer_read(self) -> float:
        ms = C.c_float()
        self._check(self.lib.fleet_timer_read(self.h, C.byref(ms)))
        return float(ms.value)

    # ---- state access ----------------------------------------------------------------------------------------
    def get(self, name: str) -> np.ndarray:
        fid, dtype, per_car = _capi.FIELDS[name]
        out = np.zeros((self.E, self.N) if per_car else (self.E,), dtype=dtype)
        self._check(self.lib.fleet_get(self.h, fid, out.ctypes.data))
        return out

    def get_dev(self, name: str, out_ptr: int):
        """Unpack a state field into a device buffer (asynchronous on the handle's stream); dtype/shape as `get`."""
        self._check(self.lib.fleet_get_dev(self.h, _capi.FIELDS[name][0], out_ptr))

    def dist_factor(self) -> np.ndarray:
        out = np.zeros((self.E, self.N))
        self._check(self.lib.fleet_get_dist_factor(self.h, out.ctypes.data))
        return out

    def check_errors(self):
        """Raise if any env carries device error bits (see _raise_device_error); `step()` does this by itself."""
        self._check(self.lib.fleet_check_errors(self.h), device_errors=True)

    def last_step_error_bits(self) -> int:
        """OR of all envs' device error bits as of the last `step()` (already on the host)."""
        out = C.c_uint32()
        rc = self.lib.fleet_last_step_error_bits(self.h, C.byref(out))
        if rc != _capi.OK:
            raise FleetHipError(rc, self.lib.fleet_last_error(self.h).decode())
        return int(out.value)
