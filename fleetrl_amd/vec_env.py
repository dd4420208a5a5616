"""The drop-in boundary: `FleetEnv` / `FleetVecEnv` / `FleetVectorEnv` with the reference's Python surface.

Reference: `fleetrl.fleet_env.fleet_environment.FleetEnv(env_config: str | dict)`
(/root/reference/fleetrl/fleet_env/fleet_environment.py:50-799), a `gymnasium.Env` that SB3 wraps as
`make_vec_env(FleetEnv, n_envs, vec_env_cls=SubprocVecEnv, env_kwargs={"env_config": cfg})`
(/root/reference/complete_pipeline.ipynb cell 13) -- one OS process per env.  Here `num_envs` is the batch dimension of
ONE process and every step is one fused HIP kernel launch (libfleet_hip.so through include/fleet_hip.h).

  FleetVecEnv    stable-baselines3 `VecEnv` duck type: reset() -> obs ; step(actions) -> (obs, rewards, dones, infos),
                 step_async/step_wait, env_method, get_attr/set_attr, seed, close, env_is_wrapped; auto-reset with
                 infos[i]["terminal_observation"] (+ "episode": {"r","l"} as SB3's Monitor would add).
  FleetVectorEnv gymnasium vector signature: reset(seed=, options=) -> (obs, infos) ;
                 step(actions) -> (obs, rewards, terminated, truncated, infos); auto-reset, "final_observation".
  FleetEnv       the reference's single-env class (num_envs = 1, no auto-reset): reset(**kwargs) -> (obs, info);
                 step(actions) -> (obs, float, bool, False, {}); is_done/get_time/get_start_time/set_start_time/
                 get_dist_factor/get_log.

The same config dict / JSON path as the reference is accepted verbatim (fleetrl_amd/config.py).  Tables come from
the CSVs named in the config (fleetrl_amd/prestage.py) or are passed in ready-made (`tables=`, e.g. from
fleetrl_amd/synth.py).  There is no CPU fallback: without the HIP library and a GPU construction raises.
"""
from __future__ import annotations

import numpy as np

from . import _capi
from .batch import FleetBatch
from .config import ResolvedConfig, resolve_config
from .params import make_params, obs_dim, time_features
from .prestage import FleetTables, build_tables_from_config
from .spaces import Box, observation_bounds

__all__ = ["FleetCore", "FleetVecEnv", "FleetVectorEnv", "FleetEnv"]

# The reference's class IS a `gymnasium.Env` (fleet_environment.py:50) and is multiplied by an SB3 `VecEnv` (requirements.txt:
# gymnasium==0.29.1, stable-baselines3==2.3.2).  SB3 and gymnasium test with isinstance (`BaseAlgorithm._wrap_env`,
# `check_env`, `Monitor`), so when those packages are installed the classes below derive from their base classes; without them
# (this build image) they are plain classes with the same surface.
try:
    import gymnasium as _gym

    _GymEnv = _gym.Env
    _GymVectorEnv = getattr(getattr(_gym, "vector", None), "VectorEnv", object)
except ImportError:  # pragma: no cover - depends on the installation
    _GymEnv = object
    _GymVectorEnv = object
try:
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
except ImportError:  # pragma: no cover
    _SB3VecEnv = object


class FleetCore:
    """Shared engine: config -> tables -> params -> FleetBatch, plus the reference's getters."""

    def __init__(self, env_config, num_envs: int = 1, *, tables: FleetTables | None = None, schedule=None,
                 device: int = 0, auto_reset: bool = True, seed: int | None = None, env_id_offset: int = 0,
                 start_rows=None, extrema: dict | None = None, start_range: tuple[int, int] | None = None):
        """`tables`: ready-made FleetTables (else built from the CSVs the config names); `start_rows`: injected episode
        start rows [n_episodes, num_envs]; `extrema` / `start_range`: overrides for tables that are a cut window of a
        longer table (see fleetrl_amd.params.make_params)."""
        self.rc: ResolvedConfig = resolve_config(env_config)
        self.env_config = self.rc.raw
        if tables is None:
            tables = build_tables_from_config(self.rc.raw, schedule=schedule)
        self.tables = tables
        self.num_envs = int(num_envs)
        self.num_cars = tables.N
        self.params = make_params(self.rc, tables, self.num_envs, auto_reset=auto_reset, env_id_offset=env_id_offset, seed=seed,
                                  extrema=extrema, start_range=start_range)
        self.batch = FleetBatch(self.params, tables, time_features(tables), device=device)
        self.obs_dim = self.batch.obs_dim
        assert self.obs_dim == obs_dim(self.rc, self.num_cars)
        low, high = observation_bounds(self.obs_dim, self.rc.normalize_in_env)
        self.single_observation_space = Box(low=low, high=high, dtype=np.float32)          # fleet_environment.py:316-319
        self.single_action_space = Box(low=-1, high=1, shape=(self.num_cars,), dtype=np.float32)  # :322-325
        self._start_time_override = [None] * self.num_envs
        if start_rows is not None:
            self.batch.set_start_schedule(start_rows)
        # DataLogger (utils/data_logger/data_logger.py): the rows live in a device-side ring written by the kernels
        # (include/fleet_hip.h "device-side data log"); get_log() rebuilds the reference's DataFrame from it
        self.log_data = bool(self.rc.raw.get("log_data", False))
        self._log_episode_len = self.rc.episode_length * (60 // self.rc.minutes)

    # -- data log (fleet_environment.py:420-432, 659-690) ---------------------------------------------------------
    def get_log(self):
        """`FleetEnv.get_log` (:741-748): the DataLogger DataFrame of every env (same columns as the reference), or None
        when `log_data` is off.  Works after single steps, K-step launches and device-side policy rollouts alike: the rows
        were written by the kernels; one device-to-host copy per array here.  `Penalties` is the reference's
        `reward - cashflow * price_multiplier` (quirk Q13); "Charging energy" replicates the carry-over of
        `charging_energy + discharging_energy` across cars (quirk Q8, ev_charger.py:81-82,114,174,212); `Episode` is
        `row_number // episode_length + 1` (data_logger.py:50).  The ring keeps the last `log_capacity` rows per env."""
        if not self.log_data:
            return [None] * self.num_envs
        import pandas as pd

        lg = self.batch.log_read()
        cap, pm, N = lg["capacity"], self.rc.price_multiplier, self.num_cars
        hour, minute = np.asarray(self.tables.hour), np.asarray(self.tables.minute)
        dropped = int(np.maximum(lg["pos"].astype(np.int64) - cap, 0).sum())
        if dropped:
            import warnings

            warnings.warn(f"get_log: the device-side log ring (log_capacity = {cap} rows per env) has overwritten {dropped} rows; "
                          "the reference's DataLogger is unbounded -- pass a larger `log_capacity` to keep every row", RuntimeWarning)
        lane = np.arange(N)
        frames = []
        for i in range(self.num_envs):  # one vectorised pass over the env's rows; the DataFrame itself is per env
            pos = int(lg["pos"][i])
            ks = np.arange(max(0, pos - cap), pos)
            sl = ks % cap
            raw = lg["row"][sl, i].astype(np.int64)
            is_reset = raw < 0
            t = raw & 0x7FFFFFFF
            envv = lg["env"][sl, i]                      # [rows, 4]: reward, cashflow, overload, SOC missing
            act, energy, deg, soh = (lg["ev"][sl, i, j] for j in range(4))   # [rows, N] each
            obs = lg["obs"][sl, i]
            # quirk Q8: charging_energy / discharging_energy are not cleared from car to car, so each car logs the energy of the
            # last car (itself included) with a charge action plus that of the last car with a discharge action
            last_c = np.maximum.accumulate(np.where(act >= 0, lane, -1), axis=1)
            last_d = np.maximum.accumulate(np.where(act < 0, lane, -1), axis=1)
            ce = np.where(last_c >= 0, np.take_along_axis(energy, np.maximum(last_c, 0), axis=1), 0.0)
            de = np.where(last_d >= 0, np.take_along_axis(energy, np.maximum(last_d, 0), axis=1), 0.0)
            charge_log = np.where(is_reset[:, None], 0.0, ce + de)
            deg_row = (~is_reset) & bool(self.rc.calc_deg) & (hour[t] == 14) & (minute[t] == 45)
            rew, cash = envv[:, 0], envv[:, 1]
            penalty = np.where(is_reset, 0.0, rew - cash * pm)
            degradation = [deg[r].copy() if deg_row[r] else 0.0 for r in range(len(ks))]
            frames.append(pd.DataFrame({
                "Episode": ks // self._log_episode_len + 1, "Time": [self._stamp(x) for x in t], "Observation": list(obs.copy()),
                "Action": list(act.copy()), "Reward": rew, "Cashflow": cash, "Penalties": penalty,
                "Grid overloading": np.abs(envv[:, 2]), "SOC violation": np.abs(envv[:, 3]), "Degradation": degradation,
                "Charging energy": list(charge_log), "SOH": list(soh.copy())}))
        return frames

    def clear_log(self):
        if self.log_data:
            self.batch.log_clear()

    # -- reference getters (fleet_environment.py:741-799), per env ----------------------------------------
    def _stamp(self, row: int):
        d = self.tables.dates[int(row)]
        try:
            import pandas as pd

            return pd.Timestamp(d)
        except Exception:  # pragma: no cover
            return d

    def is_done(self):
        return [bool(x) for x in self.batch.get("done")]

    def get_time(self):
        return [self._stamp(t) for t in self.batch.get("time_idx")]

    def get_start_time(self):
        rows = self.batch.get("start_idx")
        return [o if o is not None else self._stamp(r) for o, r in zip(self._start_time_override, rows)]

    def set_start_time(self, start_time, indices=None):
        """The reference only overwrites the attribute (`self.episode.start_time = start_time`, :772); the next
        reset() picks its own start again.  Replicated: affects get_start_time() until the next reset."""
        for i in (range(self.num_envs) if indices is None else indices):
            self._start_time_override[i] = start_time

    def get_dist_factor(self):
        return list(self.batch.dist_factor())

    def clear_start_overrides(self, mask=None):
        for i in range(self.num_envs):
            if mask is None or mask[i]:
                self._start_time_override[i] = None

    def close(self):
        self.batch.close()


class FleetVecEnv(_SB3VecEnv):
    """stable-baselines3 `VecEnv` over one fused GPU batch: a subclass of `stable_baselines3.common.vec_env.VecEnv` when SB3 is
    installed (every abstract method of SB3 2.3.2 is implemented: reset, step_async, step_wait, close, get_attr, set_attr,
    env_method, env_is_wrapped), the same duck type without it."""

    def __init__(self, env_config, num_envs: int, **kw):
        # copy_obs=True: step() returns a fresh observation array like SubprocVecEnv does; False: the pinned transfer buffer
        # itself (overwritten four steps later; saves a 4 * num_envs * obs_dim byte host copy per step, see FleetBatch.step)
        self.copy_obs = bool(kw.pop("copy_obs", True))
        self.core = FleetCore(env_config, num_envs, auto_reset=True, **kw)
        self._actions = None
        self._any_override = False
        if _SB3VecEnv is not object:  # sets num_envs / spaces / reset_infos / _seeds / _options and asks get_attr("render_mode")
            _SB3VecEnv.__init__(self, self.core.num_envs, self.core.single_observation_space, self.core.single_action_space)
        else:
            self.num_envs = self.core.num_envs
            self.observation_space = self.core.single_observation_space
            self.action_space = self.core.single_action_space
            self.render_mode = None
            self.metadata = {"render_modes": []}
            self.reset_infos = [{} for _ in range(self.num_envs)]

    def reset(self):
        self.core.clear_start_overrides()
        self._any_override = False
        return self.core.batch.reset()

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self):
        acts = np.asarray(self._actions).reshape(self.num_envs, -1)
        obs, rew, done, term = self.core.batch.step(acts, copy=self.copy_obs)
        dones = done.astype(bool)
        # the reference's info is always {} (:235): the envs that did not finish share ONE empty dict per step (4096 dict
        # allocations per step cost more than the step); an env that finished gets a dict of its own
        empty: dict = {}
        infos = [empty] * self.num_envs
        if dones.any():
            idx, ret, ln = self.core.batch.last_step_episodes()  # came over with the step's small outputs: no extra launch
            for k, i in enumerate(idx):
                infos[i] = {"terminal_observation": term[i].copy(), "TimeLimit.truncated": False,
                            "episode": {"r": float(ret[k]), "l": int(ln[k])}}
            if self._any_override:
                self.core.clear_start_overrides(dones)
        return obs, rew.astype(np.float32), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def env_method(self, method_name: str, *method_args, indices=None, **method_kwargs):
        """`VecEnv.env_method("is_done")[0]` etc. -- the reference's documented way to reach its getters (:741-799)."""
        idx = self._indices(indices)
        if method_name == "set_start_time":
            self._any_override = True
            self.core.set_start_time(*method_args, indices=idx, **method_kwargs)
            return [None] * len(idx)
        fn = getattr(self.core, method_name, None)
        if fn is None or method_name.startswith("_"):
            raise AttributeError(f"FleetEnv has no method {method_name!r}")
        out = fn(*method_args, **method_kwargs)
        return [out[i] for i in idx]

    def get_attr(self, attr_name: str, indices=None):
        idx = self._indices(indices)
        per_env = {"num_cars": self.core.num_cars, "env_config": self.core.env_config, "render_mode": None,
                   "observation_space": self.core.single_observation_space, "action_space": self.core.single_action_space}
        if attr_name not in per_env:
            raise AttributeError(attr_name)
        return [per_env[attr_name] for _ in idx]

    def set_attr(self, attr_name: str, value, indices=None):
        raise AttributeError(f"attribute {attr_name!r} cannot be set on the fused batch")

    def seed(self, seed=None):
        return [None] * self.num_envs  # the reference ignores reset(seed=...) (quirk Q12); the start sampler is seeded at construction

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None

    def close(self):
        self.core.close()

    # -- zero-copy path for torch policies living on the same GPU ------------------------------------------
    def step_torch(self, actions, obs_out=None, reward_out=None, done_out=None, terminal_out=None):
        """actions: float32 CUDA tensor [num_envs, num_cars] -> (obs f32 [E,obs_dim], reward f64 [E], done u8 [E]) on the
        GPU, asynchronous on torch's current stream (the handle adopts it: ordered after the ops that produced `actions`, and
        ops enqueued afterwards see the outputs)."""
        import torch

        dev = actions.device
        # the handle must launch on the stream that produced `actions` and will consume the outputs: its own stream is
        # non-blocking, i.e. not ordered with torch's
        cur = torch.cuda.current_stream(dev).cuda_stream
        if getattr(self, "_torch_stream", None) != cur:
            self.core.batch.set_stream(cur)
            self._torch_stream = cur
        E, D = self.num_envs, self.core.obs_dim
        a = actions.contiguous()
        dt = _capi.ACT_F64 if a.dtype == torch.float64 else _capi.ACT_F32
        if dt == _capi.ACT_F32:
            a = a.to(torch.float32)
        obs = obs_out if obs_out is not None else torch.empty((E, D), device=dev, dtype=torch.float32)
        rew = reward_out if reward_out is not None else torch.empty(E, device=dev, dtype=torch.float64)
        done = done_out if done_out is not None else torch.empty(E, device=dev, dtype=torch.uint8)
        self.core.batch.step_dev(a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(),
                                 None if terminal_out is None else terminal_out.data_ptr(), act_dtype=dt)
        return obs, rew, done


class FleetVectorEnv(_GymVectorEnv):
    """gymnasium vector-env signature (what RLlib / CleanRL style loops expect); a subclass of `gymnasium.vector.VectorEnv`
    when gymnasium is installed."""

    def __init__(self, env_config, num_envs: int, **kw):
        self.copy_obs = bool(kw.pop("copy_obs", True))  # see FleetVecEnv
        self.core = FleetCore(env_config, num_envs, auto_reset=True, **kw)
        if _GymVectorEnv is not object:
            try:  # gymnasium 0.29: VectorEnv(num_envs, observation_space, action_space); 1.x: no arguments
                _GymVectorEnv.__init__(self, self.core.num_envs, self.core.single_observation_space, self.core.single_action_space)
            except TypeError:
                _GymVectorEnv.__init__(self)
        self.num_envs = self.core.num_envs
        self.single_observation_space = self.core.single_observation_space
        self.single_action_space = self.core.single_action_space
        lo, hi = self.single_observation_space.low, self.single_observation_space.high
        self.observation_space = Box(low=np.tile(lo, (self.num_envs, 1)), high=np.tile(hi, (self.num_envs, 1)), dtype=np.float32)
        self.action_space = Box(low=-1, high=1, shape=(self.num_envs, self.core.num_cars), dtype=np.float32)

    def reset(self, *, seed=None, options=None):
        self.core.clear_start_overrides()
        return self.core.batch.reset(), {}

    def step(self, actions):
        obs, rew, done, term = self.core.batch.step(np.asarray(actions).reshape(self.core.num_envs, -1), copy=self.copy_obs)
        terminated = done.astype(bool)
        truncated = np.zeros(self.num_envs, dtype=bool)  # the reference always returns truncated=False (:702)
        infos = {}
        if terminated.any():
            final = np.empty(self.num_envs, dtype=object)
            for i in np.nonzero(terminated)[0]:
                final[i] = term[i].copy()
            infos = {"final_observation": final, "_final_observation": terminated.copy()}
        return obs, rew, terminated, truncated, infos

    def close(self, **kwargs):
        self.core.close()
        self.closed = True


class FleetEnv(_GymEnv):
    """Single environment with the reference's exact call signatures (`class FleetEnv(gym.Env)`, fleet_environment.py:50: a
    subclass of `gymnasium.Env` when gymnasium is installed; no auto-reset)."""

    metadata = {"render_modes": ["human"]}

    def __init__(self, env_config, **kw):
        self.core = FleetCore(env_config, 1, auto_reset=False, **kw)
        self.env_config = self.core.env_config
        self.num_cars = self.core.num_cars
        self.observation_space = self.core.single_observation_space
        self.action_space = self.core.single_action_space
        self.render_mode = "human"  # :327
        self.info: dict = {}        # :235

    def reset(self, **kwargs):
        """:330-434 -- ignores seed/options like the reference."""
        self.core.clear_start_overrides()
        obs = self.core.batch.reset()
        return obs[0], self.info

    def step(self, actions):
        """:436-702 -> (obs float32[obs_dim], reward float, done bool, truncated False, info {})."""
        a = np.asarray(actions)
        a = a.reshape(1, -1) if a.dtype == np.float64 else a.astype(np.float32).reshape(1, -1)
        obs, rew, done, _ = self.core.batch.step(a)
        return obs[0], float(rew[0]), bool(done[0]), False, self.info

    def close(self):
        self.core.close()
        return None

    def render(self):
        return None  # the matplotlib parking-lot picture is UI, out of scope

    def get_log(self):
        return self.core.get_log()[0]

    def is_done(self):
        return self.core.is_done()[0]

    def get_start_time(self):
        return self.core.get_start_time()[0]

    def set_start_time(self, start_time):
        self.core.set_start_time(start_time)
        return None

    def get_time(self):
        return self.core.get_time()[0]

    def get_dist_factor(self):
        return self.core.get_dist_factor()[0]
