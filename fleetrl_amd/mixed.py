"""Several env groups with their own tables / parameters behind ONE vector env on one GPU.

BASELINE.json's largest configuration mixes fleet types ("65536 envs x 200 EVs, mixed lmd/ct/ut schedules"): envs of one
type share tables and scalars, so each type is one handle of the C ABI (one `FleetBatch`), and all handles write into
slices of the same device buffers.  Every handle owns a HIP stream, so the step kernels of the groups run concurrently;
actions go to the device in one copy and observations come back in one copy.  The reference has no counterpart (its envs
are separate OS processes that may of course be configured differently, complete_pipeline.ipynb cell 13).
"""
from __future__ import annotations

import numpy as np

from . import _capi
from .vec_env import FleetCore, _SB3VecEnv

__all__ = ["FleetMixedVecEnv"]


class FleetMixedVecEnv(_SB3VecEnv):
    """stable-baselines3 `VecEnv` (a subclass of it when SB3 is installed, see vec_env.py) over several `FleetCore` groups.  `groups` = [(env_config, num_envs, kwargs)],
    kwargs as for `FleetCore` (e.g. `tables=`); all groups must have the same number of EVs and observation layout.
    Env ids (and with them the Philox start-row streams) run through the groups in order."""

    def __init__(self, groups, device: int = 0):
        import torch

        self._torch = torch
        self.cores, self.offsets = [], []
        off = 0
        for g in groups:
            cfg, n = g[0], int(g[1])
            kw = dict(g[2]) if len(g) > 2 else {}
            self.cores.append(FleetCore(cfg, n, auto_reset=True, device=device, env_id_offset=off, **kw))
            self.offsets.append(off)
            off += n
        self.num_envs = off
        c0 = self.cores[0]
        if any(c.obs_dim != c0.obs_dim or c.num_cars != c0.num_cars for c in self.cores):
            raise ValueError("all groups need the same number of EVs and the same observation flags")
        self.obs_dim, self.num_cars = c0.obs_dim, c0.num_cars
        if _SB3VecEnv is not object:
            _SB3VecEnv.__init__(self, off, c0.single_observation_space, c0.single_action_space)
        else:
            self.observation_space, self.action_space = c0.single_observation_space, c0.single_action_space
            self.render_mode = None
            self.reset_infos = [{} for _ in range(off)]
        dev = torch.device("cuda", device)
        E, D, N = self.num_envs, self.obs_dim, self.num_cars
        self._obs = torch.zeros((E, D), device=dev, dtype=torch.float32)
        self._term = torch.zeros((E, D), device=dev, dtype=torch.float32)
        self._act = torch.zeros((E, N), device=dev, dtype=torch.float32)
        self._rew = torch.zeros(E, device=dev, dtype=torch.float64)
        self._done = torch.zeros(E, device=dev, dtype=torch.uint8)

    def _slices(self):
        for core, off in zip(self.cores, self.offsets):
            yield core, off, off + core.num_envs

    def reset(self):
        self._torch.cuda.current_stream().synchronize()  # the buffers' fills ran on torch's stream, the handles have their own
        for core, lo, hi in self._slices():
            core.batch.reset_dev(self._obs[lo:hi].data_ptr())
        for core in self.cores:
            core.batch.synchronize()
        return self._obs.cpu().numpy()

    def step(self, actions):
        torch = self._torch
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self.num_cars))
        self._act.copy_(torch.from_numpy(a))
        torch.cuda.current_stream().synchronize()  # the handles' streams are not torch's
        for core, lo, hi in self._slices():  # asynchronous, one stream per group: the groups' kernels overlap
            core.batch.step_dev(self._act[lo:hi].data_ptr(), self._obs[lo:hi].data_ptr(), self._rew[lo:hi].data_ptr(),
                                self._done[lo:hi].data_ptr(), self._term[lo:hi].data_ptr(), act_dtype=_capi.ACT_F32)
        for core in self.cores:
            core.batch.synchronize()
        obs = self._obs.cpu().numpy()
        rew = self._rew.cpu().numpy()
        dones = self._done.cpu().numpy().astype(bool)
        infos = [{} for _ in range(self.num_envs)]
        if dones.any():
            term = self._term.cpu().numpy()
            for core, lo, hi in self._slices():
                if not dones[lo:hi].any():
                    continue
                ret, ln = core.batch.get("last_ep_return"), core.batch.get("last_ep_len")
                for i in np.nonzero(dones[lo:hi])[0]:
                    infos[lo + i] = {"terminal_observation": term[lo + i].copy(), "TimeLimit.truncated": False,
                                     "episode": {"r": float(ret[i]), "l": int(ln[i])}}
        return obs, rew.astype(np.float32), dones, infos

    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        return self.step(self._pending)

    def env_method(self, name, *args, indices=None, **kwargs):
        """Fan a getter of the reference's env out over the groups (`is_done`, `get_time`, `get_start_time`,
        `get_dist_factor`); returns one entry per env."""
        out = []
        for core in self.cores:
            out.extend(getattr(core, name)(*args, **kwargs))
        return out if indices is None else [out[i] for i in indices]

    def get_attr(self, attr_name: str, indices=None):
        c0 = self.cores[0]
        per_env = {"num_cars": self.num_cars, "render_mode": None, "observation_space": c0.single_observation_space,
                   "action_space": c0.single_action_space}
        if attr_name not in per_env:
            raise AttributeError(attr_name)
        n = self.num_envs if indices is None else (1 if isinstance(indices, int) else len(list(indices)))
        return [per_env[attr_name]] * n

    def set_attr(self, attr_name: str, value, indices=None):
        raise AttributeError(f"attribute {attr_name!r} cannot be set on the fused batch")

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else (1 if isinstance(indices, int) else len(list(indices)))
        return [False] * n

    def seed(self, seed=None):
        return [None] * self.num_envs  # the reference ignores reset(seed=...) (quirk Q12)

    def close(self):
        for core in self.cores:
            core.close()
