"""Host side of the device-resident benchmark policies.

The reference compares its agents against three rule-based charging strategies, each a Python loop that calls
`env.step` once per row with an action computed on the host (benchmarking/uncontrolled_charging.py:51-54,
distributed_charging.py:50-54, night_charging.py:81-98).  Here the rule is evaluated inside the multi-step HIP kernel
(`fleet_rollout_policy_dev`), so a whole evaluation year of a batch of fleets is a handful of launches.  This module
holds what the reference computes *before* its loop (the night-charging window) and a small driver.
"""
from __future__ import annotations

import math

import numpy as np

from . import _capi

__all__ = ["night_schedule", "run_policy", "POLICIES"]

POLICIES = {"uncontrolled": _capi.POLICY_UNCONTROLLED, "distributed": _capi.POLICY_DISTRIBUTED, "night": _capi.POLICY_NIGHT}


def night_schedule(tables, *, target_soc: float, init_battery_cap: float, charging_eff: float, evse_power: float):
    """(charging_hour, charging_minute, max_hours) of the night-charging benchmark, night_charging.py:50-73.

    The reference looks for the earliest clock time at which any vehicle leaves home (`Location` 'home' -> 'driving' on
    consecutive rows of the ID-major frame, so the last row of vehicle c-1 also precedes the first row of vehicle c),
    subtracts the time a full charge from 0 to the target SOC takes, and rounds the result to an hour and the nearest
    quarter.  `Location == 'home'` coincides with `There == 1` in every schedule the reference ships or generates."""
    there = np.asarray(tables.there) != 0  # [T, N]
    flat = there.T.reshape(-1)  # ID-major, like the reference's frame
    leaving = np.zeros(flat.shape, dtype=bool)
    leaving[1:] = flat[:-1] & ~flat[1:]
    if not leaving.any():
        raise ValueError("no departure in the schedule: the night-charging window is undefined")
    rows = np.nonzero(leaving)[0] % there.shape[0]
    hour = np.asarray(tables.hour, dtype=np.int64)[rows]
    minute = np.asarray(tables.minute, dtype=np.int64)[rows]
    k = np.argmin(hour * 60 + minute)
    earliest_dep = int(hour[k]) + int(minute[k]) / 60
    max_time_needed = target_soc * init_battery_cap / charging_eff / evse_power  # :63
    starting_time = 24 + (earliest_dep - max_time_needed)
    if starting_time > 24:
        starting_time = 23.99  # "always start just before midnight" :67
    frac, whole = math.modf(starting_time)
    charging_hour = int(whole)
    quarters = np.asarray([0, 15, 30, 45])
    charging_minute = int(quarters[np.abs(quarters - int(frac * 60)).argmin()])
    return charging_hour, charging_minute, int(max_time_needed)


def run_policy(batch, policy, steps: int, *, chunk: int = 96, night=None):
    """Advance every env of `batch` (a `FleetBatch`, already reset) by `steps` rows under a built-in policy.

    Returns (obs f32 [E, obs_dim] after the last step, reward_sum f64 [E], done_count i32 [E]).  `night` =
    (charging_hour, charging_minute, max_hours) is required for the night policy the first time it is used on a batch
    (it clears the per-env window state).  The rule runs on the device; `chunk` only bounds the length of one launch."""
    import torch

    pol = POLICIES[policy] if isinstance(policy, str) else int(policy)
    if night is not None:
        batch.set_night_policy(*night)
    dev = torch.device("cuda", batch.device)
    batch.use_torch_stream(dev)  # launches and the torch accumulations below on ONE stream: ordered without host round trips
    E = batch.E
    obs = torch.zeros((E, batch.obs_dim), device=dev, dtype=torch.float32)
    rsum = torch.zeros(E, device=dev, dtype=torch.float64)
    dcnt = torch.zeros(E, device=dev, dtype=torch.int32)
    rtot = torch.zeros(E, device=dev, dtype=torch.float64)
    dtot = torch.zeros(E, device=dev, dtype=torch.int32)
    left = int(steps)
    while left > 0:
        k = min(left, int(chunk))
        batch.rollout_policy_dev(pol, k, obs.data_ptr(), rsum.data_ptr(), dcnt.data_ptr())
        rtot += rsum
        dtot += dcnt
        left -= k
    batch.check_errors()
    return obs.cpu().numpy(), rtot.cpu().numpy(), dtot.cpu().numpy()
