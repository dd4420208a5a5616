"""The rainflow count stops at the episode's last degradation row (EnvRec::rf_until, include/fleet_hip.h
fleet_set_rainflow_count_all).

The reference logs one SOC sample per EV and step (fleet_environment.py:655), counts cycles over that log only on the 14:45 rows
(:665, rainflow_sei_degradation.py:132) and clears the log in reset() (:338-339): what is logged after an episode's last 14:45 row is
never read.  The kernels count while they log, so they stop there -- and everything the reference can see must be exactly what it
is with the count running to the end of every episode.  Needs an MI355X."""
import numpy as np
import pytest

from golden_util import load_trace, params_for
from fleetrl_amd import _capi

pytestmark = pytest.mark.gpu

VISIBLE = ("soc", "soc_deg", "soh", "hours_left", "target_soc", "time_idx", "start_idx", "rf_len", "fd_cyc", "fd_cal", "sei_l", "episodes",
           "ep_return", "last_ep_return", "last_ep_len", "cashflow", "penalty_record", "error_bits", "done")


def _pair(name, E, ep_steps=None, seed=21):
    from fleetrl_amd.batch import FleetBatch

    import dataclasses

    g = load_trace(name)
    if ep_steps is not None:  # shorter episodes: `episode_length` is in hours
        per_hour = 60 // g.rc.minutes
        assert ep_steps % per_hour == 0
        g.rc = dataclasses.replace(g.rc, episode_length=ep_steps // per_hour)
    p = params_for(g, num_envs=E)
    steps = ep_steps or g.ep_steps
    assert p.episode_steps == steps
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, g.tables.T - steps - 60, size=(7, E)).astype(np.int32)
    out = []
    for count_all in (False, True):
        b = FleetBatch(p, g.tables, g.time_feat)
        b.set_start_schedule(starts)
        if count_all:
            b.set_rainflow_count_all(True)
        out.append(b)
    return g, out[0], out[1], rng, steps


@pytest.mark.parametrize("ep_steps", [None, 40])
def test_stopping_the_count_changes_nothing_the_reference_can_see(ep_steps):
    """Two batches, one stopping the count at the last degradation row (default), one counting to the end of every episode, fed
    the same actions for more than three episodes: observations, rewards, done flags and every state field the reference has are
    identical after EVERY step; the count itself differs exactly where it is allowed to.  With 40-step episodes (10 h) most
    episodes hold no degradation row at all (rf_until = -1: nothing is counted) and some hold one."""
    E = 96
    g, a, b, rng, steps = _pair("ct5_both_rainflow", E, ep_steps)
    oa, ob = a.reset(), b.reset()
    np.testing.assert_array_equal(oa, ob)
    froze = differed = 0
    for k in range(3 * steps + 17):
        act = rng.uniform(-1, 1, size=(E, g.N)).astype(np.float32)
        act[rng.random((E, g.N)) < 0.15] = 0.0
        ra, rb = a.step(act), b.step(act)
        for x, y, what in zip(ra[:3], rb[:3], ("obs", "reward", "done")):
            np.testing.assert_array_equal(x, y, err_msg=f"{what}, step {k}")
        if k % 7 == 0 or k > 3 * steps:
            for f in VISIBLE:
                np.testing.assert_array_equal(a.get(f), b.get(f), err_msg=f"{f}, step {k}")
        # the count: identical while the episode's last degradation row is still ahead, frozen after it
        t, until = a.get("time_idx"), a.get("rf_until")
        live = t <= until
        ca, cb = a.get("rf_cycles"), b.get("rf_cycles")
        np.testing.assert_array_equal(ca[live], cb[live], err_msg=f"rf_cycles of envs whose count is live, step {k}")
        assert (ca <= cb).all()
        froze += int((~live).sum())
        differed += int((ca != cb).any(axis=1).sum())
    assert froze > 0 and differed > 0 and differed <= froze
    assert (b.get("rf_until") == np.iinfo(np.int32).max).all()
    a.check_errors(); b.check_errors()
    a.close(); b.close()


def test_rf_until_is_the_last_degradation_row_of_the_episode():
    """FLEET_F_RF_UNTIL against the table: the last row in (start, finish] with hour == 14 and minute == 45, -1 when there is none --
    after reset and after in-step auto-resets (full-length and 40-step episodes)."""
    from fleetrl_amd.batch import FleetBatch

    for ep_steps in (None, 40):
        g, a, b, rng, steps = _pair("ct5_both_rainflow", 64, ep_steps)
        hour, minute = g.tables.hour, g.tables.minute
        deg = (np.asarray(hour) == 14) & (np.asarray(minute) == 45)

        def want(start):
            rows = np.arange(start + 1, min(start + steps, len(deg) - 1) + 1)
            hit = rows[deg[rows]]
            return int(hit[-1]) if len(hit) else -1

        a.reset()
        for k in range(steps + 5):  # over the first auto-reset
            if k in (0, steps - 1, steps, steps + 4):
                got, st = a.get("rf_until"), a.get("start_idx")
                np.testing.assert_array_equal(got, np.array([want(int(s)) for s in st], np.int32), err_msg=f"step {k}")
            a.step(rng.uniform(-1, 1, size=(64, g.N)).astype(np.float32))
        if ep_steps == 40:
            assert (a.get("rf_until") == -1).any() and (a.get("rf_until") >= 0).any()
        a.close(); b.close()


def test_stopped_count_through_every_launch_path():
    """The same property through the K-step kernel, the hipGraph and the library's own queue at the bench's shape family: the batch
    that stops its count and the one that does not end in the same state, and each launch path of the stopping batch equals the
    single-step stream launches."""
    import torch

    E = 512
    g, a, b, rng, steps = _pair("ct5_both_rainflow", E)
    dev = torch.device("cuda", 0)
    L = 16
    tape = torch.from_numpy(rng.uniform(-1, 1, size=(L, E, g.N)).astype(np.float32)).to(dev)
    outs = []
    for x, mode in ((a, _capi.LAUNCH_DIRECT), (b, _capi.LAUNCH_GRAPH)):
        o = (torch.zeros((E, x.obs_dim), device=dev), torch.zeros(E, device=dev, dtype=torch.float64), torch.zeros(E, device=dev, dtype=torch.uint8))
        x.reset_dev(o[0].data_ptr())
        for n in (steps - 3, 7, steps + 50):
            x.run_tape_dev(n, tape.data_ptr(), L, *(t.data_ptr() for t in o), use_graph=mode)
            x.synchronize()
        outs.append(o)
    for k in range(3):
        np.testing.assert_array_equal(outs[0][k].cpu().numpy(), outs[1][k].cpu().numpy())
    for f in VISIBLE:
        np.testing.assert_array_equal(a.get(f), b.get(f), err_msg=f)
    # K steps per launch (the multi-step kernel compares the row itself instead of reading the head's bit)
    rsum = torch.zeros(E, device=dev, dtype=torch.float64)
    dcnt = torch.zeros(E, device=dev, dtype=torch.int32)
    for x, o in ((a, outs[0]), (b, outs[1])):
        x.step_many_dev(L, tape.data_ptr(), o[0].data_ptr(), rsum.data_ptr(), dcnt.data_ptr())
        x.synchronize()
    for f in VISIBLE:
        np.testing.assert_array_equal(a.get(f), b.get(f), err_msg=f"{f} after a K-step launch")
    a.check_errors(); b.check_errors()
    a.close(); b.close()
