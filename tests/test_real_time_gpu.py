"""real_time (event-skipping) mode on the GPU against traces of the unmodified reference (oracle/gen_golden.py rt).
Needs an MI355X."""
import numpy as np
import pytest

from golden_util import RT_TRACE_NAMES, load_rt_trace, params_for, replay_rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", RT_TRACE_NAMES)
def test_hip_real_time_matches_reference(name):
    from fleetrl_amd.batch import FleetBatch

    g = load_rt_trace(name)
    hip = FleetBatch(params_for(g), g.tables, g.time_feat)
    worst = replay_rt(g, hip, float_rtol=1e-9, obs_exact=False)
    assert worst["soc"] < 1e-9 and worst["soh"] < 1e-9 and worst["reward"] < 1e-9
    hip.close()


def test_real_time_multi_step_entries_are_refused():
    import torch

    from fleetrl_amd._capi import POLICY_UNCONTROLLED, FleetHipError
    from fleetrl_amd.batch import FleetBatch

    g = load_rt_trace(RT_TRACE_NAMES[0])
    hip = FleetBatch(params_for(g), g.tables, g.time_feat)
    hip.reset()
    dev = torch.device("cuda", 0)
    obs = torch.zeros((g.E, hip.obs_dim), device=dev)
    rs = torch.zeros(g.E, device=dev, dtype=torch.float64)
    tape = torch.zeros((4, g.E, g.N), device=dev)
    with pytest.raises(FleetHipError, match="real_time"):
        hip.step_many_dev(4, tape.data_ptr(), obs.data_ptr(), rs.data_ptr())
    with pytest.raises(FleetHipError, match="real_time"):
        hip.rollout_policy_dev(POLICY_UNCONTROLLED, 4, obs.data_ptr(), rs.data_ptr())
    hip.close()
