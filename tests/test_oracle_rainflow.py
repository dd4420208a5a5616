"""Rainflow counting of the oracle against the ASTM E1049-85 worked example (the only published pin for the
third-party `rainflow` package the reference depends on) and a few structural edge cases."""
import numpy as np

from oracle.fleet_oracle import rainflow


def _count(cycles):
    out = {}
    for rng, _m, c, _e in cycles:
        out[rng] = out.get(rng, 0.0) + c
    return out


def test_astm_e1049_example():
    cyc = rainflow([-2, 1, -3, 5, -1, 3, -4, 4, -2])
    assert _count(cyc) == {3.0: 0.5, 4.0: 1.5, 6.0: 0.5, 8.0: 1.0, 9.0: 0.5}
    # emission order and (mean, count, i_end) of every cycle, as the rainflow package yields them
    want = [(3, -0.5, 0.5, 1), (4, -1.0, 0.5, 2), (4, 1.0, 1.0, 5), (8, 1.0, 0.5, 3), (9, 0.5, 0.5, 6), (8, 0.0, 0.5, 7), (6, 1.0, 0.5, 8)]
    np.testing.assert_array_equal(cyc, np.array(want, dtype=np.float64))


def test_short_and_flat_series():
    assert len(rainflow([0.5])) == 0
    assert len(rainflow([0.5, 0.7])) == 0  # two samples give a single reversal -> no cycle (reference: degradation 0)
    cyc = rainflow([0.5, 0.5, 0.5, 0.5])
    assert cyc.shape == (1, 4) and cyc[0, 0] == 0.0 and cyc[0, 3] == 3  # first + last sample only
    cyc = rainflow([0.2, 0.4, 0.4, 0.4, 0.1])  # plateau: reversal reported at the last sample of the plateau
    np.testing.assert_allclose(cyc, [[0.2, 0.3, 0.5, 3], [0.3, 0.25, 0.5, 4]])


def test_last_cycle_always_ends_at_last_sample():
    rng = np.random.default_rng(0)
    for n in (3, 10, 97, 193):
        s = rng.random(n)
        cyc = rainflow(s)
        assert cyc[:, 3].max() == n - 1
        assert np.all((cyc[:, 2] == 0.5) | (cyc[:, 2] == 1.0))
