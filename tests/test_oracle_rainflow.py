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


def _reversals(series):
    """First sample, every strict turning point, last sample (rainflow.reversals as the package documents it)."""
    s = list(series)
    pts = [s[0]]
    prev, slope = s[0], 0
    for x in s[1:]:
        if x == prev:
            continue
        sl = 1 if x > prev else -1
        if slope and sl != slope:
            pts.append(prev)
        slope, prev = sl, x
    pts.append(s[-1])
    return pts


def _astm_by_rescanning(points):
    """ASTM E1049-85 section 5.4.4 as the standard words it -- a list that is re-read from its starting point after every
    count, points DELETED from it -- instead of the stack the package (and the oracle, and the kernels) keep.  Returns the
    multiset of (range, mean, count)."""
    pts = list(points)
    out = []
    i = 0  # index of the first point of range Y
    while True:
        if i + 2 >= len(pts):
            break
        y = abs(pts[i + 1] - pts[i])
        x = abs(pts[i + 2] - pts[i + 1])
        if x < y:
            i += 1  # step 2: read the next point
            continue
        if i == 0:  # Y contains the starting point: half cycle, discard the first point, the start moves to the second
            out.append((y, 0.5 * (pts[0] + pts[1]), 0.5))
            del pts[0]
        else:  # one cycle, discard both points of Y
            out.append((y, 0.5 * (pts[i] + pts[i + 1]), 1.0))
            del pts[i:i + 2]
        i = 0  # rule 5.4.4 step 2 again from the starting point
    for a, b in zip(pts[:-1], pts[1:]):  # step 6: each remaining range is a half cycle
        out.append((abs(b - a), 0.5 * (a + b), 0.5))
    return sorted(out)


def test_counts_equal_the_standard_s_list_deleting_procedure():
    """The oracle's stack form against the standard's own wording on random walks, saturating series (plateaus, exactly equal
    extremes), zigzags of growing and shrinking ranges; values on a dyadic grid so that ranges and means are exact."""
    rng = np.random.default_rng(7)
    series = []
    for n in (5, 17, 96, 193, 400):
        series.append(np.cumsum(rng.integers(-40, 41, size=n)) / 1024.0)
        series.append(np.clip(np.cumsum(rng.integers(-300, 301, size=n)) / 1024.0, 0.0, 1.0))  # saturates: plateaus
    series.append(np.array([(-1) ** k * k for k in range(40)]) / 64.0)        # growing ranges: a closure at every point
    series.append(np.array([(-1) ** k * (40 - k) for k in range(40)]) / 64.0)  # shrinking ranges: nothing closes
    for s in series:
        pts = _reversals(s)
        if len(pts) < 2 or len(s) < 3:
            continue
        got = sorted((float(r), float(m), float(c)) for r, m, c, _e in rainflow(s))
        assert got == _astm_by_rescanning(pts)


def test_structural_invariants():
    """(i) every reversal is consumed exactly once: 2 * full + half == reversals - 1; (ii) the largest range is max - min;
    (iii) negation, scaling by a power of two and a shift on the same dyadic grid leave counts and end indices unchanged."""
    rng = np.random.default_rng(11)
    for n in (3, 9, 50, 193, 777):
        s = np.cumsum(rng.integers(-25, 26, size=n)) / 512.0
        cyc = rainflow(s)
        pts = _reversals(s)
        full, half = (cyc[:, 2] == 1.0).sum(), (cyc[:, 2] == 0.5).sum()
        assert 2 * full + half == len(pts) - 1
        assert cyc[:, 0].max() == s.max() - s.min()
        t = rainflow(-2.0 * s + 3.0)
        np.testing.assert_array_equal(t[:, 0], 2.0 * cyc[:, 0])
        np.testing.assert_array_equal(t[:, 1], -2.0 * cyc[:, 1] + 3.0)
        np.testing.assert_array_equal(t[:, 2:], cyc[:, 2:])
