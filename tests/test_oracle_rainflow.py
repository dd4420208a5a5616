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


# ---------------------------------------------------------------------------------------------------------------------------
# A second, independently written counting rule: the FOUR-point method (range-pair form; Amzallag et al. 1994,
# the form most fatigue codes use).  Four successive reversals S1..S4: when the inner range |S3 - S2| is not larger than both
# outer ranges, (S2, S3) is a closed cycle and is deleted; what cannot be closed is the residue.  McInnes & Meehan (2008,
# "Equivalence of four-point and three-point rainflow cycle counting algorithms") show that the three-point ASTM rule the
# `rainflow` package implements yields the same closed cycles, and that its half cycles are the successive ranges of the
# four-point residue.  Different program, different data flow (no "contains the starting point" case at all) -- same counts.
# ---------------------------------------------------------------------------------------------------------------------------
def _four_point(points):
    stack, full = [], []
    for p in points:
        stack.append(p)
        while len(stack) >= 4:
            s1, s2, s3, s4 = stack[-4:]
            inner = abs(s3 - s2)
            if inner <= abs(s2 - s1) and inner <= abs(s4 - s3):
                full.append((inner, 0.5 * (s2 + s3)))
                del stack[-3:-1]
            else:
                break
    return full, stack  # closed cycles, residue


def _histogram(cycles):
    """{range: [sum of counts, count-weighted sum of means]}: two half cycles of one range are one full cycle of it."""
    h = {}
    for rng, mean, count in cycles:
        e = h.setdefault(float(rng), [0.0, 0.0])
        e[0] += count
        e[1] += count * mean
    return h


def _assert_same_counts(series, what=""):
    pts = _reversals(series)
    full, residue = _four_point(pts)
    four = [(r, m, 1.0) for r, m in full] + [(abs(b - a), 0.5 * (a + b), 0.5) for a, b in zip(residue[:-1], residue[1:])]
    three = [(r, m, c) for r, m, c, _e in rainflow(series)]
    h4, h3 = _histogram(four), _histogram(three)
    assert sorted(h4) == sorted(h3), f"ranges differ {what}"
    for r in h3:
        assert h4[r][0] == h3[r][0], f"count of range {r} {what}: four-point {h4[r][0]}, three-point {h3[r][0]}"
        assert abs(h4[r][1] - h3[r][1]) <= 1e-12 * max(1.0, abs(h3[r][1])), f"means of range {r} {what}"
    # the damage-relevant totals the SEI model consumes (rainflow_sei_degradation.py:140,170-174): number of cycles' worth,
    # sum of ranges * counts
    assert sum(c for _r, _m, c in four) == sum(c for _r, _m, c in three)


def test_four_point_method_gives_the_same_counts_on_synthetic_series():
    rng = np.random.default_rng(23)
    for n in (4, 5, 9, 33, 96, 193, 500, 2000):
        for _ in range(20):
            _assert_same_counts(np.cumsum(rng.integers(-40, 41, size=n)) / 1024.0, f"(random walk, n={n})")
            _assert_same_counts(np.clip(np.cumsum(rng.integers(-300, 301, size=n)) / 1024.0, 0.0, 1.0), f"(saturating, n={n})")
    _assert_same_counts(np.array([(-1) ** k * k for k in range(40)]) / 64.0, "(growing zigzag)")
    _assert_same_counts(np.array([(-1) ** k * (40 - k) for k in range(40)]) / 64.0, "(shrinking zigzag)")
    _assert_same_counts([-2, 1, -3, 5, -1, 3, -4, 4, -2], "(ASTM E1049-85 example)")


def test_three_implementations_agree_on_the_soc_series_of_every_rainflow_golden():
    """The stand-in for rainflow==3.2.0 (through which every rainflow-mode golden was produced), the C oracle's own rainflow and
    the four-point method on the logged SOC series (LogDataDeg.soc_log: the reset sample + one per step) of every EV of every
    episode of every rainflow-mode golden trace."""
    import os
    import sys

    from golden_util import TRACE_NAMES, load_trace

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "stubs"))
    import rainflow as standin  # noqa: E402  (oracle/stubs/rainflow: test infrastructure)

    seen = 0
    for name in TRACE_NAMES:
        g = load_trace(name)
        if g.rc.deg_mode != 2:
            continue
        for e in range(g.E):
            for ep in range(g.episodes):
                for c in range(g.N):
                    first = g.reset_soc[e, ep, c]
                    first = g.rc.def_soc if first == 0 else first  # fleet_environment.py:395-399
                    s = np.concatenate([[first], g.soc_deg[e, ep * g.ep_steps:(ep + 1) * g.ep_steps, c]])
                    what = f"({name}, env {e}, episode {ep}, EV {c})"
                    _assert_same_counts(s, what)
                    a = [(r, m, cnt, i1) for r, m, cnt, _i0, i1 in standin.extract_cycles(list(s))]
                    b = [tuple(row) for row in rainflow(s)]
                    assert a == b, f"stand-in and C oracle differ {what}"
                    seen += 1
    assert seen >= 100
