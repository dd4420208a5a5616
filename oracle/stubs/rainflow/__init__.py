"""Stand-in for the third-party `rainflow` package, pinned by the reference at 3.2.0
(/root/reference/requirements.txt:147; only call site on the hot path:
/root/reference/fleetrl/utils/battery_degradation/rainflow_sei_degradation.py:132).

TEST INFRASTRUCTURE ONLY.  The package is not vendored in /root/reference and cannot
be fetched (no network), so this restates its published algorithm (ASTM E1049-85
three-point rainflow, section 5.4.4) with the package's documented tuple order
`(range, mean, count, i_start, i_end)`:

* reversals: the first sample (index 0), every sample at which the slope changes sign
  strictly (`d_last * d_next < 0`; a sample equal to its predecessor is skipped, so a
  plateau is reported at the position of its last sample), and always the last sample
  (index len-1);
* cycles: while at least three points are held, X = |p3-p2|, Y = |p2-p1|; X < Y ->
  read on; exactly three points -> Y is a half cycle and the first point is dropped;
  otherwise Y is a full cycle and its two points are dropped; the leftovers are half
  cycles in order.

"Parity unpinned" at this boundary by the reference itself (it has no test for it);
pinned here by the ASTM worked example (tests/test_oracle_rainflow.py).
"""
from collections import deque

__version__ = "3.2.0-standin"


def reversals(series):
    it = iter(series)
    x_last, x = next(it, None), next(it, None)
    if x_last is None or x is None:
        return
    d_last = x - x_last
    yield 0, x_last
    index = None
    x_next = None
    for index, x_next in enumerate(it, start=1):
        if x_next == x:
            continue
        d_next = x_next - x
        if d_last * d_next < 0:
            yield index, x
        x, d_last = x_next, d_next
    if index is not None:
        yield index + 1, x_next


def extract_cycles(series):
    points = deque()

    def fmt(p1, p2, count):
        i1, x1 = p1
        i2, x2 = p2
        return abs(x1 - x2), 0.5 * (x1 + x2), count, i1, i2

    for point in reversals(series):
        points.append(point)
        while len(points) >= 3:
            x1, x2, x3 = points[-3][1], points[-2][1], points[-1][1]
            X = abs(x3 - x2)
            Y = abs(x2 - x1)
            if X < Y:
                break
            elif len(points) == 3:
                yield fmt(points[0], points[1], 0.5)
                points.popleft()
            else:
                yield fmt(points[-3], points[-2], 1.0)
                last = points.pop()
                points.pop()
                points.pop()
                points.append(last)
    while len(points) > 1:
        yield fmt(points[0], points[1], 0.5)
        points.popleft()


def count_cycles(series, ndigits=None, nbins=None, binsize=None):
    counts = {}
    for rng, _mean, count, _i0, _i1 in extract_cycles(series):
        key = round(rng, ndigits) if ndigits is not None else rng
        counts[key] = counts.get(key, 0.0) + count
    return sorted(counts.items())
