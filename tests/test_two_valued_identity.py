"""CPU: the identity behind the far-field y sweep's two-valued tiles (sdfgpu_envelope_dc.hpp, EnvDcArgs::flat_on).

For a line whose entries outside its zero sites take at most two values mn <= mx ("no site" = a value above every distance counts as a
value), the lower envelope  D(p) = min_q F(q) + (p - q)^2  equals  min(mx, d0(p)^2, mn + d1(p)^2)  with d0 / d1 = distance along the line
to the nearest zero site / mn site.  Checked here against the brute-force envelope on random lines -- walls, pillars, plates, lines without a
zero site, without an mn site, all zero, "no site" as the larger value -- and the restatement below is what the kernel's three phases compute
(chunk masks -> nearest site before / behind each chunk -> per-position chains).  A third value must make the classifier refuse the line."""
import numpy as np
import pytest

INF = 1 << 40


def brute(F):
    L = len(F)
    q = np.arange(L)
    return (F[None, :] + (q[:, None] - q[None, :]) ** 2).min(axis=1)


def nearest(mask):
    """distance along the line to the nearest set position (INF where there is none), the way the kernel does it: first / last set position
    of every chunk of 8, exclusive prefix maximum / suffix minimum over the chunks, then chains inside the chunk"""
    L = len(mask)
    M = (L + 7) // 8
    first = np.full(M, INF, np.int64)
    last = np.full(M, -INF, np.int64)
    for c in range(M):
        idx = np.flatnonzero(mask[8 * c:8 * c + 8])
        if len(idx):
            first[c], last[c] = 8 * c + idx[0], 8 * c + idx[-1]
    before = np.concatenate(([-INF], np.maximum.accumulate(last)[:-1]))
    behind = np.concatenate((np.minimum.accumulate(first[::-1])[::-1][1:], [INF]))
    d = np.empty(L, np.int64)
    for c in range(M):
        lo, hi = before[c], behind[c]
        ps = range(8 * c, min(8 * c + 8, L))
        left = {}
        for p in ps:
            if mask[p]:
                lo = p
            left[p] = p - lo
        for p in reversed(ps):
            if mask[p]:
                hi = p
            d[p] = min(left[p], hi - p)
    return d


def two_valued(F, finf):
    pos = F[F > 0]
    mx = int(pos.max()) if len(pos) else 0
    mn = int(pos.min()) if len(pos) else None
    if len(pos) and not np.all((pos == mn) | (pos == mx)):
        return None                                     # a third value: the tile is searched as usual
    d0 = nearest(F == 0)
    t0 = np.where(d0 > 2047, INF, d0 * d0)
    if mn is None:
        return np.minimum(mx, t0)
    d1 = nearest(F == mn)
    t1 = np.where(d1 > 2047, INF, mn + d1 * d1)
    D = np.minimum(mx, np.minimum(t0, t1))
    return np.where(D >= finf, INF, D)


@pytest.mark.parametrize("L", [8, 13, 61, 64, 130, 512, 515])
def test_two_valued_lines_need_no_search(L):
    rng = np.random.default_rng(L)
    finf = 3 * 1023 * 1023 + 1
    for trial in range(120):
        mn = int(rng.integers(1, 400)) ** 2
        mx = finf if rng.random() < 0.25 else mn + int(rng.integers(0, 300)) ** 2      # (mx == mn: one value)
        F = np.full(L, mx, np.int64)
        for _ in range(int(rng.integers(0, 4))):        # stretches of the smaller value (a plate over / under the line)
            a = int(rng.integers(0, L))
            F[a:a + int(rng.integers(1, L + 1))] = mn
        for _ in range(int(rng.integers(0, 4))):        # zero sites: a wall, pillars at chunk edges and elsewhere
            a = int(rng.choice([0, 7, 8, 9, L - 1, int(rng.integers(0, L))])) % L
            F[a:a + int(rng.choice([1, 1, 2, 10, 20]))] = 0
        if trial % 17 == 0:
            F[:] = 0
        got = two_valued(F, finf)
        assert got is not None
        want = brute(np.where(F >= finf, INF, F))
        want = np.where(want >= finf, INF, want)
        assert np.array_equal(got, want), (L, trial, F.tolist())


def test_a_third_value_is_refused():
    F = np.array([0, 4, 4, 9, 9, 16, 4, 0], np.int64)
    assert two_valued(F, 1 << 30) is None
    F[5] = 9
    assert np.array_equal(two_valued(F, 1 << 30), brute(F))
