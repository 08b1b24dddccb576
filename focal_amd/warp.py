"""Host side of the TimeWarp / MagWarp view augmentations (reference: data_augmenter/TimeWarpAugmenter.py:18,44,
MagWarpAugmenter.py:18,44, which call tsai 0.3.7's TSTimeWarp / TSMagWarp on the [b, c, i*s] reshape of a modality tensor).

tsai is not available in this build environment and is not vendored by the reference, so the two transforms are restated from
their published algorithm (tsai/data/transforms.py, `random_curve_generator` / `random_cum_curve_generator`):
  curve   = CubicSpline(linspace(-L, 2L - 1, 3 (ord - 1) + 1, dtype=int), N(1, magnitude) knot values)(arange(L))
  MagWarp : x * curve                                   -- one curve per call, shared by batch and channels
  TimeWarp: CubicSpline(arange(L), x)(pos),  pos = clip(cumsum(curve) re-based to [0, 1], 0, 1) * (L - 1)
The reference passes `order=` to the transforms; in tsai that keyword is the transform's pipeline order, not the spline's `ord`,
so the effective knot counts are tsai's defaults -- ord = 4 (MagWarp) and ord = 6 (TimeWarp) -- which is what MOD.yaml's
`order` entries say anyway.  Deviation (documented in DESIGN.md): the device evaluates the signal spline as a 24-tap position-dependent filter -- cardinal
form with a truncated prefilter in the interior, exact not-a-knot basis weights of a 48-sample window at the two ends -- instead of
scipy's banded solve over the whole row; the two agree to ~1e-6 of the signal scale everywhere.
This module only builds the small per-position tables; the arithmetic runs in focal_warp_fwd (csrc/warp.hip)."""
import math

import numpy as np

TAPS, RADIUS = 24, 10  # 2 * RADIUS + 4 taps: 4 B-spline supports, each widened by the prefilter's +-RADIUS
_Z1 = math.sqrt(3.0) - 2.0


def _natural_cubic_through(xk, yk, xq):
    """Values at xq of the not-a-knot cubic spline through (xk, yk) -- scipy's CubicSpline default, which tsai uses for the curve
    and for the signal.  yk: [n] or [n, m] (m splines at once).  Small n only (a dozen knots; 48-sample end windows): a dense
    solve in numpy keeps scipy out of the product path."""
    xk, yk, xq = np.asarray(xk, np.float64), np.asarray(yk, np.float64), np.asarray(xq, np.float64)
    one_d = yk.ndim == 1
    if one_d:
        yk = yk[:, None]
    n = len(xk)
    h = np.diff(xk)
    A = np.zeros((n, n))
    rhs = np.zeros((n, yk.shape[1]))
    slope = np.diff(yk, axis=0) / h[:, None]
    for i in range(1, n - 1):  # continuity of the second derivative, unknowns = first derivatives (scipy's formulation)
        A[i, i - 1], A[i, i], A[i, i + 1] = h[i], 2 * (h[i - 1] + h[i]), h[i - 1]
        rhs[i] = 3 * (h[i] * slope[i - 1] + h[i - 1] * slope[i])
    # not-a-knot at both ends
    d = xk[2] - xk[0]
    A[0, 0], A[0, 1] = h[1], d
    rhs[0] = ((h[0] + 2 * d) * h[1] * slope[0] + h[0] ** 2 * slope[1]) / d
    d = xk[-1] - xk[-3]
    A[-1, -1], A[-1, -2] = h[-2], d
    rhs[-1] = (h[-1] ** 2 * slope[-2] + (2 * d + h[-1]) * h[-2] * slope[-1]) / d
    s = np.linalg.solve(A, rhs)
    idx = np.clip(np.searchsorted(xk, xq, side="right") - 1, 0, n - 2)
    t = (xq - xk[idx])[:, None]
    hh = h[idx][:, None]
    c2 = (3 * slope[idx] - 2 * s[idx] - s[idx + 1]) / hh
    c3 = (s[idx] + s[idx + 1] - 2 * slope[idx]) / hh ** 2
    out = yk[idx] + t * (s[idx] + t * (c2 + t * c3))
    return out[:, 0] if one_d else out


def knot_positions(L, order):
    return np.linspace(-L, 2 * L - 1, 3 * (order - 1) + 1, dtype=int)


def draw_knots(order, magnitude, rng=np.random):
    """The random part of a warp: 3 (order - 1) + 1 Gaussian knot values around 1."""
    return rng.normal(loc=1.0, scale=magnitude, size=3 * (order - 1) + 1)


def random_curve(L, knots, order):
    return _natural_cubic_through(knot_positions(L, order), knots, np.arange(L))


def warp_positions(L, knots, order):
    c = random_curve(L, knots, order).cumsum()
    c -= c[0]
    c /= c[-1]
    return np.clip(c, 0.0, 1.0) * (L - 1)


_END_WIN = 48
_CACHE = {}


def _interior_matrix():
    """[4, TAPS]: row j = the prefilter response sqrt(3) z1^|.| placed where B-spline coefficient k - 1 + j gathers its samples."""
    if "G" not in _CACHE:
        g = math.sqrt(3.0) * _Z1 ** np.abs(np.arange(-RADIUS, RADIUS + 1))
        G = np.zeros((4, TAPS))
        for j in range(4):  # coefficient k - 1 + j gathers samples k - 1 + j - RADIUS .. k - 1 + j + RADIUS  -> taps j .. j + 2 RADIUS
            G[j, j:j + 2 * RADIUS + 1] = g
        _CACHE["G"] = G
    return _CACHE["G"]


def _end_window_spline():
    """Piecewise-cubic coefficients of the 48 not-a-knot basis splines of a 48-sample window (computed once): value at position q
    in segment i = floor(q) is c0[i] + t (c1[i] + t (c2[i] + t c3[i])), t = q - i, each c* a row of 48 basis weights."""
    if "end" not in _CACHE:
        win = _END_WIN
        xk = np.arange(win, dtype=np.float64)
        # evaluate every basis spline and its derivatives' finite form at 4 points per segment -> exact cubic coefficients
        tt = np.array([0.0, 0.25, 0.5, 0.75])
        q = (xk[:-1, None] + tt[None, :]).ravel()
        vals = _natural_cubic_through(xk, np.eye(win), q).reshape(win - 1, 4, win)  # [segment, point, basis]
        V = np.vander(tt, 4, increasing=True)  # [point, power]
        coef = np.einsum("kp,spb->skb", np.linalg.inv(V), vals)  # [segment, power, basis]
        _CACHE["end"] = coef
    return _CACHE["end"]


def end_window_coefficients():
    """fp32 [47 segments][4 powers][48 basis splines]: the table focal_warp_plan_multi evaluates the end windows from on the device."""
    return np.ascontiguousarray(_end_window_spline().astype(np.float32))


def _end_weights(q):
    """[len(q), 48] basis weights of the end-window spline at positions q in [0, 47]."""
    coef = _end_window_spline()
    i = np.clip(np.floor(q).astype(np.int64), 0, _END_WIN - 2)
    t = (q - i)[:, None]
    c = coef[i]  # [n, 4, 48]
    return c[:, 0] + t * (c[:, 1] + t * (c[:, 2] + t * c[:, 3]))


def time_warp_tables(pos):
    """(k0 int32 [L], w float32 [L, TAPS]): y[n] = sum_t w[n, t] x[clamp(k0[n] + t)] evaluates the interpolating cubic spline of x at
    pos[n]: cubic B-spline weights of the 4 coefficients around pos, each coefficient = sqrt(3) sum_j z1^|j| x[. + j]."""
    pos = np.asarray(pos, np.float64)
    k = np.floor(pos).astype(np.int64)
    f = pos - k
    f2 = f * f
    f3 = f2 * f
    bsp = np.stack([(1 - f) ** 3, 3 * f3 - 6 * f2 + 4, -3 * f3 + 3 * f2 + 3 * f + 1, f3], 1) * (1.0 / 6.0)  # coefficients k-1 .. k+2
    # [L, 4] x [4, TAPS] as four fp32 outer products (a BLAS call on this tall-skinny shape takes 20 ms, the numpy float64 form 5 ms,
    # this 1.7 ms at L = 16000 -- the host draws a view's tables between two 6 ms steps)
    b32, G = bsp.astype(np.float32), _interior_matrix().astype(np.float32)
    w = b32[:, 0:1] * G[0]
    for j in range(1, 4):
        w += b32[:, j:j + 1] * G[j]
    k0 = k - 1 - RADIUS
    # The two ends: the infinite (cardinal) form knows nothing about scipy's not-a-knot end condition, whose influence reaches ~12
    # samples inwards (0.27^12 = 1.5e-7).  There the weights are the exact ones: the spline basis of a 48-sample end window
    # (the far side of the window is 36+ samples away from every position it serves), first / last TAPS samples.
    L, win, reach = len(pos), _END_WIN, TAPS // 2
    if L >= win:
        lo = pos < reach
        if lo.any():
            w[lo] = _end_weights(pos[lo])[:, :TAPS]
            k0[lo] = 0
        hi = pos > L - 1 - reach
        if hi.any():
            w[hi] = _end_weights(pos[hi] - (L - win))[:, win - TAPS:]
            k0[hi] = L - TAPS
    return k0.astype(np.int32), w
