"""oracle/exact_shift.py -- second-opinion CPU oracle (numpy, float64-capable).

TEST INFRASTRUCTURE ONLY (same import rule as ref_mirror.py).

For equal sizes n = m and uniform weights the value the reference's bisection
(max_spherical_sliced_w.py:117-207) converges to is the minimum of the convex sequence

    c(k) = (1/n) * sum_i | u_(i) - v_ext(i + k) |^p ,   v_ext(q) = v_(q mod n) + floor(q / n),

over the integer shifts k in [-n, n]  (between two neighbouring kinks
theta = k/n the cost is the linear interpolation of c(k) and c(k+1), so the minimum over theta
sits on a kink).  This file evaluates that definition by brute force -- O(n^2) per slice -- and
also restates the analytic gradient (row A9) and the p = 1 level-median formula (row A7,
reference :210-247) in scalar-level numpy so that the HIP kernels have an oracle that shares no
code with torch's sort / searchsorted.
"""
from __future__ import annotations

import math

import numpy as np

TWO_PI = 2.0 * math.pi


def circle_coords(X, U, dtype=np.float64):
    """X (n,3), U (L,3,2) -> (L,n) circle coordinates (reference :270-279), computed in `dtype`."""
    X = np.asarray(X, dtype=dtype)
    U = np.asarray(U, dtype=dtype)
    a = X @ U[:, :, 0].T          # (n, L)
    b = X @ U[:, :, 1].T
    ang = np.arctan2(-b, -a)      # normalisation is a positive rescale: no effect on the angle
    return ((ang + dtype(math.pi)) / dtype(TWO_PI)).T.astype(dtype)


def shift_costs(u_sorted, v_sorted, p):
    """All c(k), k = -n..n, for one slice (float64)."""
    n = u_sorted.shape[0]
    i = np.arange(n)
    ks = np.arange(-n, n + 1)
    q = i[None, :] + ks[:, None]                      # (2n+1, n)
    v_ext = v_sorted[np.mod(q, n)] + np.floor_divide(q, n)
    diff = np.abs(u_sorted[None, :] - v_ext)
    return ks, (diff ** p).mean(axis=1)


def circular_ot_equal(u, v, p=2):
    """u, v (L,n) coordinates -> (cost (L,), k* (L,)) by exhaustive shift search."""
    u = np.asarray(u, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    L = u.shape[0]
    cost = np.empty(L)
    kbest = np.empty(L, dtype=np.int64)
    for l in range(L):
        ks, c = shift_costs(np.sort(u[l]), np.sort(v[l]), p)
        j = int(np.argmin(c))
        cost[l], kbest[l] = c[j], ks[j]
    return cost, kbest


def w1_level_median(u, v):
    """Reference p = 1 formula (:210-247) for one slice, uniform weights, any n, m (float64).

    Keeps the reference's omission of the segment [0, smallest atom) and its fixed 0.5 threshold;
    if the accumulated gap weight never reaches 0.5 the reference's argmin over an all-inf row
    returns index 0, i.e. the smallest level.
    """
    u = np.sort(np.asarray(u, dtype=np.float64))
    v = np.sort(np.asarray(v, dtype=np.float64))
    n, m = u.shape[0], v.shape[0]
    vals = np.concatenate([u, v])
    sign = np.concatenate([np.full(n, 1.0 / n), np.full(m, -1.0 / m)])
    order = np.argsort(vals, kind="stable")
    vals, sign = vals[order], sign[order]
    level = np.cumsum(sign)
    gaps = np.diff(np.concatenate([vals, [1.0]]))
    by_level = np.argsort(level, kind="stable")
    acc = np.cumsum(gaps[by_level]) - 0.5
    ok = np.nonzero(acc >= 0)[0]
    if ok.size:
        # first index attaining the smallest non-negative value (torch.argmin tie rule)
        pick = ok[np.argmin(acc[ok])]
    else:
        pick = 0
    med = level[by_level][pick]
    return float(np.sum(gaps * np.abs(level - med)))


def ssw_pair(Xs, Xt, U, p=2):
    """mean over slices of the circular OT cost, float64, n = m for p != 1."""
    cu = circle_coords(Xs, U)
    cv = circle_coords(Xt, U)
    if p == 1:
        return float(np.mean([w1_level_median(cu[l], cv[l]) for l in range(cu.shape[0])]))
    cost, _ = circular_ot_equal(cu, cv, p)
    return float(cost.mean())


def ssw_pair_grad(Xs, Xt, U, p=2):
    """Analytic gradient of ssw_pair w.r.t. Xs and Xt, float64, n = m, p != 1.

    d coord / d x = (-b U[:,0] + a U[:,1]) / (2 pi (a^2 + b^2)),  (a, b) = U^T x;
    d c(k*) / d u_(i) = (p/n) |D_i|^(p-1) sgn D_i with D_i = u_(i) - v_ext(i + k*), and the target
    atom paired with u_(i) receives the negative of it.
    """
    Xs = np.asarray(Xs, dtype=np.float64)
    Xt = np.asarray(Xt, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    L, n = U.shape[0], Xs.shape[0]
    gs = np.zeros_like(Xs)
    gt = np.zeros_like(Xt)
    cu = circle_coords(Xs, U)
    cv = circle_coords(Xt, U)
    for l in range(L):
        iu = np.argsort(cu[l], kind="stable")
        iv = np.argsort(cv[l], kind="stable")
        us, vs = cu[l][iu], cv[l][iv]
        ks, c = shift_costs(us, vs, p)
        k = int(ks[int(np.argmin(c))])
        q = np.arange(n) + k
        d = us - (vs[np.mod(q, n)] + np.floor_divide(q, n))
        g = (p / n) * np.abs(d) ** (p - 1) * np.sign(d)
        for X, G, idx, coef in ((Xs, gs, iu, g), (Xt, gt, iv[np.mod(q, n)], -g)):
            a = X[idx] @ U[l, :, 0]
            b = X[idx] @ U[l, :, 1]
            r2 = a * a + b * b
            w = (-b[:, None] * U[l, :, 0][None, :] + a[:, None] * U[l, :, 1][None, :]) / (TWO_PI * r2[:, None])
            np.add.at(G, idx, coef[:, None] * w / L)
    return gs, gt


def chamfer(x, y, batch_reduction="mean"):
    """pytorch3d.loss.chamfer_distance defaults restated: squared-L2 nearest
    neighbour in both directions, mean over points, summed, then batch mean / sum.  The arithmetic
    lives in pytorch3d (un-vendored, un-pinned, not installed): PARITY UNPINNED by the reference."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    d = ((x[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)     # (B,N,M)
    per_pair = d.min(axis=2).mean(axis=1) + d.min(axis=1).mean(axis=1)
    if batch_reduction == "mean":
        return float(per_pair.mean())
    if batch_reduction == "sum":
        return float(per_pair.sum())
    return per_pair
