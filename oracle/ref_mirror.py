"""oracle/ref_mirror.py -- CPU restatement of the reference's spherical sliced-Wasserstein path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this file.  Allowed users:
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

What it restates (all citations relative to /root/reference/Point_Cloud_Resistration/losses/):

* great-circle projection + circle coordinate ......... max_spherical_sliced_w.py:270-279
* shifted / rotated target quantile function ........... max_spherical_sliced_w.py:31-48, 74-92
* one-sided slopes of the circular OT cost in theta .... max_spherical_sliced_w.py:25-65   (dCost)
* circular OT cost at a fixed cut theta ................ max_spherical_sliced_w.py:68-113  (Cost)
* bisection on the cut + tangent-intersection exit ..... max_spherical_sliced_w.py:117-207 (binary_search_circle)
* p == 1 level-median closed form (with its omitted wrap segment) ... :210-247 (emd1D_circle)
* per-pair mean over slices / batched sum over pairs ... :251-286 and max_spherical_sliced_w_fast.py:258-295
* direction sampling (randn + reduced QR) .............. :304-308 and _fast.py:314-318

It keeps the reference's *algorithmic shape* (materialised projections, torch sort / cumsum /
searchsorted / gather, ~log2(n)+1 bisection steps with data-dependent exit) because it is also the
"reference CPU path" that bench.py times on the GPU box's host cores.

Pinning: `oracle/make_golden.py` runs the real reference (imported by file path in the build
container) and this restatement on the same inputs and stores the reference's outputs under
tests/golden/; tests/test_oracle_golden.py checks this file against those vectors.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

TWO_PI = 2.0 * math.pi


# --------------------------------------------------------------------------------------------
# directions and circle coordinates
# --------------------------------------------------------------------------------------------
def draw_directions(num_projections, device="cpu", batch=None, d=3):
    """Orthonormal 2-frames of R^d from the *global* torch generator (ref :307-308, _fast :317-318)."""
    shape = (num_projections, d, 2) if batch is None else (batch, num_projections, d, 2)
    gauss = torch.randn(shape, device=device)
    frames, _ = torch.linalg.qr(gauss)
    return frames


def circle_coords(X, U):
    """X (..., n, 3), U (..., L, 3, 2)  ->  coords (..., L, n) in [0, 1]   (ref :270-279).

    (a, b) = U_l^T x ; unit-normalise with eps 1e-12 ; coord = (atan2(-b, -a) + pi) / (2 pi).
    """
    planar = torch.einsum("...ldk,...nd->...lnk", U, X)
    planar = F.normalize(planar, p=2, dim=-1)
    return (torch.atan2(-planar[..., 1], -planar[..., 0]) + math.pi) / TWO_PI


# --------------------------------------------------------------------------------------------
# circular OT for p != 1 : bisection over the cut theta
# --------------------------------------------------------------------------------------------
def _rotated_target(theta, v_sorted, v_cdf):
    """Quantile function of the target after moving mass theta around the circle (ref :31-48).

    theta (R,1); v_sorted, v_cdf (R,m).  Returns the re-based CDF (R,m), ascending from the first
    non-wrapped atom, and the matching atom positions unrolled onto the real line with one extra
    trailing atom (= first atom + 1), shape (R, m+1).
    """
    turns = torch.floor(theta)
    shifted = v_cdf - (theta - turns)
    wrapped = shifted < 0
    atoms = v_sorted + (turns + wrapped.to(v_sorted.dtype))
    shifted = torch.where(wrapped, shifted + 1, shifted)
    # first atom whose shifted CDF is still >= 0 becomes position 0 (ref: argmin over the
    # non-negative entries + roll_by_gather, :42-47)
    masked = torch.where(wrapped, torch.full_like(shifted, float("inf")), shifted)
    first = torch.argmin(masked, dim=-1, keepdim=True)
    m = v_sorted.shape[-1]
    take = (torch.arange(m, device=v_sorted.device).unsqueeze(0) + first) % m
    shifted = torch.gather(shifted, 1, take)
    atoms = torch.gather(atoms, 1, take)
    atoms = torch.cat([atoms, atoms[:, :1] + 1], dim=1)
    return shifted, atoms


def _pow_abs(x, p):
    if p == 2:
        return x * x
    if p == 1:
        return x.abs()
    return x.abs().pow(p)


def cut_slopes(theta, u_sorted, v_sorted, u_cdf, v_cdf, p):
    """Right / left derivative of the transport cost w.r.t. the cut theta (ref dCost, :25-65)."""
    n = u_sorted.shape[-1]
    cdf_rot, atoms = _rotated_target(theta, v_sorted, v_cdf)
    # source quantile at the target's CDF levels, left-continuous ...
    at = torch.searchsorted(u_cdf, cdf_rot).clamp(0, n - 1)
    src_left = torch.gather(u_sorted, -1, at)
    # ... and right-continuous on the arrays extended by one wrapped atom (:54-57)
    u_cdf_ext = torch.cat([u_cdf, u_cdf[:, :1] + 1], dim=1)
    u_ext = torch.cat([u_sorted, u_sorted[:, :1] + 1], dim=1)
    at_r = torch.searchsorted(u_cdf_ext, cdf_rot, right=True).clamp(0, n)
    src_right = torch.gather(u_ext, -1, at_r)
    nxt, cur = atoms[:, 1:], atoms[:, :-1]
    d_plus = (torch.pow(torch.abs(src_left - nxt), p) - torch.pow(torch.abs(src_left - cur), p)).sum(-1, keepdim=True)
    d_minus = (torch.pow(torch.abs(src_right - nxt), p) - torch.pow(torch.abs(src_right - cur), p)).sum(-1, keepdim=True)
    return d_plus, d_minus


def cut_cost(theta, u_sorted, v_sorted, u_cdf, v_cdf, p):
    """int_0^1 |F_u^-1(t) - (F_v - theta)^-1(t)|^p dt by merging both CDF grids (ref Cost, :68-113)."""
    n = u_sorted.shape[-1]
    m = v_sorted.shape[-1]
    cdf_rot, atoms = _rotated_target(theta, v_sorted, v_cdf)
    grid, _ = torch.sort(torch.cat([u_cdf, cdf_rot], dim=-1), dim=-1)
    widths = torch.diff(grid, dim=-1, prepend=torch.zeros_like(grid[:, :1]))
    src = torch.gather(u_sorted, -1, torch.searchsorted(u_cdf, grid).clamp(0, n - 1))
    atoms = torch.cat([atoms, atoms[:, :1] + 1], dim=1)          # second wrap pad (:103)
    dst = torch.gather(atoms, -1, torch.searchsorted(cdf_rot, grid).clamp(0, m))
    return (widths * _pow_abs(src - dst, p)).sum(-1)


def circular_ot_bisect(u, v, p=2, u_weights=None, v_weights=None, lo=-1.0, hi=1.0,
                       eps=1e-6, slope_bound=10, count_steps=None):
    """Rows of circle coordinates u (R,n), v (R,m)  ->  per-row W_p^p on the circle (ref :117-207).

    The cut theta is detached: only the final `cut_cost` evaluation is differentiable, exactly as
    in the reference (:207).
    """
    R, n = u.shape
    m = v.shape[-1]
    dt, dev = u.dtype, u.device
    wu = torch.full((n,), 1.0 / n, dtype=dt, device=dev) if u_weights is None else u_weights
    wv = torch.full((m,), 1.0 / m, dtype=dt, device=dev) if v_weights is None else v_weights
    u_sorted, iu = torch.sort(u, -1)
    v_sorted, iv = torch.sort(v, -1)
    u_cdf = torch.cumsum(wu[..., iu], -1)
    v_cdf = torch.cumsum(wv[..., iv], -1)
    args = (u_sorted.detach(), v_sorted.detach(), u_cdf, v_cdf, p)

    t_lo = torch.full((R, 1), lo, dtype=dt, device=dev)
    t_hi = torch.full((R, 1), hi, dtype=dt, device=dev)
    t_mid = (t_lo + t_hi) / 2
    steps = 0
    with torch.no_grad():
        while True:
            steps += 1
            dp, dm = cut_slopes(t_mid, *args)
            settled = (dp * dm) <= 0
            if bool(settled.all()):
                break
            tiny = ((t_hi - t_lo) < eps / slope_bound) & ~settled
            if bool(tiny.any()):
                # every unsettled row reaches this width on the same step (all rows halve in
                # lock-step), so this is the exit: intersect the two boundary tangents (:191-200)
                dp_lo, _ = cut_slopes(t_lo, *args)
                _, dm_hi = cut_slopes(t_hi, *args)
                c_lo = cut_cost(t_lo, *args).reshape(-1, 1)
                c_hi = cut_cost(t_hi, *args).reshape(-1, 1)
                usable = tiny & ((dp_lo - dm_hi).abs() > 1e-3)
                crossing = (c_hi - c_lo + t_lo * dp_lo - t_hi * dm_hi) / (dp_lo - dm_hi)
                t_mid = torch.where(usable, crossing, t_mid)
                break
            go_right = dp < 0
            t_lo = torch.where(go_right, t_mid, t_lo)
            t_hi = torch.where(~go_right, t_mid, t_hi)
            t_mid = torch.where(settled, t_mid, (t_lo + t_hi) / 2)
    if count_steps is not None:
        count_steps.append(steps)
    return cut_cost(t_mid, u_sorted, v_sorted, u_cdf, v_cdf, p)


# --------------------------------------------------------------------------------------------
# circular OT for p == 1 : level-median closed form, reference quirk included
# --------------------------------------------------------------------------------------------
def circular_w1_level_median(u, v, u_weights=None, v_weights=None):
    """Rows u (R,n), v (R,m) -> per-row value of the reference's p=1 formula (ref :210-247).

    NOTE: the segment [0, smallest atom) of the circle is *not* integrated and
    the median threshold stays at 0.5 although the integrated weights sum to 1 - smallest atom.
    That is what the reference returns, so it is what this oracle returns.
    """
    R, n = u.shape
    m = v.shape[-1]
    dt, dev = u.dtype, u.device
    wu = torch.full((n,), 1.0 / n, dtype=dt, device=dev) if u_weights is None else u_weights
    wv = torch.full((m,), 1.0 / m, dtype=dt, device=dev) if v_weights is None else v_weights
    u_sorted, iu = torch.sort(u, -1)
    v_sorted, iv = torch.sort(v, -1)
    wu = wu[..., iu].expand(R, n)
    wv = wv[..., iv].expand(R, m)
    merged, order = torch.sort(torch.cat([u_sorted, v_sorted], -1), -1)
    level = torch.cumsum(torch.gather(torch.cat([wu, -wv], -1), -1, order), -1)
    level_sorted, by_level = torch.sort(level, dim=-1)
    gaps = torch.diff(merged, dim=-1, append=torch.ones_like(merged[:, :1]))
    mass = torch.cumsum(torch.gather(gaps, -1, by_level), -1) - 0.5
    mass = torch.where(mass < 0, torch.full_like(mass, float("inf")), mass)
    pick = torch.argmin(mass, dim=-1, keepdim=True)
    median = torch.gather(level_sorted, -1, pick)
    return (gaps * (level - median).abs()).sum(-1)


# --------------------------------------------------------------------------------------------
# the public call shapes
# --------------------------------------------------------------------------------------------
def per_slice_costs(Xs, Xt, Us, p=2, u_weights=None, v_weights=None):
    """One pair: Xs (n,3), Xt (m,3), Us (L,3,2) -> (L,) per-slice circular OT costs."""
    cs = circle_coords(Xs, Us)
    ct = circle_coords(Xt, Us)
    if p == 1:
        return circular_w1_level_median(cs, ct, u_weights, v_weights)
    return circular_ot_bisect(cs, ct, p=p, u_weights=u_weights, v_weights=v_weights)


def sliced_cost(Xs, Xt, Us, p=2, u_weights=None, v_weights=None):
    """Per-pair entry (ref :251-286): mean over slices, 0-dim tensor."""
    return per_slice_costs(Xs, Xt, Us, p, u_weights, v_weights).mean()


def sliced_cost_batched(Xs, Xt, Us, p=2, u_weights=None, v_weights=None):
    """Batched entry (ref _fast.py:258-295): SUM over pairs of the per-pair slice mean, shape [1].

    The reference's batched p == 1 branch raises (it feeds 3-D tensors to a 2-D-only routine);
    here p == 1 is evaluated pair by pair like p != 1, a documented extension.
    """
    total = torch.zeros(1, dtype=Xs.dtype, device=Xs.device)
    for b in range(Xs.shape[0]):
        total = total + per_slice_costs(Xs[b], Xt[b], Us[b], p, u_weights, v_weights).mean()
    return total


def sliced_wasserstein_sphere(Xs, Xt, num_projections, device="cpu", u_weights=None, v_weights=None, p=2):
    """ref :289-310 (per pair, directions from the global generator)."""
    U = draw_directions(num_projections, device=device, d=Xs.shape[1])
    return sliced_cost(Xs, Xt, U, p=p, u_weights=u_weights, v_weights=v_weights)


def sliced_wasserstein_sphere_fast(Xs, Xt, num_projections, device="cpu", u_weights=None, v_weights=None, p=2):
    """ref _fast.py:298-319 (batched)."""
    U = draw_directions(num_projections, device=device, batch=Xs.shape[0], d=Xs.shape[2])
    return sliced_cost_batched(Xs, Xt, U, p=p, u_weights=u_weights, v_weights=v_weights)
