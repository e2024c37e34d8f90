"""oracle/make_golden.py -- capture golden vectors from the REAL reference (build container only).

Imports /root/reference/Point_Cloud_Resistration/losses/max_spherical_sliced_w.py and
max_spherical_sliced_w_fast.py *by file path* (the `losses` package itself needs POT, which is not
installed -- SURVEY.md 8c), runs them on explicit inputs and writes inputs + the reference's outputs
as small .npz fixtures under tests/golden/.  The reference never travels to the GPU box; these
fixtures (data only: inputs and expected outputs) do.

Run:  MPLBACKEND=Agg python oracle/make_golden.py
Fixture ids follow SURVEY.md 8c (G1..G5).
"""
from __future__ import annotations

import importlib.util
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference/Point_Cloud_Resistration/losses"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _np(t):
    return t.detach().cpu().numpy()


def rot_x(deg):
    c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    return torch.tensor([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=torch.float32)


def pair_with_grads(ref, x, y, U, p):
    xs = x.clone().requires_grad_(True)
    ys = y.clone().requires_grad_(True)
    loss = ref.sliced_cost(xs, ys, U, p=p)
    loss.backward()
    # per-slice costs straight from the reference's circle-level routines
    per = per_slice_from_reference(ref, x, y, U, p)
    return _np(loss), _np(per), _np(xs.grad), _np(ys.grad)


def per_slice_from_reference(ref, X, Y, U, p):
    """Per-slice costs obtained purely by calling the reference: one slice at a time, so that its
    `mean` over a single slice is that slice's circular OT cost."""
    with torch.no_grad():
        return torch.stack([ref.sliced_cost(X, Y, U[l:l + 1], p=p) for l in range(U.shape[0])])


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = _load("ref_ssw", os.path.join(REF, "max_spherical_sliced_w.py"))
    ref_fast = _load("ref_ssw_fast", os.path.join(REF, "max_spherical_sliced_w_fast.py"))
    torch.set_num_threads(8)

    # ---- G1: config 1 (N=256, L=64), x-axis rotations, p in {1,2}, values + grads -------------
    g = torch.Generator().manual_seed(20250101)
    x = F.normalize(torch.randn(256, 3, generator=g), dim=-1)
    U, _ = torch.linalg.qr(torch.randn(64, 3, 2, generator=g))
    out = {"x": _np(x), "U": _np(U)}
    for deg in (90, 135, 180):
        y = x @ rot_x(deg).T
        out[f"y_{deg}"] = _np(y)
        for p in (1, 2):
            loss, per, gx, gy = pair_with_grads(ref, x, y, U, p)
            out[f"loss_{deg}_p{p}"] = loss
            out[f"per_slice_{deg}_p{p}"] = per
            out[f"gx_{deg}_p{p}"] = gx
            out[f"gy_{deg}_p{p}"] = gy
    np.savez_compressed(os.path.join(OUT, "g1_config1.npz"), **out)

    # ---- G2: batched entry, B=2, N=128, L=16, p=2 (and p=3) -----------------------------------
    g = torch.Generator().manual_seed(20250102)
    xb = F.normalize(torch.randn(2, 128, 3, generator=g), dim=-1)
    yb = F.normalize(torch.randn(2, 128, 3, generator=g), dim=-1)
    Ub, _ = torch.linalg.qr(torch.randn(2, 16, 3, 2, generator=g))
    out = {"x": _np(xb), "y": _np(yb), "U": _np(Ub)}
    for p in (2, 3):
        xs = xb.clone().requires_grad_(True)
        ys = yb.clone().requires_grad_(True)
        val = ref_fast.sliced_cost(xs, ys, Ub, p=p)
        val.backward()
        out[f"value_p{p}"] = _np(val)
        out[f"gx_p{p}"] = _np(xs.grad)
        out[f"gy_p{p}"] = _np(ys.grad)
        out[f"per_pair_p{p}"] = np.stack([_np(ref.sliced_cost(xb[b], yb[b], Ub[b], p=p)) for b in range(2)])
    np.savez_compressed(os.path.join(OUT, "g2_batched.npz"), **out)

    # ---- G3: circle level ----------------------------------------------------------------------
    g = torch.Generator().manual_seed(20250103)
    out = {}
    for (n, m) in ((64, 64), (100, 100), (256, 256), (128, 100)):
        u = torch.rand(8, n, generator=g)
        v = torch.rand(8, m, generator=g)
        tag = f"{n}x{m}"
        out[f"u_{tag}"] = _np(u)
        out[f"v_{tag}"] = _np(v)
        for p in (2, 3):
            out[f"bsc_p{p}_{tag}_f32"] = _np(ref.binary_search_circle(u, v, p=p))
            out[f"bsc_p{p}_{tag}_f64"] = _np(ref.binary_search_circle(u.double(), v.double(), p=p))
        out[f"emd1_{tag}_f32"] = _np(ref.emd1D_circle(u, v))
        out[f"emd1_{tag}_f64"] = _np(ref.emd1D_circle(u.double(), v.double()))
    np.savez_compressed(os.path.join(OUT, "g3_circle.npz"), **out)

    # ---- G4: edge cases ------------------------------------------------------------------------
    g = torch.Generator().manual_seed(20250104)
    x = F.normalize(torch.randn(256, 3, generator=g), dim=-1)
    U, _ = torch.linalg.qr(torch.randn(32, 3, 2, generator=g))
    out = {"x": _np(x), "U": _np(U)}
    # identical clouds
    for p in (1, 2):
        out[f"identical_p{p}"] = _np(ref.sliced_cost(x, x.clone(), U, p=p))
    # all-zero target (cf. _fast.py:409): every target coordinate becomes 0
    zeros = torch.zeros(256, 3)
    for p in (1, 2):
        xs = x.clone().requires_grad_(True)
        val = ref.sliced_cost(xs, zeros, U, p=p)
        val.backward()
        out[f"zero_target_p{p}"] = _np(val)
        out[f"zero_target_gx_p{p}"] = _np(xs.grad)
    # un-normalised inputs: points on the surface of the cube [-1,1]^3 (cf. Flow_cube.ipynb:127-159)
    pts = torch.rand(256, 3, generator=g) * 2 - 1
    face = torch.randint(0, 3, (256,), generator=g)
    sign = torch.randint(0, 2, (256,), generator=g).float() * 2 - 1
    pts[torch.arange(256), face] = sign
    tgt = torch.randn(256, 3, generator=g) * 1.7 + 0.3
    out["cube"] = _np(pts)
    out["blob"] = _np(tgt)
    for p in (1, 2):
        loss, per, gx, gy = pair_with_grads(ref, pts, tgt, U, p)
        out[f"cube_loss_p{p}"] = loss
        out[f"cube_per_slice_p{p}"] = per
        out[f"cube_gx_p{p}"] = gx
        out[f"cube_gy_p{p}"] = gy
    # n != m (256 vs 200)
    y200 = F.normalize(torch.randn(200, 3, generator=g), dim=-1)
    out["y200"] = _np(y200)
    for p in (1, 2):
        loss, per, gx, gy = pair_with_grads(ref, x, y200, U, p)
        out[f"n256_m200_loss_p{p}"] = loss
        out[f"n256_m200_per_slice_p{p}"] = per
        out[f"n256_m200_gx_p{p}"] = gx
        out[f"n256_m200_gy_p{p}"] = gy
    # non-uniform weights, n = m = 128
    x128 = F.normalize(torch.randn(128, 3, generator=g), dim=-1)
    y128 = F.normalize(torch.randn(128, 3, generator=g), dim=-1)
    wu = torch.rand(128, generator=g) + 0.1
    wv = torch.rand(128, generator=g) + 0.1
    wu, wv = wu / wu.sum(), wv / wv.sum()
    out.update({"x128": _np(x128), "y128": _np(y128), "wu": _np(wu), "wv": _np(wv)})
    for p in (1, 2):
        out[f"weighted_loss_p{p}"] = _np(ref.sliced_cost(x128, y128, U, p=p, u_weights=wu, v_weights=wv))
    np.savez_compressed(os.path.join(OUT, "g4_edges.npz"), **out)

    # ---- G5: end-to-end RNG parity -------------------------------------------------------------
    g = torch.Generator().manual_seed(20250105)
    x = F.normalize(torch.randn(128, 3, generator=g), dim=-1)
    y = F.normalize(torch.randn(128, 3, generator=g), dim=-1)
    xb = F.normalize(torch.randn(3, 64, 3, generator=g), dim=-1)
    yb = F.normalize(torch.randn(3, 64, 3, generator=g), dim=-1)
    out = {"x": _np(x), "y": _np(y), "xb": _np(xb), "yb": _np(yb), "seed": np.int64(777)}
    torch.manual_seed(777)
    out["value_pair"] = _np(ref.sliced_wasserstein_sphere(x, y, 24, "cpu", p=2))
    torch.manual_seed(777)
    Z = torch.randn((24, 3, 2))
    out["U_pair"] = _np(torch.linalg.qr(Z)[0])
    torch.manual_seed(777)
    out["value_batched"] = _np(ref_fast.sliced_wasserstein_sphere_fast(xb, yb, 12, "cpu", p=2))
    torch.manual_seed(777)
    Z = torch.randn((3, 12, 3, 2))
    out["U_batched"] = _np(torch.linalg.qr(Z)[0])
    np.savez_compressed(os.path.join(OUT, "g5_rng.npz"), **out)

    # ---- G6: a config-2-shaped slice of the headline workload (B=2 of 64, N=1024, L=256) -------
    # and one pair at the headline N=2048 with a reduced slice count, for GPU parity at full N.
    g = torch.Generator().manual_seed(20250106)
    out = {}
    for tag, (B, N, L) in {"c2": (2, 1024, 32), "c3": (1, 2048, 16)}.items():
        xb = F.normalize(torch.randn(B, N, 3, generator=g), dim=-1)
        yb = F.normalize(torch.randn(B, N, 3, generator=g), dim=-1)
        Ub, _ = torch.linalg.qr(torch.randn(B, L, 3, 2, generator=g))
        out[f"x_{tag}"], out[f"y_{tag}"], out[f"U_{tag}"] = _np(xb), _np(yb), _np(Ub)
        for p in (1, 2):
            out[f"per_slice_{tag}_p{p}"] = np.stack(
                [_np(per_slice_from_reference(ref, xb[b], yb[b], Ub[b], p)) for b in range(B)])
        out[f"value_{tag}_p2"] = _np(ref_fast.sliced_cost(xb, yb, Ub, p=2))
    np.savez_compressed(os.path.join(OUT, "g6_headline_shapes.npz"), **out)

    # ---- G7: log-domain Sinkhorn (the reference class, importable: needs only torch) --------------
    sk = _load("ref_sinkhorn", "/root/reference/Comparison_Wasserstein_with_Chamfer_distance/losses/sinkhorn.py")
    g = torch.Generator().manual_seed(20250107)
    xs = F.normalize(torch.randn(2, 96, 3, generator=g), dim=-1)
    ys = xs @ rot_x(30).T + 0.05 * torch.randn(2, 96, 3, generator=g)
    ys = ys[:, :80].contiguous()
    out = {"x": _np(xs), "y": _np(ys)}
    for eps, iters in ((0.05, 60), (0.01, 100)):
        crit = sk.log_Sinkhorn_Distance_Loss(eps=eps, max_iter=iters, batch_reduction="none", type_of_cost_norm="L2")
        cost, P, C = crit(xs, ys, "cpu")
        tag = f"eps{eps}_it{iters}"
        out[f"cost_{tag}"] = _np(cost)
        out[f"P_rowsum_{tag}"] = _np(P.sum(-1))
        out[f"P_colsum_{tag}"] = _np(P.sum(-2))
    crit = sk.log_Sinkhorn_Distance_Loss(eps=0.05, max_iter=60, batch_reduction="sum", type_of_cost_norm="L1")
    out["cost_L1_sum"] = _np(crit(xs, ys, "cpu")[0])
    critN = sk.log_N_Sinkhorn_Distance_Loss(eps=0.05, max_iter=60, batch_reduction="mean", type_of_cost_norm="L2",
                                            type_of_Wasserstein_N="2")
    out["cost_N2_mean"] = _np(critN(xs, ys, "cpu")[0])
    out["C_first_row"] = _np(C[0, 0])
    np.savez_compressed(os.path.join(OUT, "g7_sinkhorn.npz"), **out)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    sys.exit(main())
