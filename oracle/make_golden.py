"""oracle/make_golden.py -- capture golden vectors from the REAL reference (build container only).

Imports /root/reference/Point_Cloud_Resistration/losses/max_spherical_sliced_w.py and
max_spherical_sliced_w_fast.py *by file path* (the `losses` package itself needs POT, which is not
installed -- SURVEY.md 8c), runs them on explicit inputs and writes inputs + the reference's outputs
as small .npz fixtures under tests/golden/.  The reference never travels to the GPU box; these
fixtures (data only: inputs and expected outputs) do.

Run:  MPLBACKEND=Agg python oracle/make_golden.py            (every fixture)
      MPLBACKEND=Agg python oracle/make_golden.py g8 g9      (only the named ones)
Fixture ids follow SURVEY.md 8c (G1..G5); G6/G7 were added in round 1, G8 (the phi-max wrappers) and G9 (the
notebook's Euclidean sliced-W cell, exec'd from the .ipynb JSON) in round 2, G3b (binary_search_circle at its default
p = 1) and G10 (the notebooks' call shape: N = 1200, L = 100, cube-surface clouds, five Adam steps) in round 3.
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


class LinearSphereMap(torch.nn.Module):
    """The deterministic phi of fixture G8: x -> normalize(x W^T + b).  (The reference's own sphere maps are
    normalising flows from a vendored package; the wrappers only ever call `self.phi(x)`.)"""

    def __init__(self, W, b):
        super().__init__()
        self.lin = torch.nn.Linear(3, 3)
        with torch.no_grad():
            self.lin.weight.copy_(torch.as_tensor(W))
            self.lin.bias.copy_(torch.as_tensor(b))

    def forward(self, x):
        return F.normalize(self.lin(x), dim=-1)


def g8_phi_max_wrappers(ref, ref_fast):
    """G8: the REAL max_spherical_wassersten_distance (:498-536) and _fast (_fast.py:346-380) on CPU, with a seeded
    linear sphere map, Adam, and an SSW callable that evaluates the reference's own sliced_cost on FIXED stored
    directions (the wrappers pass `num_projections, device` through to SSW and never look at them)."""
    import contextlib
    import io
    g = torch.Generator().manual_seed(20250108)
    B, N, L, iters, lr = 3, 96, 16, 3, 0.05
    first = F.normalize(torch.randn(B, N, 3, generator=g), dim=-1)
    second = F.normalize(first @ rot_x(40).T + 0.1 * torch.randn(B, N, 3, generator=g), dim=-1)
    W0 = torch.eye(3) + 0.3 * torch.randn(3, 3, generator=g)
    b0 = 0.1 * torch.randn(3, generator=g)
    U_pair, _ = torch.linalg.qr(torch.randn(L, 3, 2, generator=g))
    U_batch, _ = torch.linalg.qr(torch.randn(B, L, 3, 2, generator=g))
    out = {"first": _np(first), "second": _np(second), "W0": _np(W0), "b0": _np(b0), "U_pair": _np(U_pair),
           "U_batch": _np(U_batch), "max_iter": np.int64(iters), "lr": np.float64(lr)}

    def ssw_pair(a, b, num_projections, device, p=2):
        return ref.sliced_cost(a, b, U_pair, p=p)

    def ssw_batch(a, b, num_projections, device, p=2):
        return ref_fast.sliced_cost(a, b, U_batch, p=p)

    for tag, cls, fn in (("pair", ref.max_spherical_wassersten_distance, ssw_pair),
                         ("fast", ref_fast.max_spherical_wassersten_distance_fast, ssw_batch)):
        for mode in ("train", "test"):
            phi = LinearSphereMap(W0, b0)
            opt = torch.optim.Adam(phi.parameters(), lr=lr)
            crit = cls(L, phi, fn, opt, p=2, max_iter=iters, device="cpu")
            a = first.clone().requires_grad_(True)
            b = second.clone().requires_grad_(True)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):            # the wrappers print ssw.item() per inner iteration
                ssw, fa, fb = crit(a, b, train_or_test=mode)
            opt.zero_grad()
            ssw.sum().backward()
            key = f"{tag}_{mode}"
            out[f"{key}_ssw"] = _np(ssw).reshape(-1)
            out[f"{key}_trace"] = np.array([float(t) for t in buf.getvalue().split()], dtype=np.float64)
            out[f"{key}_W"] = _np(phi.lin.weight)
            out[f"{key}_b"] = _np(phi.lin.bias)
            out[f"{key}_phi_first"] = _np(fa)
            out[f"{key}_phi_second"] = _np(fb)
            out[f"{key}_g_first"] = _np(a.grad)
            out[f"{key}_g_second"] = _np(b.grad)
            out[f"{key}_gW"] = _np(phi.lin.weight.grad)
    np.savez_compressed(os.path.join(OUT, "g8_phi_max.npz"), **out)


def g7b_sinkhorn_gradients():
    """G7b: gradients of the REAL log_Sinkhorn_Distance_Loss / log_N_... w.r.t. both clouds (the reference's forward is
    differentiable through its unrolled iterations, sinkhorn.py:35-49).  Same inputs as G7."""
    sk = _load("ref_sinkhorn", "/root/reference/Comparison_Wasserstein_with_Chamfer_distance/losses/sinkhorn.py")
    g = torch.Generator().manual_seed(20250107)
    xs = F.normalize(torch.randn(2, 96, 3, generator=g), dim=-1)
    ys = xs @ rot_x(30).T + 0.05 * torch.randn(2, 96, 3, generator=g)
    ys = ys[:, :80].contiguous()
    out = {"x": _np(xs), "y": _np(ys)}
    cases = {"eps0.05_it60": (sk.log_Sinkhorn_Distance_Loss, dict(eps=0.05, max_iter=60, batch_reduction="sum", type_of_cost_norm="L2")),
             "eps0.01_it100": (sk.log_Sinkhorn_Distance_Loss, dict(eps=0.01, max_iter=100, batch_reduction="sum", type_of_cost_norm="L2")),
             "eps0.1_it5": (sk.log_Sinkhorn_Distance_Loss, dict(eps=0.1, max_iter=5, batch_reduction="mean", type_of_cost_norm="L2")),
             "L1_eps0.05_it30": (sk.log_Sinkhorn_Distance_Loss, dict(eps=0.05, max_iter=30, batch_reduction="sum", type_of_cost_norm="L1")),
             "N2_eps0.05_it30": (sk.log_N_Sinkhorn_Distance_Loss, dict(eps=0.05, max_iter=30, batch_reduction="mean", type_of_cost_norm="L2",
                                                                         type_of_Wasserstein_N="2"))}
    for tag, (cls, kw) in cases.items():
        a, b = xs.clone().requires_grad_(True), ys.clone().requires_grad_(True)
        cost = cls(**kw)(a, b, "cpu")[0]
        cost.backward()
        out[f"cost_{tag}"] = _np(cost)
        out[f"gx_{tag}"] = _np(a.grad)
        out[f"gy_{tag}"] = _np(b.grad)
    np.savez_compressed(os.path.join(OUT, "g7b_sinkhorn_grad.npz"), **out)


def notebook_cell_namespace(path, needle):
    """exec the SOURCE TEXT of the notebook code cell that contains `needle` (definitions only) and return its
    namespace.  The notebook cannot be imported (its `datas` / `losses` modules are not shipped),
    but the cell that holds the Euclidean sliced-W family is self-contained torch code."""
    import itertools
    import json
    nb = json.load(open(path))
    cell = next(c for c in nb["cells"] if c["cell_type"] == "code" and needle in "".join(c["source"]))
    ns = {"torch": torch, "np": np, "optim": torch.optim, "nn": torch.nn, "F": F,
          "combinations": itertools.combinations}
    exec(compile("".join(cell["source"]), path, "exec"), ns)
    return ns


def g9_notebook_euclidean_sw():
    """G9: the notebook's `sliced_wasserstein_distance` / `max_sliced_wasserstein_distance`
    (Wasserstein_flow_problem/Flow_cube.ipynb:275-323), exec'd from the cell text.  The cell reads a GLOBAL
    `num_projections` (set at Flow_cube.ipynb:747 to 100) instead of its `num_projection` argument: the global
    is set to the wanted count before every call.  Directions come from the global CPU generator: the seed is
    stored, and the directions each call drew are re-derived with the cell's own rand_projections."""
    ns = notebook_cell_namespace("/root/reference/Wasserstein_flow_problem/Flow_cube.ipynb", "def rand_projections")
    g = torch.Generator().manual_seed(20250109)
    out = {}
    for tag, n in (("n200", 200), ("n1200", 1200)):          # 1200 = the notebooks' cloud size (:200)
        pts = torch.rand(n, 3, generator=g) * 2 - 1
        face = torch.randint(0, 3, (n,), generator=g)
        pts[torch.arange(n), face] = torch.randint(0, 2, (n,), generator=g).float() * 2 - 1   # cube surface
        tgt = F.normalize(torch.randn(n, 3, generator=g), dim=-1) * 0.8 + 0.1
        out[f"first_{tag}"], out[f"second_{tag}"] = _np(pts), _np(tgt)
        for p in (1, 2, 3):
            for L in (1, 50):
                seed = 9000 + 10 * p + L
                ns["num_projections"] = L
                torch.manual_seed(seed)
                theta = ns["rand_projections"](3, L)
                a = pts.clone().requires_grad_(True)
                torch.manual_seed(seed)
                val = ns["sliced_wasserstein_distance"](a, tgt, num_projection=L, p=p, device="cpu")
                val.backward()
                key = f"{tag}_p{p}_L{L}"
                out[f"swd_{key}"] = _np(val)
                out[f"swd_theta_{key}"] = _np(theta)
                out[f"swd_seed_{key}"] = np.int64(seed)
                out[f"swd_gfirst_{key}"] = _np(a.grad)
        for p in (2,):
            seed = 9500 + p
            torch.manual_seed(seed)
            theta0 = ns["rand_projections"](3, 1)
            torch.manual_seed(seed)
            val = ns["max_sliced_wasserstein_distance"](pts, tgt, p=p, max_iter=10, device="cpu")
            out[f"maxswd_{tag}_p{p}"] = _np(val)
            out[f"maxswd_theta0_{tag}_p{p}"] = _np(theta0)
            out[f"maxswd_seed_{tag}_p{p}"] = np.int64(seed)
    np.savez_compressed(os.path.join(OUT, "g9_notebook_esw.npz"), **out)


def g3b_bisection_at_p1(ref):
    """G3b (round 3): `binary_search_circle` with its DEFAULT p = 1 (:117) -- the bisection ending in Cost's p == 1 branch
    (:107-108), which is NOT emd1D_circle's value (that formula omits the wrap segment).  Same rows as
    G3 plus weighted / unequal-size rows; values in f32 and f64 and the reference's autograd gradients (f32)."""
    base = np.load(os.path.join(OUT, "g3_circle.npz"))
    out = {}
    for tag in ("64x64", "100x100", "256x256", "128x100"):
        u = torch.from_numpy(base[f"u_{tag}"])
        v = torch.from_numpy(base[f"v_{tag}"])
        a, b = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
        val = ref.binary_search_circle(a, b)                  # default p = 1
        val.sum().backward()
        out[f"bsc_p1_{tag}_f32"] = _np(val)
        out[f"bsc_p1_{tag}_gu"] = _np(a.grad)
        out[f"bsc_p1_{tag}_gv"] = _np(b.grad)
        out[f"bsc_p1_{tag}_f64"] = _np(ref.binary_search_circle(u.double(), v.double(), p=1))
    g = torch.Generator().manual_seed(20250110)
    for (n, m) in ((96, 96), (80, 96), (1200, 1200), (1000, 750)):
        u = torch.rand(6, n, generator=g)
        v = torch.rand(6, m, generator=g)
        wu = torch.rand(n, generator=g) + 0.1
        wv = torch.rand(m, generator=g) + 0.1
        wu, wv = wu / wu.sum(), wv / wv.sum()
        tag = f"{n}x{m}"
        out.update({f"u_{tag}": _np(u), f"v_{tag}": _np(v), f"wu_{tag}": _np(wu), f"wv_{tag}": _np(wv)})
        a, b = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
        val = ref.binary_search_circle(a, b, wu, wv, p=1)
        val.sum().backward()
        out[f"bsc_p1_w_{tag}_f32"] = _np(val)
        out[f"bsc_p1_w_{tag}_gu"] = _np(a.grad)
        out[f"bsc_p1_w_{tag}_gv"] = _np(b.grad)
        out[f"bsc_p1_w_{tag}_f64"] = _np(ref.binary_search_circle(u.double(), v.double(), wu.double(), wv.double(), p=1))
        out[f"bsc_p1_{tag}_f32"] = _np(ref.binary_search_circle(u, v, p=1))       # the same rows without weights
        out[f"bsc_p1_{tag}_f64"] = _np(ref.binary_search_circle(u.double(), v.double(), p=1))
        out[f"emd1_w_{tag}_f32"] = _np(ref.emd1D_circle(u, v, wu, wv))
    np.savez_compressed(os.path.join(OUT, "g3b_bisection_p1.npz"), **out)


def cube_surface_points(rng, num_points, side, bias=None):
    """The notebooks' recipe for their clouds (Flow_cube.ipynb:127-200, restated): num_points // 6 rounds over the six
    faces of [0, side]^3, in-face coordinates uniform -- or Beta(bias, 1), the "biased" target -- and the face's own
    coordinate exactly 0 or side.  Un-normalised: the sliced loss projects them as they are."""
    pts = []
    for _ in range(num_points // 6):
        for face in range(6):
            a, b = (rng.uniform(0, side, 2) if bias is None else rng.beta(bias, 1, 2) * side)
            fixed = 0.0 if face % 2 == 0 else side
            pts.append({0: [fixed, a, b], 1: [a, fixed, b], 2: [a, b, fixed]}[face // 2])
    return np.asarray(pts, dtype=np.float32)


def g10_notebook_flow_shape(ref):
    """G10 (round 3): the ONLY live call site of the spherical loss -- Flow_cube.ipynb:1381,
    `sliced_wasserstein_sphere(evolving, target, 100, device, p=2)` with N = 1200 (:200), L = 100 (:747), an un-normalised
    cube-surface evolving cloud against the biased cube-surface target (:7 of the parameter cell), followed by
    loss.backward() and an Adam step (:1382-1383).  Stored: the reference's value, per-slice costs and d loss / d evolving
    on FIXED directions (sliced_cost, p in {1, 2}); the same against a unit-sphere target; and the loss trace of five
    Adam steps of that flow (lr = 0.01 as at :749) with one stored direction set per step."""
    rng = np.random.default_rng(20250111)
    g = torch.Generator().manual_seed(20250111)
    N, L, steps, lr = 1200, 100, 5, 0.01
    source = torch.from_numpy(cube_surface_points(rng, N, 1.0))
    target = torch.from_numpy(cube_surface_points(rng, N, 1.0, bias=15))
    sphere = F.normalize(torch.randn(N, 3, generator=g), dim=-1)
    U = torch.linalg.qr(torch.randn(L, 3, 2, generator=g))[0]
    U_steps = torch.linalg.qr(torch.randn(steps, L, 3, 2, generator=g))[0]
    out = {"source": _np(source), "target": _np(target), "sphere": _np(sphere), "U": _np(U), "U_steps": _np(U_steps),
           "lr": np.float64(lr)}
    for tname, tgt in (("cube", target), ("sphere", sphere)):
        for p in (1, 2):
            loss, per, gx, gy = pair_with_grads(ref, source, tgt, U, p)
            out[f"loss_{tname}_p{p}"] = loss
            out[f"per_slice_{tname}_p{p}"] = per
            out[f"g_evolving_{tname}_p{p}"] = gx
            out[f"g_target_{tname}_p{p}"] = gy
    for p in (1, 2):
        evolving = source.clone().requires_grad_(True)
        opt = torch.optim.Adam([evolving], lr=lr, betas=(0.9, 0.999))
        trace = []
        for i in range(steps):
            opt.zero_grad()
            loss = ref.sliced_cost(evolving, target, U_steps[i], p=p)
            loss.backward(retain_graph=True)
            opt.step()
            trace.append(float(loss))
        out[f"flow_trace_p{p}"] = np.asarray(trace, dtype=np.float64)
        out[f"flow_evolved_p{p}"] = _np(evolving)
    np.savez_compressed(os.path.join(OUT, "g10_notebook_flow.npz"), **out)


def main(only=()):
    os.makedirs(OUT, exist_ok=True)
    ref = _load("ref_ssw", os.path.join(REF, "max_spherical_sliced_w.py"))
    ref_fast = _load("ref_ssw_fast", os.path.join(REF, "max_spherical_sliced_w_fast.py"))
    torch.set_num_threads(8)
    if only:
        if "g8" in only:
            g8_phi_max_wrappers(ref, ref_fast)
        if "g9" in only:
            g9_notebook_euclidean_sw()
        if "g7b" in only:
            g7b_sinkhorn_gradients()
        if "g3b" in only:
            g3b_bisection_at_p1(ref)
        if "g10" in only:
            g10_notebook_flow_shape(ref)
        for f in sorted(os.listdir(OUT)):
            print(f, os.path.getsize(os.path.join(OUT, f)))
        return 0

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

    g8_phi_max_wrappers(ref, ref_fast)
    g9_notebook_euclidean_sw()
    g7b_sinkhorn_gradients()
    g3b_bisection_at_p1(ref)
    g10_notebook_flow_shape(ref)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    sys.exit(main(tuple(sys.argv[1:])))
