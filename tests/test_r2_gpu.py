"""Round-2 GPU parity tests (`-m gpu`, MI355X): what VERDICT round 1 listed as unpinned or untested.

  G8  the phi-max wrappers against the REAL wrappers' outputs (fixture g8_phi_max.npz);
  G9  the Euclidean sliced-W family against the notebook cell's outputs (fixture g9_notebook_esw.npz);
  p = 1 gradients above 2048 points (one-wave search kernel, GRAD = true) against torch autograd of the
      restatement, and the <= 2048 instantiations of that kernel (SHW_P1_SEARCH_KERNEL=1, subprocess);
  full-size BASELINE config 2 (B=64, N=1024, L=256) properties with Chamfer on the same clouds;
  gradient rows of exactly-zero points, evaluation under no_grad, workspace / lease semantics,
  batches beyond 65535 pairs.
Tolerances are stated at each assertion.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shw():
    import shw_amd
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    shw_amd._lib.load()
    return shw_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def unit_cloud(gen, *shape):
    return torch.nn.functional.normalize(torch.randn(*shape, 3, generator=gen), dim=-1)


def directions(gen, *shape):
    return torch.linalg.qr(torch.randn(*shape, 3, 2, generator=gen))[0]


class LinearSphereMap(torch.nn.Module):
    """phi of fixture G8 (oracle/make_golden.py): x -> normalize(x W^T + b)"""

    def __init__(self, W, b):
        super().__init__()
        self.lin = torch.nn.Linear(3, 3)
        with torch.no_grad():
            self.lin.weight.copy_(torch.as_tensor(W))
            self.lin.bias.copy_(torch.as_tensor(b))

    def forward(self, x):
        return torch.nn.functional.normalize(self.lin(x), dim=-1)


# ------------------------------------------------------------------------------------------- G8
@pytest.mark.parametrize("tag", ["pair", "fast"])
@pytest.mark.parametrize("mode", ["train", "test"])
def test_g8_phi_max_wrappers_against_the_real_wrappers(shw, golden, tag, mode):
    """SURVEY 8a row A10.  Same phi, same Adam, SSW = this package's sliced_cost on the fixture's fixed directions.
    Tolerances: per-iteration and returned ssw 1e-4 relative, phi weights 1e-3 absolute
    after max_iter = 3 ascent steps (Adam's first steps are ~lr * sign(g): insensitive to fp32 noise in g)."""
    g = golden("g8_phi_max.npz")
    U = dev(g["U_pair"] if tag == "pair" else g["U_batch"])
    cls = shw.max_spherical_wassersten_distance if tag == "pair" else shw.max_spherical_wassersten_distance_fast
    ssw = lambda a, b, L, device, p=2: shw.sliced_cost(a, b, U, p=p)                 # noqa: E731
    phi = LinearSphereMap(g["W0"], g["b0"]).cuda()
    opt = torch.optim.Adam(phi.parameters(), lr=float(g["lr"]))
    crit = cls(U.shape[-3], phi, ssw, opt, p=2, max_iter=int(g["max_iter"]), device="cuda")
    trace = []
    crit.on_inner_value = trace.append
    a, b = dev(g["first"]).requires_grad_(True), dev(g["second"]).requires_grad_(True)
    val, fa, fb = crit(a, b, train_or_test=mode)
    key = f"{tag}_{mode}"
    assert rel(val.detach().cpu().numpy().reshape(-1), g[f"{key}_ssw"]) < 1e-4
    assert len(trace) == len(g[f"{key}_trace"])
    if trace:
        assert rel(np.array(trace), g[f"{key}_trace"]) < 1e-4
    assert np.abs(phi.lin.weight.detach().cpu().numpy() - g[f"{key}_W"]).max() < 1e-3
    assert np.abs(phi.lin.bias.detach().cpu().numpy() - g[f"{key}_b"]).max() < 1e-3
    assert np.abs(fa.detach().cpu().numpy() - g[f"{key}_phi_first"]).max() < 1e-3
    assert np.abs(fb.detach().cpu().numpy() - g[f"{key}_phi_second"]).max() < 1e-3
    # the trainer back-propagates the returned value into the network (train_W_COS.py:172)
    opt.zero_grad()
    val.sum().backward()
    for got, want in ((a.grad, g[f"{key}_g_first"]), (b.grad, g[f"{key}_g_second"])):
        scale = np.abs(want).max()
        err = np.abs(got.cpu().numpy() - want)
        assert err.max() < 5e-2 * scale and (err > 2e-3 * scale).mean() < 0.02, (err.max(), scale)


def test_g8_builtin_per_pair_function_takes_the_batched_launch_and_equals_the_loop(shw, golden):
    """modules._sum_over_pairs: with the package's own per-pair function in the SSW slot the B direction sets are
    drawn in the loop's order and evaluated by ONE launch; the value must equal the literal loop of :518-519."""
    g = golden("g8_phi_max.npz")
    a, b = dev(g["first"]), dev(g["second"])
    crit = shw.max_spherical_wassersten_distance(16, torch.nn.Identity(), shw.sliced_wasserstein_sphere, None,
                                                 p=2, max_iter=0, device="cuda")
    torch.manual_seed(5)
    fused, _, _ = crit(a, b, train_or_test="test")
    torch.manual_seed(5)
    loop = 0
    for i in range(a.shape[0]):
        loop = loop + shw.sliced_wasserstein_sphere(a[i], b[i], 16, "cuda", p=2)
    assert fused.dim() == 0 and abs(fused.item() - loop.item()) <= 2e-7 * abs(loop.item())


# ------------------------------------------------------------------------------------------- G9
@pytest.mark.parametrize("tag", ["n200", "n1200"])
@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("L", [1, 50])
def test_g9_euclidean_sliced_w_against_the_notebook_cell(shw, golden, tag, p, L):
    """value of Flow_cube.ipynb:280-292 (exec'd cell) for the directions it drew: 1e-5 relative;
    gradient w.r.t. the first cloud 2e-4 of the largest entry (sorted differences of near-equal projections can swap:
    0.5 % of the entries may be off by up to 2 %, as for the spherical loss).  Same seed -> same directions: the
    mirror's own call shape reproduces the notebook's number end to end."""
    g = golden("g9_notebook_esw.npz")
    key = f"{tag}_p{p}_L{L}"
    a = dev(g[f"first_{tag}"]).requires_grad_(True)
    b = dev(g[f"second_{tag}"])
    sums = shw.esw_slice_sums(a[None], b[None], dev(g[f"swd_theta_{key}"]), p)
    val = torch.pow(sums.mean(), 1.0 / p)
    val.backward()
    assert rel(val.item(), g[f"swd_{key}"]) < 1e-5
    want = g[f"swd_gfirst_{key}"]
    err = np.abs(a.grad.cpu().numpy() - want)
    scale = np.abs(want).max()
    assert err.max() < 2e-2 * scale and (err > 2e-4 * scale).mean() <= 0.005, (err.max(), scale)
    torch.manual_seed(int(g[f"swd_seed_{key}"]))
    end_to_end = shw.sliced_wasserstein_distance(a.detach(), b, num_projection=L, p=p, device="cuda")
    assert rel(end_to_end.item(), g[f"swd_{key}"]) < 1e-5


@pytest.mark.parametrize("tag", ["n200", "n1200"])
def test_g9_max_sliced_w_against_the_notebook_cell(shw, golden, tag):
    """Flow_cube.ipynb:294-323: ten Adam ascent steps on one direction from the stored seed, then the distance.
    Adam with betas (0.999, 0.999) moves the direction by ~lr per step whatever the gradient's scale, so ten steps
    stay within 1e-4 relative of the notebook's result."""
    g = golden("g9_notebook_esw.npz")
    torch.manual_seed(int(g[f"maxswd_seed_{tag}_p2"]))
    val = shw.max_sliced_wasserstein_distance(dev(g[f"first_{tag}"]), dev(g[f"second_{tag}"]), p=2, max_iter=10,
                                              device="cuda")
    assert rel(val.item(), g[f"maxswd_{tag}_p2"]) < 1e-4


# ------------------------------------------------------------------------------------------- p = 1 above 2048 points
@pytest.mark.parametrize("n,m", [(3000, 3000), (4096, 4096), (2500, 4000)])
def test_p1_gradients_above_2048_points_against_autograd_of_the_restatement(shw, n, m):
    """the p = 1 training path for n or m > 2048 had no oracle.  Loss 2e-5 relative; gradient
    entries compared as in test_ssw_gpu.grad_close (piecewise-constant gradient: near-ties may swap)."""
    from oracle import ref_mirror
    from helpers.compare import grad_close
    g = torch.Generator().manual_seed(31 * n + m)
    x, y = unit_cloud(g, n), unit_cloud(g, m)
    U = directions(g, 6)
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    val = shw.sliced_cost(xs, ys, U.cuda(), p=1)
    val.backward()
    xc, yc = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    ref = ref_mirror.sliced_cost(xc, yc, U, p=1)
    ref.backward()
    assert abs(val.item() - ref.item()) <= 2e-5 * abs(ref.item()) + 1e-7
    grad_close(xs.grad.cpu().numpy(), xc.grad.numpy(), loose=0.1)
    grad_close(ys.grad.cpu().numpy(), yc.grad.numpy(), loose=0.1)
    assert torch.isfinite(xs.grad).all() and torch.isfinite(ys.grad).all()


_P1_SCRIPT = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import shw_amd
out = {}
for (n, m) in ((64, 64), (700, 1300), (2048, 2048), (1000, 1000), (5, 3)):
    g = torch.Generator().manual_seed(1000 * n + m)
    x = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    y = torch.nn.functional.normalize(torch.randn(m, 3, generator=g), dim=-1)
    U = torch.linalg.qr(torch.randn(8, 3, 2, generator=g))[0]
    xs = x.cuda().requires_grad_(True)
    pair, cost, _ = shw_amd.ssw_pair_losses(xs[None], y.cuda()[None], U.cuda(), p=1, return_slices=True)
    pair.sum().backward()
    _, cost_fwd, _ = shw_amd.ssw_pair_losses(x.cuda()[None], y.cuda()[None], U.cuda(), p=1, return_slices=True)
    out[f"{n}x{m}"] = {"cost": cost[0].tolist(), "cost_fwd": cost_fwd[0].tolist(), "gx": xs.grad.cpu().numpy().tolist()}
print(json.dumps(out))
"""


def test_p1_search_kernel_small_size_instantiations_agree_with_the_merge_kernel(shw):
    """The one-wave search kernel is the production p = 1 path only above 2048 points; its smaller size classes are
    reachable with SHW_P1_SEARCH_KERNEL=1.  Run the same seeded cases in two subprocesses (merge kernel / forced
    search kernel): two different algorithms for the same closed form must agree -- costs to 2e-5 relative (the
    merge kernel clears one mantissa bit per coordinate), gradients up to near-tie swaps."""
    from helpers.compare import grad_close
    res = {}
    for forced in ("0", "1"):
        env = dict(os.environ, SHW_P1_SEARCH_KERNEL=forced)
        r = subprocess.run([sys.executable, "-c", _P1_SCRIPT, ROOT], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[forced] = json.loads(r.stdout.strip().split("\n")[-1])
    for key in res["0"]:
        a, b = res["0"][key], res["1"][key]
        for field in ("cost", "cost_fwd"):
            ca, cb = np.array(a[field]), np.array(b[field])
            assert np.all(np.abs(ca - cb) <= 2e-5 * np.abs(cb) + 2e-7), (key, field)
        grad_close(np.array(a["gx"]), np.array(b["gx"]), loose=0.1)
    # the loss-only call takes the cooperative kernel from 1025 merged atoms on (end of round 2); the two-wave merge
    # kernel it replaced there (SHW_P1_KERNEL=merge) and the cooperative kernel forced into training (=coop) must agree
    for forced in ("merge", "coop"):
        env = dict(os.environ, SHW_P1_SEARCH_KERNEL="0", SHW_P1_KERNEL=forced)
        r = subprocess.run([sys.executable, "-c", _P1_SCRIPT, ROOT], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        other = json.loads(r.stdout.strip().split("\n")[-1])
        for key in res["0"]:
            for field in ("cost", "cost_fwd"):
                ca, cb = np.array(res["0"][key][field]), np.array(other[key][field])
                assert np.all(np.abs(ca - cb) <= 2e-5 * np.abs(cb) + 2e-7), (forced, key, field)
            grad_close(np.array(res["0"][key]["gx"]), np.array(other[key]["gx"]), loose=0.1)


# ------------------------------------------------------------------------- cooperative p = 1 kernels, n > 2048
_P1_COOP_SCRIPT = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import shw_amd
out = {}
for (n, m, kind) in ((3000, 3000, "sphere"), (4096, 4096, "sphere"), (2500, 4000, "sphere"), (8192, 8192, "sphere"),
                     (5000, 5000, "sphere"), (4096, 4096, "lattice"), (2049, 7, "sphere")):
    g = torch.Generator().manual_seed(13 * n + m + len(kind))
    x, y = torch.randn(n, 3, generator=g), torch.randn(m, 3, generator=g)
    if kind == "sphere":
        x, y = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(y, dim=-1)
    else:                                # few lattice sites: duplicate points, exact coordinate ties within and across clouds
        x, y = torch.round(x * 4) / 4 + 0.01, torch.round(y * 4) / 4 + 0.01
    U = torch.linalg.qr(torch.randn(6, 3, 2, generator=g))[0]
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, _ = shw_amd.ssw_pair_losses(xs[None], ys[None], U.cuda(), p=1, return_slices=True)
    pair.sum().backward()
    _, cost_fwd, _ = shw_amd.ssw_pair_losses(x.cuda()[None], y.cuda()[None], U.cuda(), p=1, return_slices=True)
    out[f"{n}x{m}{kind}"] = {"cost": cost[0].tolist(), "cost_fwd": cost_fwd[0].tolist(),
                            "gx": xs.grad.cpu().numpy().tolist(), "gy": ys.grad.cpu().numpy().tolist()}
print(json.dumps(out))
"""


def test_cooperative_p1_kernels_agree_with_the_search_kernel_above_2048_points(shw):
    """Round 2: p = 1 above 2048 points runs shw_ssw_p1_coop.hip (the merge of the two clouds IS one cooperative
    distribution sort of the tagged concatenation; training on 64-bit items up to n + m = 8192).  The one-wave search
    kernel it replaces stays reachable with SHW_P1_SEARCH_KERNEL=1: same seeded cases in two subprocesses -- equal
    sizes, unequal sizes, a lopsided pair, duplicate points -- costs to 2e-5 relative (the merge clears one mantissa
    bit per coordinate), gradients up to near-tie swaps (the rule of the <= 2048 test above)."""
    from helpers.compare import grad_close
    res = {}
    for forced in ("0", "1"):
        env = dict(os.environ, SHW_P1_SEARCH_KERNEL=forced)
        r = subprocess.run([sys.executable, "-c", _P1_COOP_SCRIPT, ROOT], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[forced] = json.loads(r.stdout.strip().split("\n")[-1])
    for key in res["0"]:
        a, b = res["0"][key], res["1"][key]
        for field in ("cost", "cost_fwd"):
            ca, cb = np.array(a[field]), np.array(b[field])
            assert np.all(np.abs(ca - cb) <= 2e-5 * np.abs(cb) + 2e-7), (key, field)
        if "lattice" not in key:         # (ties across the clouds: sub-gradients may differ between the two merges)
            grad_close(np.array(a["gx"]), np.array(b["gx"]), loose=0.1)
            grad_close(np.array(a["gy"]), np.array(b["gy"]), loose=0.1)
        assert np.isfinite(np.array(a["gx"])).all() and np.isfinite(np.array(a["gy"])).all()


# ------------------------------------------------------------------------- cooperative training kernel, n > 2048
_GRADCOOP_SCRIPT = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import shw_amd
out = {}
for (n, kind) in ((3000, "sphere"), (4096, "sphere"), (8192, "sphere"), (5000, "lattice"), (4096, "lattice"), (2600, "lattice"),
                  (6000, "sphere")):
    g = torch.Generator().manual_seed(77 * n + len(kind))
    x, y = torch.randn(2, n, 3, generator=g), torch.randn(2, n, 3, generator=g)
    if kind == "sphere":
        x, y = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(y, dim=-1)
    else:                                # few lattice sites: masses of duplicate points, i.e. exact coordinate ties
        x, y = torch.round(x * 4) / 4 + 0.01, torch.round(y * 4) / 4 + 0.01
    U = torch.linalg.qr(torch.randn(2, 5, 3, 2, generator=g))[0]
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, shift = shw_amd.ssw_pair_losses(xs, ys, U.cuda(), p=2, return_slices=True)
    (pair * torch.tensor([1.0, -0.5], device="cuda")).sum().backward()
    _, cost_fwd, _ = shw_amd.ssw_pair_losses(x.cuda(), y.cuda(), U.cuda(), p=2, return_slices=True)   # the loss-only kernel
    out[f"{n}{kind}"] = {"cost": cost.cpu().tolist(), "shift": shift.cpu().tolist(), "cost_fwd": cost_fwd.cpu().tolist(),
                         "gx": xs.grad.cpu().numpy().tolist(), "gy": ys.grad.cpu().numpy().tolist()}
print(json.dumps(out))
"""


def test_cooperative_training_kernel_agrees_with_the_one_wave_kernels_above_2048_points(shw):
    """Round 2: 2049..8192 points train through shw_ssw_grad_coop.hip (2 / 4 waves per slice, cooperative distribution
    sort on 64-bit items).  The one-wave kernels it replaces (packed words at 4096, 64-bit items on the bitonic network
    at 8192) stay reachable with SHW_GRAD_KERNEL=onewave: the same seeded cases in two subprocesses must give the same
    shifts and per-slice costs to 3e-6 and -- both sorts being stable -- the same gradients also on clouds made of
    duplicate points (lattice: long runs of equal coordinates, which take the cooperative sort's bitonic fallback)."""
    res = {}
    for forced in ("", "onewave"):
        env = dict(os.environ, SHW_GRAD_KERNEL=forced)
        r = subprocess.run([sys.executable, "-c", _GRADCOOP_SCRIPT, ROOT], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[forced] = json.loads(r.stdout.strip().split("\n")[-1])
    for key in res[""]:
        a, b = res[""][key], res["onewave"][key]
        ca, cb = np.array(a["cost"]), np.array(b["cost"])
        assert np.all(np.abs(ca - cb) <= 3e-6 * np.abs(cb) + 1e-12), key
        # round 3: the loss-only cooperative kernel (20 / 24 / 32 keys per lane; on duplicate points its sort falls back to the
        # bitonic network -- in LDS with virtual padding for the odd classes) against the training kernel's costs
        cf = np.array(a["cost_fwd"])
        assert np.all(np.abs(cf - ca) <= 3e-6 * np.abs(ca) + 1e-12), key
        same_shift = np.array(a["shift"]) == np.array(b["shift"])
        assert same_shift.mean() > 0.99, key                      # (exact cost ties may pick either shift)
        for f in ("gx", "gy"):
            ga, gb = np.array(a[f]), np.array(b[f])
            assert np.isfinite(ga).all()
            if same_shift.all():
                assert np.abs(ga - gb).max() <= 1e-6 * np.abs(gb).max() + 1e-12, (key, f)


# ------------------------------------------------------------------------------------------- config 2, full size
def test_full_size_properties_config2_with_chamfer(shw):
    """BASELINE configs[1]: batch=64, N=1024, L=256, sliced-W loss vs Chamfer, fp32 -- at full size, through
    size-independent properties (the oracle needs minutes at this size; a 4-pair sample is checked against it)."""
    from oracle import exact_shift, ref_mirror
    g = torch.Generator().manual_seed(2222)
    B, N, L = 64, 1024, 256
    x, y = unit_cloud(g, B, N), unit_cloud(g, B, N)
    U = directions(g, B, L)
    xd, yd, Ud = x.cuda(), y.cuda(), U.cuda()
    pair, cost, shift = shw.ssw_pair_losses(xd, yd, Ud, p=2, return_slices=True)
    total = shw.sliced_cost(xd, yd, Ud, p=2)
    assert tuple(total.shape) == (1,) and tuple(pair.shape) == (B,) and tuple(cost.shape) == (B, L)
    pc = pair.cpu().numpy().astype(np.float64)
    assert rel(cost.double().mean(1).cpu().numpy(), pc) < 1e-6                 # mean over slices (:286)
    assert abs(total.item() - pc.sum()) < 1e-6 * pc.sum()                      # SUM over pairs (_fast.py:291-293)
    assert int(shift.abs().max()) <= N
    # symmetry, exact zero on identical clouds, rotation invariance (rotate clouds and frames together)
    back = shw.ssw_pair_losses(yd, xd, Ud, p=2)
    assert rel(back.cpu().numpy(), pc) < 1e-5
    assert float(shw.ssw_pair_losses(xd, xd.clone(), Ud, p=2).abs().max()) == 0.0
    Q = torch.linalg.qr(torch.randn(3, 3, generator=g))[0].cuda()
    rot = shw.ssw_pair_losses(xd @ Q.T, yd @ Q.T, torch.einsum("ij,bljk->blik", Q, Ud).contiguous(), p=2)
    assert rel(rot.cpu().numpy(), pc) < 2e-5
    # slice additivity (what sharding the slices over ranks relies on)
    halves = shw.ssw_pair_losses(xd, yd, Ud[:, :128].contiguous(), p=2) * 0.5 \
        + shw.ssw_pair_losses(xd, yd, Ud[:, 128:].contiguous(), p=2) * 0.5
    assert rel(halves.cpu().numpy(), pc) < 1e-6
    # oracle on a sample of pairs and slices: restatement of the reference (fp32) and float64 exhaustive shifts
    for b in (0, 17, 40, 63):
        ref = ref_mirror.per_slice_costs(x[b], y[b], U[b, :8], p=2).numpy()
        assert np.allclose(cost[b, :8].cpu().numpy(), ref, rtol=2e-5, atol=1e-10)
    assert abs(exact_shift.ssw_pair(x[5].numpy(), y[5].numpy(), U[5, :16].numpy(), 2)
               - cost[5, :16].double().mean().item()) < 1e-5 * pc[5]
    # p = 1 on the same clouds (batched p = 1 is the documented pair-wise extension)
    p1 = shw.ssw_pair_losses(xd, yd, Ud, p=1)
    r1 = ref_mirror.per_slice_costs(x[3], y[3], U[3, :8], p=1).numpy()
    _, c1, _ = shw.ssw_pair_losses(xd[3:4], yd[3:4], Ud[3:4, :8].contiguous(), p=1, return_slices=True)
    assert np.allclose(c1[0].cpu().numpy(), r1, rtol=2e-5, atol=1e-10) and torch.isfinite(p1).all()
    # Chamfer on the same clouds: definition (float64), both batch reductions, symmetry
    cd_mean, none = shw.chamfer_distance(xd, yd)
    cd_sum, _ = shw.chamfer_distance(xd, yd, batch_reduction="sum")
    assert none is None
    want = exact_shift.chamfer(x.numpy(), y.numpy(), "mean")
    assert abs(cd_mean.item() - want) < 1e-5 * want and abs(cd_sum.item() - B * want) < 1e-5 * B * want
    assert abs(shw.chamfer_distance(yd, xd)[0].item() - cd_mean.item()) < 1e-6 * want
    # SSW vs Chamfer on a registration-like sweep (the comparison config 2 is about): both vanish at the identity
    # and grow with the rotation angle over [0, 60] degrees
    def rot_x(deg):
        c, s = np.cos(np.radians(deg)), np.sin(np.radians(deg))
        return torch.tensor([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=torch.float32).cuda()
    ssw_curve, cd_curve = [], []
    for deg in (0, 15, 30, 60):
        yr = xd[:8] @ rot_x(deg).T
        ssw_curve.append(shw.ssw_pair_losses(xd[:8], yr, Ud[:8], p=2).mean().item())
        cd_curve.append(shw.chamfer_distance(xd[:8], yr)[0].item())
    assert ssw_curve[0] < 1e-12 and cd_curve[0] < 1e-12
    assert all(a < b for a, b in zip(ssw_curve, ssw_curve[1:])) and all(a < b for a, b in zip(cd_curve, cd_curve[1:]))


# ------------------------------------------------------------------------------------------- ADVICE r1
@pytest.mark.parametrize("p,m", [(2, 256), (1, 256), (2, 200), (3, 256)])
def test_gradient_row_of_an_exactly_zero_point_is_zero_like_the_reference(shw, p, m):
    """A zero (e.g. zero-padded) point projects to (0, 0) on every slice: the reference's autograd gives its row a
    ZERO gradient (atan2 backward masks 0/0, F.normalize clamps the norm); the kernel used to divide by
    a^2 + b^2 = 0 and poison the row with NaN.  Checked against torch autograd of the restatement."""
    from oracle import ref_mirror
    g = torch.Generator().manual_seed(4040 + p + m)
    x, y = unit_cloud(g, 256), unit_cloud(g, m)
    x[7] = 0.0
    x[200] = 0.0
    y[3] = 0.0
    U = directions(g, 16)
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    val = shw.sliced_cost(xs, ys, U.cuda(), p=p)
    val.backward()
    xc, yc = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    ref = ref_mirror.sliced_cost(xc, yc, U, p=p)
    ref.backward()
    assert torch.isfinite(xs.grad).all() and torch.isfinite(ys.grad).all()
    assert float(xs.grad[7].abs().max()) == 0.0 and float(xs.grad[200].abs().max()) == 0.0
    assert float(ys.grad[3].abs().max()) == 0.0
    assert float(xc.grad[7].abs().max()) == 0.0 and float(yc.grad[3].abs().max()) == 0.0     # the reference's value
    assert abs(val.item() - ref.item()) <= 2e-5 * abs(ref.item())


def test_no_grad_evaluation_takes_the_loss_only_path_and_leaves_no_coefficient_scratch(shw):
    """Under torch.no_grad() inputs that carry requires_grad must not trigger the training kernel: no B*L*(n+m)
    coefficient buffers are allocated (0.5 GB at config 3) and the result equals the loss-only path bit for bit."""
    g = torch.Generator().manual_seed(8)
    B, N, L = 16, 2048, 256
    x, y, U = unit_cloud(g, B, N).cuda(), unit_cloud(g, B, N).cuda(), directions(g, B, L).cuda()
    plain = shw.ssw_pair_losses(x, y, U, p=2)
    shw.ssw.SSWWorkspace.clear()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    before = torch.cuda.memory_allocated()
    xr = x.clone().requires_grad_(True)
    with torch.no_grad():
        val = shw.ssw_pair_losses(xr, y, U, p=2)
        tot = shw.sliced_cost(xr, y, U, p=2)
    torch.cuda.synchronize()
    assert not val.requires_grad and not tot.requires_grad
    assert torch.equal(val, plain)
    coef_bytes = 4 * B * L * 2 * N
    assert torch.cuda.max_memory_allocated() - before < coef_bytes // 8
    assert all(not key[-1] for key in shw.ssw.SSWWorkspace._pools)


def test_workspace_leases_results_stay_valid_and_retain_graph_backward_twice(shw):
    """Scratch is pooled, results are not: a loss kept from an earlier call keeps its value; a graph held with
    retain_graph (the phi-max loop's `loss.backward(retain_graph=True)`, :522) keeps its coefficient rows, so a
    second backward gives the same gradient although other evaluations of the same shape ran in between; a loop of
    evaluations of one shape settles on at most two pooled buffer sets."""
    g = torch.Generator().manual_seed(81)
    B, N, L = 4, 512, 64
    x, y, U = unit_cloud(g, B, N).cuda(), unit_cloud(g, B, N).cuda(), directions(g, B, L).cuda()
    x2 = unit_cloud(g, B, N).cuda()
    shw.ssw.SSWWorkspace.clear()
    xs = x.clone().requires_grad_(True)
    first = shw.sliced_cost(xs, y, U, p=2)
    keep = first.detach().clone()
    first.backward(retain_graph=True)
    g1 = xs.grad.clone()
    for _ in range(5):                                   # same shape, other data: must not touch `first`'s buffers
        other = shw.sliced_cost(x2.clone().requires_grad_(True), y, U, p=2)
        other.backward()
    assert torch.equal(first.detach(), keep)
    xs.grad = None
    first.backward()
    assert torch.equal(xs.grad, g1)
    del first, other
    pools = shw.ssw.SSWWorkspace._pools
    assert sum(len(v) for v in pools.values()) <= 2 * len(pools)
    n_coef_sets = sum(len(v) for k, v in pools.items() if k[-1])
    assert 1 <= n_coef_sets <= 2


def test_more_than_65535_pairs_forward_and_backward(shw):
    """The point-gradient kernel maps pairs to gridDim.y; batches beyond 65535 pairs go out in pair blocks (they used
    to fail in backward only)."""
    g = torch.Generator().manual_seed(82)
    B, N, L = 70000, 8, 2
    x, y, U = unit_cloud(g, B, N).cuda(), unit_cloud(g, B, N).cuda(), directions(g, L).cuda()
    xs = x.clone().requires_grad_(True)
    pair = shw.ssw_pair_losses(xs, y, U, p=2)
    w = torch.rand(B, generator=g).cuda()
    (pair * w).sum().backward()
    tail = slice(69990, 70000)
    xt = x[tail].clone().requires_grad_(True)
    ref = shw.ssw_pair_losses(xt, y[tail], U, p=2)
    (ref * w[tail]).sum().backward()
    assert torch.equal(pair[tail].detach(), ref.detach())
    assert torch.allclose(xs.grad[tail], xt.grad, rtol=1e-6, atol=1e-12)


# ------------------------------------------------------------------------------------------- config 4 (8 x MI355X)
def test_config4_decomposition_on_one_gpu(shw):
    """BASELINE configs[3]: batch=512, N=2048, L=1024, slices sharded across 8 GPUs.  No 8-GPU node is available to
    the tests, so the decomposition is rehearsed on one GPU with the HIP evaluator: the 8 per-rank shares
    (512 pairs x 128 directions each, partials scaled by 1/L_global as bench.py and dist.py do) must add up to the
    single-launch evaluation of the whole problem -- per pair to 1e-6 relative (fp32 summation order only), and
    likewise the 8 "pairs" shares (64 pairs x 1024 directions) bit for bit."""
    lib = shw._lib.load()
    g = torch.Generator().manual_seed(44)
    B, N, L, world = 512, 2048, 1024, 8
    x, y = unit_cloud(g, B, N).cuda(), unit_cloud(g, B, N).cuda()
    U = shw.stiefel_frames(torch.randn(B, L, 3, 2, generator=g).cuda())
    full = shw.ssw_pair_losses(x, y, U, p=2)
    st = torch.cuda.current_stream().cuda_stream
    acc = torch.zeros(B, dtype=torch.float64, device="cuda")
    for r in range(world):
        lo, hi = shw.dist.shard_bounds(L, world, r)
        Ul = U[:, lo:hi].contiguous()
        cost = torch.empty(B * (hi - lo), device="cuda")
        part = torch.empty(B + 2, device="cuda")
        shw._lib.check(lib.shw_ssw_forward(x.data_ptr(), y.data_ptr(), Ul.data_ptr(), B, N, N, hi - lo, (hi - lo) * 6,
                                           2.0, cost.data_ptr(), None, st), "fwd")
        shw._lib.check(lib.shw_ssw_reduce(cost.data_ptr(), B, hi - lo, 1.0 / L, part.data_ptr(),
                                          part.data_ptr() + 4 * B, st), "reduce")
        acc += part[:B].double()
    assert rel(acc.cpu().numpy(), full.double().cpu().numpy()) < 1e-6
    by_pairs = torch.cat([shw.ssw_pair_losses(x[lo:hi], y[lo:hi], U[lo:hi], p=2)
                          for lo, hi in (shw.dist.shard_bounds(B, world, r) for r in range(world))])
    assert torch.equal(by_pairs, full)


# ------------------------------------------------------------------------------------------- phi-max fusion (8f rank 2)
@pytest.mark.parametrize("which", ["ssw_fast", "live_criterion"])
def test_graphed_phi_max_loop_equals_the_eager_loop_bit_for_bit(shw, golden, which):
    """GraphedAscent: one inner iteration (phi forward, loss forward + gradient kernels, phi backward, Adam step)
    captured into a hipGraph and replayed.  Same kernels on the same values as the eager loop: after two trainer-level
    calls (the first warms up, captures and replays; the second only replays) the phi weights, the returned value
    and the gradient that flows back to the clouds must be IDENTICAL to the eager wrapper's."""
    g = golden("g8_phi_max.npz")
    U = dev(g["U_batch"])
    first, second = dev(g["first"]), dev(g["second"])
    results = []
    for graph in (False, True):
        phi = LinearSphereMap(g["W0"], g["b0"]).cuda()
        opt = torch.optim.Adam(phi.parameters(), lr=0.05, capturable=True)
        if which == "ssw_fast":
            ssw = lambda a, b, L, device, p=2: shw.sliced_cost(a, b, U, p=p)                 # noqa: E731
            crit = shw.max_spherical_wassersten_distance_fast(16, phi, ssw, opt, p=2, max_iter=5, device="cuda",
                                                              graph=graph)
        else:
            csw = lambda a, b: shw.ssw_pair_losses(a, b, U, 2).sqrt().mean()                 # noqa: E731
            crit = shw.max_cos_disimilarity_wassersten_distance(phi, csw, "cuda", opt, max_iter=5, lam=0.1, graph=graph)
        out = []
        for call in range(2):
            a = first.clone().requires_grad_(True)
            val, fa, _ = crit(a, second, train_or_test="train")
            val.sum().backward()
            out.append((val.detach().clone(), phi.lin.weight.detach().clone(), phi.lin.bias.detach().clone(),
                        fa.detach().clone(), a.grad.clone()))
            opt.zero_grad(set_to_none=True)
        results.append(out)
    for call in range(2):
        for x, y in zip(results[0][call], results[1][call]):
            assert torch.equal(x, y), (which, call)


def test_graphed_ascent_refuses_an_optimizer_that_cannot_be_captured(shw, golden):
    g = golden("g8_phi_max.npz")
    phi = LinearSphereMap(g["W0"], g["b0"]).cuda()
    opt = torch.optim.Adam(phi.parameters(), lr=0.05)                   # capturable=False
    crit = shw.max_spherical_wassersten_distance_fast(16, phi, shw.sliced_wasserstein_sphere_fast, opt, max_iter=2,
                                                      device="cuda", graph=True)
    with pytest.raises(RuntimeError, match="capturable"):
        crit(dev(g["first"]), dev(g["second"]), train_or_test="train")


# ------------------------------------------------------------------------------------------- circle level (G3, direct)
@pytest.mark.parametrize("tag,tol", [("64x64", 1e-5), ("100x100", 2e-5), ("256x256", 1e-5), ("128x100", 2e-5)])
@pytest.mark.parametrize("p", [2, 3])
def test_g3_binary_search_circle_on_coordinate_rows(shw, golden, tag, tol, p):
    """VERDICT r1 "missing" 5: `binary_search_circle(u, v, p)` is callable on coordinate rows in the reference
    (max_spherical_sliced_w.py:117); the C ABI had no coordinates-in entry and the G3 fixtures were only reachable
    through a planar embedding at 5x the tolerance.  shw_circle_ot reads the rows directly: per-row costs against the
    reference's float64 evaluation of the same fp32 inputs at the per-slice tolerance (1e-5; 2e-5 where the reference
    itself leaves the bisection through its tangent step)."""
    g = golden("g3_circle.npz")
    u, v = dev(g[f"u_{tag}"]), dev(g[f"v_{tag}"])
    cost = shw.binary_search_circle(u, v, p=p)
    assert tuple(cost.shape) == (8,)
    assert rel(cost.cpu().numpy(), g[f"bsc_p{p}_{tag}_f64"]) < tol
    assert rel(cost.cpu().numpy(), g[f"bsc_p{p}_{tag}_f32"]) < 2 * tol


@pytest.mark.parametrize("tag", ["64x64", "100x100", "256x256", "128x100"])
def test_g3_emd1d_circle_on_coordinate_rows(shw, golden, tag):
    g = golden("g3_circle.npz")
    u, v = dev(g[f"u_{tag}"]), dev(g[f"v_{tag}"])
    cost = shw.emd1D_circle(u, v)
    assert rel(cost.cpu().numpy(), g[f"emd1_{tag}_f64"]) < 1e-5
    one = shw.emd1D_circle(u[3], v[3])                       # a single row, like the reference's 1-D call
    assert tuple(one.shape) == (1,) and abs(one.item() - cost[3].item()) <= 1e-7 * abs(cost[3].item())


@pytest.mark.parametrize("n,m,p,weighted", [(256, 256, 2, False), (100, 100, 3, False), (1000, 1000, 2, False),
                                            (3000, 3000, 2, False), (128, 100, 2, False), (64, 64, 2, True),
                                            (200, 200, 1, False), (130, 77, 1, False), (700, 700, 1, False),
                                            (3000, 2000, 1, False)])
def test_circle_level_values_and_gradients_against_the_cpu_oracle(shw, n, m, p, weighted):
    """Coordinate rows through every kernel family (equal sizes, > 2048 atoms, unequal sizes, weights, p = 1):
    values and d cost / d coordinate against torch autograd of the CPU restatement."""
    from helpers.compare import grad_close
    from oracle import ref_mirror
    g = torch.Generator().manual_seed(17 * n + m + p)
    rows = 3
    u, v = torch.rand(rows, n, generator=g), torch.rand(rows, m, generator=g)
    wu = wv = None
    if weighted:
        wu, wv = torch.rand(n, generator=g) + 0.1, torch.rand(m, generator=g) + 0.1
        wu, wv = wu / wu.sum(), wv / wv.sum()
    ud, vd = u.cuda().requires_grad_(True), v.cuda().requires_grad_(True)
    # p = 1 here means the level-median formula = emd1D_circle (round 3: binary_search_circle(p=1) is the BISECTION, as in
    # the reference; its tests are in test_r3_gpu.py)
    entry = shw.emd1D_circle if p == 1 else shw.binary_search_circle
    cost = entry(ud, vd, u_weights=None if wu is None else wu.cuda(),
                 v_weights=None if wv is None else wv.cuda(), p=p)
    w = torch.tensor([1.0, -0.5, 2.0])
    (cost * w.cuda()).sum().backward()
    uc, vc = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
    if p == 1:
        ref = ref_mirror.circular_w1_level_median(uc, vc, wu, wv)
    else:
        ref = ref_mirror.circular_ot_bisect(uc, vc, p=p, u_weights=wu, v_weights=wv)
    (ref * w).sum().backward()
    assert rel(cost.detach().cpu().numpy(), ref.detach().numpy()) < 2e-5
    with torch.no_grad():                # the loss-only kernels read coordinate rows too (p = 1 from 1025 merged atoms on:
        plain = entry(u.cuda(), v.cuda(), u_weights=None if wu is None else wu.cuda(),   # cooperative)
                      v_weights=None if wv is None else wv.cuda(), p=p)
    assert rel(plain.cpu().numpy(), ref.detach().numpy()) < 2e-5
    loose = 0.2 if p == 1 else 2e-2
    # torch.rand coordinates sit on a 2^-24 grid: near-ties (and exact ties) are far more frequent than among projected
    # points, and every one moves two gradient entries by ~gap/n (measured: 28 of 9000 entries at 2.6e-4 of the
    # largest for n = 3000) -- the share of entries allowed outside the strict bound is 0.5 % here
    grad_close(ud.grad.cpu().numpy(), uc.grad.numpy(), loose=loose, frac=0.005,
               exact=(max(n, m) <= 130 and not weighted and p != 1))
    grad_close(vd.grad.cpu().numpy(), vc.grad.numpy(), loose=loose, frac=0.005,
               exact=(max(n, m) <= 130 and not weighted and p != 1))


# ------------------------------------------------------------------------------------------- Sinkhorn backward (G7b)
@pytest.mark.parametrize("tag,kind,kw", [
    ("eps0.05_it60", "plain", dict(eps=0.05, max_iter=60, batch_reduction="sum", type_of_cost_norm="L2")),
    ("eps0.01_it100", "plain", dict(eps=0.01, max_iter=100, batch_reduction="sum", type_of_cost_norm="L2")),
    ("eps0.1_it5", "plain", dict(eps=0.1, max_iter=5, batch_reduction="mean", type_of_cost_norm="L2")),
    ("L1_eps0.05_it30", "plain", dict(eps=0.05, max_iter=30, batch_reduction="sum", type_of_cost_norm="L1")),
    ("N2_eps0.05_it30", "N", dict(eps=0.05, max_iter=30, batch_reduction="mean", type_of_cost_norm="L2",
                                  type_of_Wasserstein_N="2"))])
def test_g7b_sinkhorn_gradients_against_the_reference_autograd(shw, golden, tag, kind, kw):
    """VERDICT r1 "missing" 4: the reference's Sinkhorn forward is differentiable through its unrolled iterations
    (sinkhorn.py:35-49).  Fixture G7b = gradients of the REAL classes w.r.t. both clouds (CPU autograd).  The HIP
    backward walks the stored trajectory of the duals.  Tolerances: value as in G7 (5e-4: 1/eps amplifies fp32 rounding
    of the duals); gradients 2e-3 of the largest entry (they carry the same amplification through ~max_iter steps;
    the reference's own fp32-vs-fp64 gap on these gradients is ~1e-3 at eps = 0.01)."""
    g = golden("g7b_sinkhorn_grad.npz")
    x, y = dev(g["x"]).requires_grad_(True), dev(g["y"]).requires_grad_(True)
    cls = shw.log_Sinkhorn_Distance_Loss if kind == "plain" else shw.log_N_Sinkhorn_Distance_Loss
    cost, P, C = cls(**kw)(x, y, "cuda")
    cost.backward()
    assert rel(cost.item(), g[f"cost_{tag}"]) < 5e-4
    assert P is not None and tuple(P.shape) == (2, 96, 80)
    for got, want in ((x.grad, g[f"gx_{tag}"]), (y.grad, g[f"gy_{tag}"])):
        scale = np.abs(want).max()
        assert np.abs(got.cpu().numpy() - want).max() < 2e-3 * scale, (np.abs(got.cpu().numpy() - want).max(), scale)


@pytest.mark.parametrize("n,m", [(1, 1), (5, 700), (300, 257), (1024, 1024)])
def test_sinkhorn_gradients_against_autograd_of_the_restatement(shw, n, m):
    """Other sizes (ragged tiles, single points, config-2 size): float64 autograd of oracle/sinkhorn_mirror.py."""
    from oracle import sinkhorn_mirror
    g = torch.Generator().manual_seed(5 * n + m)
    B = 2
    x, y = unit_cloud(g, B, n), unit_cloud(g, B, m) * 0.9 + 0.05
    xd, yd = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    cost, _, _ = shw.sinkhorn_pair_costs(xd, yd, 0.05, 25)
    w = torch.tensor([1.0, -0.7])
    (cost * w.cuda()).sum().backward()
    xc, yc = x.double().requires_grad_(True), y.double().requires_grad_(True)
    ref = sinkhorn_mirror.sinkhorn_costs(xc, yc, 0.05, 25)[0]
    (ref * w.double()).sum().backward()
    assert rel(cost.detach().cpu().numpy(), ref.detach().numpy()) < 5e-4
    for got, want in ((xd.grad, xc.grad), (yd.grad, yc.grad)):
        scale = want.abs().max().item()
        assert (got.cpu().double() - want).abs().max().item() < 2e-3 * scale + 1e-9
    with torch.no_grad():                                   # no_grad: the value-only path, identical value
        plain, _, _ = shw.sinkhorn_pair_costs(xd, yd, 0.05, 25)
    assert torch.allclose(plain, cost.detach(), rtol=1e-6)


# ------------------------------------------------------------------------------- weighted clouds above 2048 points
@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(3000, 2500), (4096, 4096)])
@pytest.mark.parametrize("p", [2, 1])
def test_weighted_clouds_above_2048_points_against_the_cpu_oracle(shw, n, m, p):
    """The weighted kernels of round 2 (walking ranks with window rows, level-median in registers) in their 64-atoms-per-
    lane class (66 / 102 KB of LDS per slice): per-slice costs against the float64 restatement, finite gradients."""
    from oracle import ref_mirror
    g = torch.Generator().manual_seed(3 + n + p)
    x, y, U = unit_cloud(g, n), unit_cloud(g, m), directions(g, 3)
    wu, wv = torch.rand(n, generator=g) + 0.1, torch.rand(m, generator=g) + 0.1
    wu, wv = wu / wu.sum(), wv / wv.sum()
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, _ = shw.ssw_pair_losses(xs[None], ys[None], U.cuda(), p=p, return_slices=True, u_weights=wu.cuda(),
                                        v_weights=wv.cuda())
    pair.sum().backward()
    ref = ref_mirror.per_slice_costs(x.double(), y.double(), U.double(), p=p, u_weights=wu.double(), v_weights=wv.double())
    assert np.allclose(cost[0].detach().cpu().numpy(), ref.numpy(), rtol=5e-5, atol=1e-9)
    assert torch.isfinite(xs.grad).all() and torch.isfinite(ys.grad).all()
