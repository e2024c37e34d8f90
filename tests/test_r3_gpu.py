"""Round-3 GPU parity tests (`-m gpu`, MI355X): what VERDICT round 2 listed as unpinned, untested or non-deterministic.

  G3b  `binary_search_circle` at its DEFAULT p = 1 (the bisection ending in Cost's p == 1 branch, max_spherical_sliced_w.py
       :107-108, :117) against the real reference's outputs -- values and autograd gradients, equal / unequal sizes, weights;
  G10  the notebooks' call shape (Flow_cube.ipynb:1381: N = 1200, L = 100, cube-surface evolving cloud, p in {1, 2},
       value + per-slice costs + d/d evolving) against the real `sliced_cost`;
  run-to-run bit-equality of every gradient the package produces (general path and Chamfer included);
  the size classes between the powers of two (1200, 1280, 1500, 2000 ...) for loss and gradients.
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


# ------------------------------------------------------------------------------------------- G3b: bisection at p = 1
G3B_PLAIN = ["64x64", "100x100", "256x256", "128x100", "96x96", "80x96", "1200x1200", "1000x750"]
G3B_WITH_GRADS = ["64x64", "100x100", "256x256", "128x100"]
G3B_WEIGHTED = ["96x96", "80x96", "1200x1200", "1000x750"]


def g3b_rows(golden, tag):
    g = golden("g3b_bisection_p1.npz")
    src = g if f"u_{tag}" in g.files else golden("g3_circle.npz")
    return g, dev(src[f"u_{tag}"]), dev(src[f"v_{tag}"])


def sign_gradients_close(got, ref, max_wrong):
    """Gradients of a p = 1 cost are sums of +-(CDF widths): piecewise CONSTANT in the coordinates.  An entry is either
    right to rounding or off by a whole width; count the entries off by more than 2 % of the largest."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    wrong = int((np.abs(got - ref) > 2e-2 * np.abs(ref).max()).sum())
    assert wrong <= max_wrong, (wrong, got.size)
    return wrong


@pytest.mark.parametrize("tag", G3B_PLAIN)
def test_g3b_binary_search_circle_default_p_is_the_bisection(shw, golden, tag):
    """VERDICT r2 missing 1 / ADVICE r2 (medium).  `binary_search_circle(u, v)` -- p defaults to 1 as at :117 -- against the
    REAL reference's output on the same rows: 2e-5 per row against its float64 evaluation, 4e-5 against its fp32 one
    (the reference's own fp32-vs-fp64 gap on these rows is <= 4.4e-7)."""
    g, u, v = g3b_rows(golden, tag)
    cost = shw.binary_search_circle(u, v)
    assert tuple(cost.shape) == (u.shape[0],)
    assert rel(cost.cpu().numpy(), g[f"bsc_p1_{tag}_f64"]) < 2e-5
    assert rel(cost.cpu().numpy(), g[f"bsc_p1_{tag}_f32"]) < 4e-5
    # and it is NOT the level-median value (up to 2.5 % away): emd1D_circle stays on that formula
    median = shw.emd1D_circle(u, v)
    assert rel(median.cpu().numpy(), g[f"bsc_p1_{tag}_f32"]) > 1e-4


@pytest.mark.parametrize("tag", G3B_WITH_GRADS)
def test_g3b_bisection_at_p1_gradients_against_reference_autograd(shw, golden, tag):
    g, u, v = g3b_rows(golden, tag)
    a, b = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
    shw.binary_search_circle(a, b).sum().backward()
    sign_gradients_close(a.grad.cpu().numpy(), g[f"bsc_p1_{tag}_gu"], max_wrong=2)
    sign_gradients_close(b.grad.cpu().numpy(), g[f"bsc_p1_{tag}_gv"], max_wrong=2)


@pytest.mark.parametrize("tag", G3B_WEIGHTED)
def test_g3b_weighted_bisection_at_p1(shw, golden, tag):
    """Weights and unequal sizes at p = 1 through the general kernel (PMODE 0 on |d|^1), values 2e-5 per row against the
    reference's float64 run, gradients against its fp32 autograd."""
    g, u, v = g3b_rows(golden, tag)
    wu, wv = dev(g[f"wu_{tag}"]), dev(g[f"wv_{tag}"])
    a, b = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
    cost = shw.binary_search_circle(a, b, wu, wv, p=1)
    cost.sum().backward()
    assert rel(cost.detach().cpu().numpy(), g[f"bsc_p1_w_{tag}_f64"]) < 2e-5
    assert rel(cost.detach().cpu().numpy(), g[f"bsc_p1_w_{tag}_f32"]) < 4e-5
    n = u.shape[1]
    sign_gradients_close(a.grad.cpu().numpy(), g[f"bsc_p1_w_{tag}_gu"], max_wrong=max(4, n // 100))
    sign_gradients_close(b.grad.cpu().numpy(), g[f"bsc_p1_w_{tag}_gv"], max_wrong=max(4, n // 100))
    with torch.no_grad():
        assert rel(shw.binary_search_circle(u, v, wu, wv, p=1).cpu().numpy(), g[f"bsc_p1_w_{tag}_f64"]) < 2e-5


@pytest.mark.parametrize("n,m,weighted", [(2048, 2048, False), (3000, 3000, False), (5000, 5000, False), (8192, 8192, False),
                                          (700, 512, False), (2048, 1536, False), (2048, 2048, True), (1500, 900, True)])
def test_bisection_at_p1_against_the_cpu_restatement(shw, n, m, weighted):
    """Every kernel family that can serve binary_search_circle(p=1): equal sizes up to 8192 (one-wave, two-wave and
    cooperative shift kernels on |.|^1), unequal sizes and weights (general kernel), against torch autograd of
    oracle/ref_mirror.circular_ot_bisect (held to fixture G3b on the CPU)."""
    from oracle import ref_mirror
    gen = torch.Generator().manual_seed(5 * n + m + int(weighted))
    rows = 3
    u, v = torch.rand(rows, n, generator=gen), torch.rand(rows, m, generator=gen)
    wu = wv = None
    if weighted:
        wu, wv = torch.rand(n, generator=gen) + 0.1, torch.rand(m, generator=gen) + 0.1
        wu, wv = wu / wu.sum(), wv / wv.sum()
    cu = None if wu is None else wu.cuda()
    cv = None if wv is None else wv.cuda()
    a, b = u.cuda().requires_grad_(True), v.cuda().requires_grad_(True)
    cost = shw.binary_search_circle(a, b, cu, cv)
    cost.sum().backward()
    ra, rb = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
    ref = ref_mirror.circular_ot_bisect(ra, rb, p=1, u_weights=wu, v_weights=wv)
    ref.sum().backward()
    ref64 = ref_mirror.circular_ot_bisect(u.double(), v.double(), p=1, u_weights=None if wu is None else wu.double(),
                                          v_weights=None if wv is None else wv.double())
    assert rel(cost.detach().cpu().numpy(), ref64.numpy()) < 2e-5
    with torch.no_grad():
        assert rel(shw.binary_search_circle(u.cuda(), v.cuda(), cu, cv).cpu().numpy(), ref64.numpy()) < 2e-5
    sign_gradients_close(a.grad.cpu().numpy(), ra.grad.numpy(), max_wrong=max(4, (n + m) // 100))
    sign_gradients_close(b.grad.cpu().numpy(), rb.grad.numpy(), max_wrong=max(4, (n + m) // 100))


def test_emd1d_circle_refuses_other_powers(shw):
    u = torch.rand(2, 16, device="cuda")
    with pytest.raises(ValueError, match="level-median"):
        shw.emd1D_circle(u, u, p=2)


# ------------------------------------------------------------------------------------------- G10: the notebooks' shape
@pytest.mark.parametrize("target", ["cube", "sphere"])
@pytest.mark.parametrize("p", [1, 2])
def test_g10_notebook_call_shape_against_the_real_sliced_cost(shw, golden, target, p):
    """VERDICT r2 missing 2: the only live call site (Flow_cube.ipynb:1381) -- N = 1200 (:200), L = 100 (:747), an
    un-normalised cube-surface evolving cloud, loss.backward().  Value 1e-5, per-slice 2e-5 (north_star / the suite's
    per-slice bar), gradient through grad_close with the count of near-tie entries pinned at two swapped pairs."""
    from helpers.compare import grad_close
    g = golden("g10_notebook_flow.npz")
    x = dev(g["source"]).requires_grad_(True)
    y = dev(g["target" if target == "cube" else "sphere"]).requires_grad_(True)
    U = dev(g["U"])
    loss = shw.sliced_cost(x, y, U, p=p)
    assert loss.dim() == 0
    loss.backward()
    assert rel(loss.item(), g[f"loss_{target}_p{p}"]) < 1e-5
    _, per, _ = shw.ssw_pair_losses(x.detach()[None], y.detach()[None], U, p=p, return_slices=True)
    assert rel(per[0].cpu().numpy(), g[f"per_slice_{target}_p{p}"]) < 2e-5
    loose = 0.2 if p == 1 else 2e-2
    grad_close(x.grad.cpu().numpy(), g[f"g_evolving_{target}_p{p}"], loose=loose, max_outside=12)
    grad_close(y.grad.cpu().numpy(), g[f"g_target_{target}_p{p}"], loose=loose, max_outside=12)


@pytest.mark.parametrize("p", [1, 2])
def test_g10_five_adam_steps_of_the_notebook_flow(shw, golden, p):
    """The notebook's loop (:1372-1395): zero_grad, loss, backward(retain_graph=True), Adam step -- five steps on the
    stored per-step directions.  Loss trace 1e-4 per step (Adam's first steps move every coordinate by ~lr whatever the
    gradient's size, so fp32 noise in tiny gradient entries shows up in the next loss at the 1e-5 level), evolved cloud
    within 2 lr of the reference's for all but a handful of coordinates."""
    g = golden("g10_notebook_flow.npz")
    lr = float(g["lr"])
    evolving = dev(g["source"]).requires_grad_(True)
    target = dev(g["target"])
    U_steps = dev(g["U_steps"])
    opt = torch.optim.Adam([evolving], lr=lr, betas=(0.9, 0.999))
    trace = []
    for i in range(U_steps.shape[0]):
        opt.zero_grad()
        loss = shw.sliced_cost(evolving, target, U_steps[i], p=p)
        loss.backward(retain_graph=True)
        opt.step()
        trace.append(loss.item())
    assert rel(trace, g[f"flow_trace_p{p}"]) < 1e-4
    moved = np.abs(evolving.detach().cpu().numpy() - g[f"flow_evolved_p{p}"])
    assert (moved > 2 * lr).sum() <= 8 and np.median(moved) < 1e-4


# ------------------------------------------------------------------------------------------- bench.py starts its own ranks
def test_bench_self_launch_path_prints_one_json_line():
    """`python bench.py --gpus N` (N > 1) must start its N ranks itself.  One GPU per box here, so the
    same code path is forced at N = 1 (SHW_BENCH_FORCE_SPAWN=1): the parent starts torch.distributed.run with one rank as
    a CHILD process (RCCL communicator, the 514-float all-reduce per step), relays rank 0's line and its exit status."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, SHW_BENCH_FORCE_SPAWN="1")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = res.stdout.strip().split("\n")
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and "roofline" in d
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert "starting 1 ranks" in res.stderr


# ------------------------------------------------------------------------------------------- deterministic gradients
@pytest.mark.parametrize("n,m,p,weighted", [(2048, 1536, 2, False), (2048, 2048, 2, True), (1000, 1300, 3, True),
                                            (300, 200, 2, False), (700, 700, 2.5, True), (4096, 3000, 2, False),
                                            (1024, 1024, 1, True), (50, 64, 2, True)])
def test_general_path_gradients_are_bit_identical_from_run_to_run(shw, n, m, p, weighted):
    """VERDICT r2 missing 4: the reference's autograd on the CPU is deterministic; round 2 accumulated the n != m /
    weighted gradient with LDS float atomics.  Round 3 computes every coefficient by its owner: three evaluations of the
    same problem must agree bit for bit, costs and gradients (with several waves per slice from 1024 points on)."""
    gen = torch.Generator().manual_seed(31 * n + m)
    B, L = 3, 24
    x, y, U = unit_cloud(gen, B, n).cuda(), unit_cloud(gen, B, m).cuda(), directions(gen, B, L).cuda()
    wu = wv = None
    if weighted:
        wu, wv = torch.rand(n, generator=gen) + 0.1, torch.rand(B, m, generator=gen) + 0.1
        wu, wv = (wu / wu.sum()).cuda(), (wv / wv.sum(1, keepdim=True)).cuda()
    runs = []
    for _ in range(3):
        xs, ys = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        pair, cost, _ = shw.ssw_pair_losses(xs, ys, U, p=p, return_slices=True, u_weights=wu, v_weights=wv)
        pair.sum().backward()
        runs.append((cost.clone(), xs.grad.clone(), ys.grad.clone()))
        torch.empty(1 << 22, device="cuda").fill_(float("nan"))          # stir the allocator between runs
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)
    assert torch.isfinite(runs[0][1]).all() and torch.isfinite(runs[0][2]).all()


# ------------------------------------------------------------------------------------------- the notebook step as a hipGraph
@pytest.mark.parametrize("fused", [False, True])
def test_graphed_step_replays_the_eager_flow_bit_for_bit(shw, golden, fused):
    """the notebooks' step (zero_grad, loss, backward, Adam step; Flow_cube.ipynb:1372-1395) captured
    once and replayed (shw.GraphedStep) must do what the eager loop does: same kernels, same buffers, same bits.  Fixed
    directions so that both loops see the same slices; 3 eager warm-up steps + capture + 6 replays against 10 eager steps."""
    g = golden("g10_notebook_flow.npz")
    target, U = dev(g["target"]), dev(g["U"])

    def flow(graphed):
        evolving = dev(g["source"]).requires_grad_(True)
        opt = torch.optim.Adam([evolving], lr=0.01, capturable=True, fused=fused)

        def loss_fn():
            return shw.sliced_cost(evolving, target, U, p=2)
        losses = []
        if graphed:
            step = shw.GraphedStep(loss_fn, opt, warmup=3)
            for _ in range(10):
                losses.append(step().clone())
        else:
            for _ in range(10):
                opt.zero_grad(set_to_none=True)
                loss = loss_fn()
                loss.backward()
                opt.step()
                losses.append(loss.detach().clone())
        return torch.stack(losses), evolving.detach().clone()

    le, xe = flow(False)
    lg, xg = flow(True)
    assert torch.equal(le, lg) and torch.equal(xe, xg)
    assert le[-1] < le[0]                                   # and it is a descent


def test_graphed_step_refuses_an_optimizer_that_cannot_be_captured(shw):
    x = torch.zeros(8, 3, device="cuda", requires_grad=True)
    with pytest.raises(RuntimeError, match="capturable"):
        shw.GraphedStep(lambda: x.sum(), torch.optim.Adam([x], lr=0.1))


# ------------------------------------------------------------------------------------------- small grids: W waves per slice
_SMALL_GRID_SCRIPT = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import shw_amd
out = {}
for (n, p, kind) in ((600, 2, "sphere"), (1000, 2, "sphere"), (1200, 2, "cube"), (1500, 3, "sphere"), (2000, 2, "sphere"),
                     (2048, 2, "sphere"), (1024, 2.5, "sphere"), (1200, 2, "lattice"), (512, 2, "sphere"),
                     (768, 2, "sphere"), (769, 2, "sphere"), (1280, 2, "sphere"), (1281, 3, "sphere"), (1536, 2, "sphere"),
                     (1700, 2, "sphere"), (1792, 2.5, "sphere"), (1793, 2, "sphere"), (1700, 2, "lattice"), (700, 2, "lattice")):
    g = torch.Generator().manual_seed(7 * n + len(kind))
    x, y = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g)
    if kind == "sphere":
        x, y = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(y, dim=-1)
    elif kind == "cube":                 # un-normalised, like the notebooks' clouds
        x = x.clamp(-1, 1)
    else:                                # few lattice sites: duplicate points, long runs of equal coordinates
        x, y = torch.round(x * 4) / 4 + 0.01, torch.round(y * 4) / 4 + 0.01
    U = torch.linalg.qr(torch.randn(10, 3, 2, generator=g))[0]
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, shift = shw_amd.ssw_pair_losses(xs[None], ys[None], U.cuda(), p=p, return_slices=True)
    pair.sum().backward()
    _, cost_fwd, shift_fwd = shw_amd.ssw_pair_losses(x.cuda()[None], y.cuda()[None], U.cuda(), p=p, return_slices=True)
    out[f"{n}p{p}{kind}"] = {"cost": cost[0].tolist(), "shift": shift[0].tolist(), "cost_fwd": cost_fwd[0].tolist(),
                            "shift_fwd": shift_fwd[0].tolist(), "gx": xs.grad.cpu().numpy().tolist(),
                            "gy": ys.grad.cpu().numpy().tolist()}
print(json.dumps(out))
"""


def test_small_grid_kernels_agree_with_the_throughput_kernels(shw):
    """a launch with fewer (pair, slice) problems than SIMDs (the notebooks: 1 pair x 100 slices) takes
    the cooperative kernels with 8 keys per lane and W = padded / 512 waves per slice.  SHW_SMALL_GRID=0 keeps the
    throughput kernels (two waves per slice; 12 / 16 / 20 / 24 / 28 / 32 keys per lane: every class of round 3 is in the list,
    at its upper edge and one point above it) for any grid: the same seeded cases in two subprocesses must give the same
    shifts, costs to 3e-6 and -- both sorts being stable -- the same gradients, also on un-normalised clouds and on clouds
    of duplicate points (which take the sorts' network fallback, through the staging buffer for the odd classes)."""
    res = {}
    for forced in ("0", "1024"):
        env = dict(os.environ, SHW_SMALL_GRID=forced)
        r = subprocess.run([sys.executable, "-c", _SMALL_GRID_SCRIPT, ROOT], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[forced] = json.loads(r.stdout.strip().split("\n")[-1])
    for key in res["0"]:
        a, b = res["0"][key], res["1024"][key]
        for f in ("cost", "cost_fwd"):
            ca, cb = np.array(a[f]), np.array(b[f])
            assert np.all(np.abs(ca - cb) <= 3e-6 * np.abs(cb) + 1e-12), (key, f)
        same_shift = np.array(a["shift"]) == np.array(b["shift"])
        assert same_shift.mean() >= 0.9, key                       # (exact cost ties may pick either shift)
        for f in ("gx", "gy"):
            ga, gb = np.array(a[f]), np.array(b[f])
            assert np.isfinite(gb).all()
            if same_shift.all():
                assert np.abs(ga - gb).max() <= 1e-6 * np.abs(ga).max() + 1e-12, (key, f)


# ------------------------------------------------------------------------------------------- pool cap, plan of one solve
def test_workspace_pool_is_bounded_over_shapes(shw):
    """idle workspaces were bounded per shape only -- every distinct (B, n, m, L) pinned its coefficient rows
    until SSWWorkspace.clear().  The pool now evicts the least recently used shapes above MAX_IDLE_BYTES."""
    ws = shw.ssw.SSWWorkspace
    ws.clear()
    old = ws.MAX_IDLE_BYTES
    ws.MAX_IDLE_BYTES = 24 << 20
    try:
        gen = torch.Generator().manual_seed(1)
        for n in (300, 400, 500, 600, 700, 800):                # 6 training shapes of 2*8*64*n*4 B = 1.2 .. 3.3 MB... x2 clouds
            x = unit_cloud(gen, 8, n).cuda().requires_grad_(True)
            y = unit_cloud(gen, 8, n).cuda()
            shw.sliced_cost(x, y, directions(gen, 8, 256).cuda(), p=2).backward()
            del x, y
            assert ws.idle_bytes() <= ws.MAX_IDLE_BYTES
        assert 0 < ws.idle_bytes() <= ws.MAX_IDLE_BYTES
        assert all(key[3] != 300 for key in ws._pools)          # the oldest shape went first
    finally:
        ws.MAX_IDLE_BYTES = old
        ws.clear()


def test_sinkhorn_training_returns_the_plan_of_the_same_solve(shw, golden):
    """with return_plan=True (the class default) a training call ran the whole solve twice.  P and C now come out
    of the trajectory's last executed slot of the ONE solve: they must equal what the value-only entry point returns, sum to
    the marginals like the reference's P (fixture G7), carry no gradient, and leave the cost's gradient as it was."""
    g = golden("g7_sinkhorn.npz")
    x, y = dev(g["x"]).requires_grad_(True), dev(g["y"]).requires_grad_(True)
    crit = shw.log_Sinkhorn_Distance_Loss(eps=0.05, max_iter=60, batch_reduction="none", type_of_cost_norm="L2")
    cost, P, C = crit(x, y, "cuda")
    assert P is not None and not P.requires_grad and not C.requires_grad
    with torch.no_grad():
        cost0, P0, C0 = crit(x.detach(), y.detach(), "cuda")
    assert torch.allclose(cost, cost0, rtol=1e-6) and torch.allclose(P, P0, rtol=1e-5, atol=1e-9) and torch.equal(C, C0)
    assert rel(P.sum(-1).cpu().numpy(), g["P_rowsum_eps0.05_it60"]) < 5e-4
    cost.sum().backward()
    gx = x.grad.clone()
    x2, y2 = dev(g["x"]).requires_grad_(True), dev(g["y"]).requires_grad_(True)
    c2, P2, _ = shw.log_Sinkhorn_Distance_Loss(eps=0.05, max_iter=60, batch_reduction="none", type_of_cost_norm="L2",
                                               return_plan=False)(x2, y2, "cuda")
    assert P2 is None
    c2.sum().backward()
    assert torch.equal(gx, x2.grad)


# ------------------------------------------------------------------------------------------- keys-per-lane classes (item 4)
@pytest.mark.parametrize("n", [600, 768, 1200, 1280, 1500, 1700, 2000])
@pytest.mark.parametrize("p", [2, 3])
def test_sizes_between_the_powers_of_two_against_cpu_oracle(shw, n, p):
    """clouds of 513..2047 points take keys-per-lane classes of 12 / 20 / 24 / 28 (the next multiple of
    256 points) in the two-wave kernels instead of the next power of two.  A launch of 4 pairs x 300 slices (more than the
    small-grid limit, so the throughput kernels run) against the float64 oracle on a sample of its slices, loss and
    gradients; plus the loss-only kernel against the training kernel on every slice."""
    from helpers.compare import grad_close
    from oracle import exact_shift
    gen = torch.Generator().manual_seed(77 * n + int(p))
    B, L = 4, 300
    x, y, U = unit_cloud(gen, B, n), unit_cloud(gen, B, n), directions(gen, B, L)
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, shift = shw.ssw_pair_losses(xs, ys, U.cuda(), p=p, return_slices=True)
    sample = [0, 1, 149, 298, 299]
    for b in (0, 3):
        cu = exact_shift.circle_coords(x[b].numpy(), U[b, sample].numpy())
        cv = exact_shift.circle_coords(y[b].numpy(), U[b, sample].numpy())
        ref64, k64 = exact_shift.circular_ot_equal(cu, cv, p=p)
        assert np.allclose(cost[b, sample].detach().cpu().numpy(), ref64, rtol=2e-5 if p == 2 else 4e-5, atol=1e-10)
    with torch.no_grad():
        _, cost_fwd, shift_fwd = shw.ssw_pair_losses(x.cuda(), y.cuda(), U.cuda(), p=p, return_slices=True)
    assert torch.allclose(cost_fwd, cost.detach(), rtol=3e-6, atol=1e-12)
    # gradients: the same problem slice-sharded into small grids (the cooperative kernels, checked against the oracle
    # elsewhere) must add up to the gradient of the one big launch
    pair.sum().backward()
    xa, ya = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    total = 0
    for l0 in range(0, L, 100):
        total = total + shw.ssw_pair_losses(xa, ya, U[:, l0:l0 + 100].cuda().contiguous(), p=p).sum() * (100.0 / L)
    total.backward()
    grad_close(xs.grad.cpu().numpy(), xa.grad.cpu().numpy(), strict=2e-5, loose=2e-2, frac=0.0005)
    grad_close(ys.grad.cpu().numpy(), ya.grad.cpu().numpy(), strict=2e-5, loose=2e-2, frac=0.0005)


def test_chamfer_gradients_are_bit_identical_from_run_to_run(shw):
    """VERDICT r2 missing 4, second half: Chamfer's backward scattered with global float atomics.  Owner-computed now: three
    runs agree bit for bit, and with autograd of the definition (brute force in float64) to 1e-5 of the largest entry."""
    gen = torch.Generator().manual_seed(99)
    B, n, m = 5, 1500, 1100
    x, y = torch.randn(B, n, 3, generator=gen), torch.randn(B, m, 3, generator=gen) * 0.9 + 0.05
    w = torch.tensor([1.0, -0.5, 2.0, 0.25, 1.5])
    runs = []
    for _ in range(3):
        xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
        (shw.chamfer_pair_losses(xs, ys) * w.cuda()).sum().backward()
        runs.append((xs.grad.clone(), ys.grad.clone()))
    for other in runs[1:]:
        assert torch.equal(runs[0][0], other[0]) and torch.equal(runs[0][1], other[1])
    xd, yd = x.double().requires_grad_(True), y.double().requires_grad_(True)
    d = ((xd[:, :, None, :] - yd[:, None, :, :]) ** 2).sum(-1)
    ((d.min(2).values.mean(1) + d.min(1).values.mean(1)) * w.double()).sum().backward()
    for got, ref in ((runs[0][0], xd.grad), (runs[0][1], yd.grad)):
        assert (got.cpu().double() - ref).abs().max() < 1e-5 * ref.abs().max()


# ------------------------------------------------------------------------------- p = 1: classes between powers of two
@pytest.mark.parametrize("n,m", [(513, 513), (600, 600), (640, 640), (700, 800), (1200, 1200), (1280, 1280), (1500, 1500),
                                 (1536, 1536), (1200, 900), (2500, 2500), (2560, 2560), (3000, 3000), (5000, 5000),
                                 (6000, 6144)])
def test_p1_cooperative_kernel_classes_against_cpu_oracle(shw, n, m):
    """The cooperative p = 1 kernel picks W waves of 20 / 24 / 32 merged atoms per lane (1280 ... 16384 slots): values of
    every class and its edges against the level-median restatement (3e-5 per slice), and the loss of the training launch
    against the loss-only launch (different kernels below 2048 points, the same above)."""
    from oracle import exact_shift
    g = torch.Generator().manual_seed(7000 + n + m)
    L = 6
    x, y, U = unit_cloud(g, n), unit_cloud(g, m), directions(g, L)
    _, cost, _ = shw.ssw_pair_losses(x.cuda().unsqueeze(0), y.cuda().unsqueeze(0), U.cuda(), p=1, return_slices=True)
    cu = exact_shift.circle_coords(x.numpy(), U.numpy())
    cv = exact_shift.circle_coords(y.numpy(), U.numpy())
    ref = np.array([exact_shift.w1_level_median(cu[l], cv[l]) for l in range(L)])
    assert np.allclose(cost[0].cpu().numpy(), ref, rtol=3e-5, atol=1e-9)
    xs = x.cuda().requires_grad_(True)
    pair, cost_t, _ = shw.ssw_pair_losses(xs.unsqueeze(0), y.cuda().unsqueeze(0), U.cuda(), p=1, return_slices=True)
    pair.sum().backward()
    assert np.allclose(cost_t[0].detach().cpu().numpy(), ref, rtol=3e-5, atol=1e-9)
    assert torch.isfinite(xs.grad).all()


@pytest.mark.parametrize("n,m", [(600, 600), (1200, 1200), (1500, 1400), (2500, 2500), (3000, 2200), (5000, 5000)])
def test_p1_training_classes_against_autograd_of_the_restatement(shw, n, m):
    """Gradients of the cooperative p = 1 training kernel in the 20 / 24 keys-per-lane classes (merged 1200 -> 1 x 20,
    2400 -> 2 x 20, 2900 -> 2 x 24, 5000 -> 4 x 20, 5200 -> 4 x 24, 10000 -> 8 x 20; below 2048 points it takes over from the
    merge kernel where its class is the smaller one) against autograd of the reference's restatement."""
    from oracle import ref_mirror
    from helpers.compare import grad_close
    g = torch.Generator().manual_seed(17 * n + m)
    x, y = unit_cloud(g, n), unit_cloud(g, m)
    U = directions(g, 4)
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    val = shw.sliced_cost(xs, ys, U.cuda(), p=1)
    val.backward()
    xc, yc = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    ref = ref_mirror.sliced_cost(xc, yc, U, p=1)
    ref.backward()
    assert abs(val.item() - ref.item()) <= 2e-5 * abs(ref.item()) + 1e-7
    grad_close(xs.grad.cpu().numpy(), xc.grad.numpy(), loose=0.1)
    grad_close(ys.grad.cpu().numpy(), yc.grad.numpy(), loose=0.1)


# ------------------------------------------------------------------------------- backward-points kernel, four points per lane
_BWD_WIDE_SCRIPT = r"""
import json, sys, hashlib, torch
sys.path.insert(0, sys.argv[1])
import shw_amd
out = {}
for (B, n, m, L, p) in ((8, 1200, 1200, 37, 2), (2, 2048, 1024, 16, 1), (1, 4, 8, 5, 2), (2, 3000, 3000, 9, 2), (2, 1200, 900, 8, 2)):
    g = torch.Generator().manual_seed(100 * n + m)
    x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).cuda().requires_grad_(True)
    y = torch.nn.functional.normalize(torch.randn(B, m, 3, generator=g), dim=-1).cuda().requires_grad_(True)
    U = shw_amd.stiefel_frames(torch.randn(B, L, 3, 2, generator=g).cuda())
    w = torch.linspace(0.5, 1.5, B).cuda()
    (shw_amd.ssw_pair_losses(x, y, U, p) * w).sum().backward()
    out[f"{n}x{m}p{p}"] = [hashlib.sha256(t.grad.cpu().numpy().tobytes()).hexdigest() for t in (x, y)]
print(json.dumps(out))
"""


def test_wide_backward_points_kernel_is_bit_identical_to_the_one_point_per_lane_kernel(shw):
    """Sizes that are multiples of 4 take `ssw_backward_points4_kernel` (a lane owns four consecutive points, 16-byte
    coefficient loads).  It adds the slices of a point in the same order as the one-point-per-lane kernel
    (`SHW_BWD_WIDE=0`): the gradients of both clouds must be the same BITS, for equal and unequal sizes, p = 1 and 2, sizes
    above 2048 and the smallest size that qualifies.  (Shapes chosen so that the one-point-per-lane kernel runs in its
    four-wave form: with fewer than 256 workgroups and at least 32 slices it splits the slices over sixteen waves, another
    order of the sixteen partial sums.)"""
    res = {}
    for wide in ("0", "2"):                         # never / whenever sizes and alignment allow (the default also asks
        env = dict(os.environ, SHW_BWD_WIDE=wide)   # for a grid that fills the chip)
        r = subprocess.run([sys.executable, "-c", _BWD_WIDE_SCRIPT, ROOT], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[wide] = json.loads(r.stdout.strip().split("\n")[-1])
    assert res["0"] == res["2"]


# ------------------------------------------------------------------------------- SHW_KPL_CLASSES=0 against the default
_KPL_SCRIPT = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import shw_amd
out = {}
for (n, p) in ((600, 2), (1200, 2), (1500, 3), (1700, 2), (3000, 2), (5000, 2), (600, 1), (1200, 1), (1500, 1), (3000, 1), (5000, 1)):
    g = torch.Generator().manual_seed(11 * n + int(p))
    x = torch.nn.functional.normalize(torch.randn(2, n, 3, generator=g), dim=-1).cuda().requires_grad_(True)
    y = torch.nn.functional.normalize(torch.randn(2, n, 3, generator=g), dim=-1).cuda().requires_grad_(True)
    U = shw_amd.stiefel_frames(torch.randn(2, 6, 3, 2, generator=g).cuda())
    pair, cost, _ = shw_amd.ssw_pair_losses(x, y, U, p, return_slices=True)
    pair.sum().backward()
    with torch.no_grad():
        _, cost_fwd, _ = shw_amd.ssw_pair_losses(x.detach(), y.detach(), U, p, return_slices=True)
    out[f"{n}p{p}"] = {"cost": cost.detach().cpu().tolist(), "cost_fwd": cost_fwd.cpu().tolist(),
                       "gx": x.grad.cpu().numpy().tolist(), "gy": y.grad.cpu().numpy().tolist()}
print(json.dumps(out))
"""


def test_keys_per_lane_classes_agree_with_the_power_of_two_classes(shw):
    """`SHW_KPL_CLASSES=0` keeps only the power-of-two keys-per-lane classes; the default adds 12 / 20 / 24 / 28 keys per lane
    below 2048 points, 20 / 24 above, and 20 / 24 merged atoms per lane in the cooperative p = 1 kernel.  Different kernels
    (and below 2048 points at p = 1 a different family: merge against cooperative) on the same seeded clouds: per-slice costs
    of the loss-only and of the training launch within 3e-6, gradients as `grad_close`."""
    from helpers.compare import grad_close
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ, SHW_KPL_CLASSES=flag)
        r = subprocess.run([sys.executable, "-c", _KPL_SCRIPT, ROOT], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[flag] = json.loads(r.stdout.strip().split("\n")[-1])
    for key in res["0"]:
        a, b = res["0"][key], res["1"][key]
        for f in ("cost", "cost_fwd"):
            ca, cb = np.array(a[f]), np.array(b[f])
            assert np.all(np.abs(ca - cb) <= 3e-6 * np.abs(cb) + 1e-12), (key, f)
        for f in ("gx", "gy"):
            grad_close(np.array(a[f]), np.array(b[f]), strict=2e-5, loose=5e-2, frac=0.002)


# ------------------------------------------------------------------------------- clouds that are not uniform in angle
def _nonuniform_cloud(kind, gen, B, n):
    """The families of tools/nonuniform_time.py (profiles/r03_nonuniform.txt): their circle coordinates crowd into few bins
    of the distribution sort -- equal-bin runs of 20..40 keys are fixed up by odd-even phases, longer ones send the slice
    to the bitonic network (csrc/bin_sort.hpp, SHW_BINSORT_MAX_RUN)."""
    x = torch.randn(B, n, 3, generator=gen)
    if kind == "cube_surface":                       # Flow_cube.ipynb:127-200: faces of [0, 1]^3, un-normalised, not centred
        face = torch.randint(0, 3, (B, n), generator=gen)
        pts = torch.rand(B, n, 3, generator=gen)
        pts.scatter_(2, face.unsqueeze(-1), torch.randint(0, 2, (B, n, 1), generator=gen).float())
        return pts
    if kind == "tight_clusters":
        centres = torch.nn.functional.normalize(torch.randn(B, 16, 3, generator=gen), dim=-1)
        pick = torch.randint(0, 16, (B, n), generator=gen)
        return torch.nn.functional.normalize(torch.gather(centres, 1, pick.unsqueeze(-1).expand(B, n, 3)) + 0.02 * x, dim=-1)
    if kind == "great_circle_band":
        x[..., 2] *= 0.02
        return torch.nn.functional.normalize(x, dim=-1)
    if kind == "duplicates_32_fold":
        base = torch.nn.functional.normalize(torch.randn(B, n // 32, 3, generator=gen), dim=-1)
        return base.repeat(1, 32, 1)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["cube_surface", "tight_clusters", "great_circle_band", "duplicates_32_fold"])
@pytest.mark.parametrize("p", [2, 1])
def test_headline_kernels_on_clouds_not_uniform_in_angle_against_cpu_oracle(shw, kind, p):
    """ADVICE r2 (data-dependent sort path): the one-wave kernels of the headline shape (N = M = 2048, the full 32-keys-per-lane
    class; 4 pairs x 300 slices, above the small-grid limit) on clouds whose slices take the long odd-even fix-up or the
    network fallback, against the float64 oracle on a sample of slices (2e-5 per slice), and the loss-only kernel against
    the training kernel on every slice (the two carry different sorts: keys only / keys with indices)."""
    from oracle import exact_shift
    gen = torch.Generator().manual_seed(4242 + len(kind) + int(p))
    B, n, L = 4, 2048, 300
    x, y, U = _nonuniform_cloud(kind, gen, B, n), _nonuniform_cloud(kind, gen, B, n), directions(gen, B, L)
    with torch.no_grad():
        _, cost_fwd, _ = shw.ssw_pair_losses(x.cuda(), y.cuda(), U.cuda(), p=p, return_slices=True)
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, _ = shw.ssw_pair_losses(xs, ys, U.cuda(), p=p, return_slices=True)
    sample = [0, 1, 77, 150, 298, 299]
    for b in (0, 3):
        cu = exact_shift.circle_coords(x[b].numpy(), U[b, sample].numpy())
        cv = exact_shift.circle_coords(y[b].numpy(), U[b, sample].numpy())
        if p == 1:
            ref64 = np.array([exact_shift.w1_level_median(cu[i], cv[i]) for i in range(len(sample))])
        else:
            ref64, _ = exact_shift.circular_ot_equal(cu, cv, p=p)
        got = cost_fwd[b, sample].cpu().numpy()
        assert np.allclose(got, ref64, rtol=2e-5, atol=1e-9), (kind, b, rel(got, ref64))
    assert torch.allclose(cost_fwd, cost.detach(), rtol=3e-6 if p == 2 else 1e-4, atol=1e-10)
    pair.sum().backward()
    assert torch.isfinite(xs.grad).all() and torch.isfinite(ys.grad).all()
