"""Pins the CPU oracle (oracle/ref_mirror.py, oracle/exact_shift.py) against golden vectors that
oracle/make_golden.py captured from the real reference (max_spherical_sliced_w.py /
max_spherical_sliced_w_fast.py) in the build container.  CPU only.

Tolerances: the restatement runs the same fp32 op chain, so it agrees with the reference to a few
fp32 ulps of the loss (1e-6 relative asserted).  The float64 exhaustive-shift oracle is compared at
1e-5 relative for power-of-two n (the reference's own fp32-vs-fp64 noise there is < 1e-7) and at
2e-5 for n = 100 where the reference itself leaves the bisection through its tangent step
(4.9e-6 measured)."""
import numpy as np
import pytest
import torch

from oracle import exact_shift, ref_mirror


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))


def T(a):
    return torch.from_numpy(np.asarray(a))


# ---------------------------------------------------------------- G1: config 1
@pytest.mark.parametrize("deg", [90, 135, 180])
@pytest.mark.parametrize("p", [1, 2])
def test_g1_config1_values_and_grads(golden, deg, p):
    g = golden("g1_config1.npz")
    x, y, U = T(g["x"]).requires_grad_(True), T(g[f"y_{deg}"]).requires_grad_(True), T(g["U"])
    per = ref_mirror.per_slice_costs(x, y, U, p=p)
    loss = per.mean()
    loss.backward()
    assert rel(loss.detach().numpy(), g[f"loss_{deg}_p{p}"]) < 1e-6
    assert np.allclose(per.detach().numpy(), g[f"per_slice_{deg}_p{p}"], rtol=2e-5, atol=1e-9)
    scale = np.abs(g[f"gx_{deg}_p{p}"]).max()
    assert np.abs(x.grad.numpy() - g[f"gx_{deg}_p{p}"]).max() < 1e-4 * scale
    assert np.abs(y.grad.numpy() - g[f"gy_{deg}_p{p}"]).max() < 1e-4 * scale


@pytest.mark.parametrize("deg", [90, 135, 180])
def test_g1_exact_shift_matches_reference(golden, deg):
    g = golden("g1_config1.npz")
    val = exact_shift.ssw_pair(g["x"], g[f"y_{deg}"], g["U"], p=2)
    assert abs(val - float(g[f"loss_{deg}_p2"])) < 1e-5 * float(g[f"loss_{deg}_p2"])
    val1 = exact_shift.ssw_pair(g["x"], g[f"y_{deg}"], g["U"], p=1)
    assert abs(val1 - float(g[f"loss_{deg}_p1"])) < 1e-5 * float(g[f"loss_{deg}_p1"])


def test_g1_analytic_gradient_matches_reference_autograd(golden):
    g = golden("g1_config1.npz")
    gx, gy = exact_shift.ssw_pair_grad(g["x"], g["y_135"], g["U"], p=2)
    scale = np.abs(g["gx_135_p2"]).max()
    assert np.abs(gx - g["gx_135_p2"]).max() < 2e-4 * scale
    assert np.abs(gy - g["gy_135_p2"]).max() < 2e-4 * scale


# ---------------------------------------------------------------- G2: batched entry
@pytest.mark.parametrize("p", [2, 3])
def test_g2_batched_sum_over_pairs(golden, p):
    g = golden("g2_batched.npz")
    x, y, U = T(g["x"]).requires_grad_(True), T(g["y"]).requires_grad_(True), T(g["U"])
    val = ref_mirror.sliced_cost_batched(x, y, U, p=p)
    assert val.shape == (1,)
    val.backward()
    assert rel(val.detach().numpy(), g[f"value_p{p}"]) < 1e-6
    assert rel(g[f"per_pair_p{p}"].sum(), g[f"value_p{p}"]) < 1e-6   # sum, not mean, over the batch
    scale = np.abs(g[f"gx_p{p}"]).max()
    assert np.abs(x.grad.numpy() - g[f"gx_p{p}"]).max() < 1e-4 * scale
    assert np.abs(y.grad.numpy() - g[f"gy_p{p}"]).max() < 1e-4 * scale


# ---------------------------------------------------------------- G3: circle level
@pytest.mark.parametrize("tag", ["64x64", "100x100", "256x256", "128x100"])
@pytest.mark.parametrize("p", [2, 3])
def test_g3_bisection_rows(golden, tag, p):
    g = golden("g3_circle.npz")
    u, v = T(g[f"u_{tag}"]), T(g[f"v_{tag}"])
    got32 = ref_mirror.circular_ot_bisect(u, v, p=p).numpy()
    got64 = ref_mirror.circular_ot_bisect(u.double(), v.double(), p=p).numpy()
    assert rel(got32, g[f"bsc_p{p}_{tag}_f32"]) < 2e-6
    assert rel(got64, g[f"bsc_p{p}_{tag}_f64"]) < 1e-12


@pytest.mark.parametrize("tag", ["64x64", "100x100", "256x256", "128x100"])
def test_g3_level_median_rows(golden, tag):
    g = golden("g3_circle.npz")
    u, v = T(g[f"u_{tag}"]), T(g[f"v_{tag}"])
    assert rel(ref_mirror.circular_w1_level_median(u, v).numpy(), g[f"emd1_{tag}_f32"]) < 2e-6
    assert rel(ref_mirror.circular_w1_level_median(u.double(), v.double()).numpy(), g[f"emd1_{tag}_f64"]) < 1e-12
    scalar = [exact_shift.w1_level_median(g[f"u_{tag}"][r], g[f"v_{tag}"][r]) for r in range(8)]
    assert rel(scalar, g[f"emd1_{tag}_f64"]) < 1e-6     # inputs are fp32 values, levels differ by rounding only


@pytest.mark.parametrize("tag,tol", [("64x64", 1e-5), ("256x256", 1e-5), ("100x100", 2e-5)])
@pytest.mark.parametrize("p", [2, 3])
def test_g3_min_over_shifts_equals_bisection(golden, tag, tol, p):
    g = golden("g3_circle.npz")
    cost, _ = exact_shift.circular_ot_equal(g[f"u_{tag}"], g[f"v_{tag}"], p=p)
    assert rel(cost, g[f"bsc_p{p}_{tag}_f32"]) < tol
    assert rel(cost, g[f"bsc_p{p}_{tag}_f64"]) < 1e-6


# ---------------------------------------------------------------- G3b: binary_search_circle at its default p = 1 (round 3)
G3B_PLAIN = ["64x64", "100x100", "256x256", "128x100", "96x96", "80x96", "1200x1200", "1000x750"]
G3B_WEIGHTED = ["96x96", "80x96", "1200x1200", "1000x750"]


def _g3b_rows(golden, tag):
    g = golden("g3b_bisection_p1.npz")
    src = g if f"u_{tag}" in g.files else golden("g3_circle.npz")
    return g, T(src[f"u_{tag}"]), T(src[f"v_{tag}"])


@pytest.mark.parametrize("tag", G3B_PLAIN)
def test_g3b_bisection_at_p1_is_not_the_level_median(golden, tag):
    """VERDICT r2 missing 1: `binary_search_circle(u, v)` (default p = 1, max_spherical_sliced_w.py:117) bisects and ends in
    Cost's p == 1 branch (:107-108).  The restatement must return the reference's numbers -- and those differ from
    emd1D_circle's (the omitted wrap segment) by far more than any tolerance in this suite."""
    g, u, v = _g3b_rows(golden, tag)
    assert rel(ref_mirror.circular_ot_bisect(u, v, p=1).numpy(), g[f"bsc_p1_{tag}_f32"]) < 2e-6
    assert rel(ref_mirror.circular_ot_bisect(u.double(), v.double(), p=1).numpy(), g[f"bsc_p1_{tag}_f64"]) < 1e-12
    level_median = ref_mirror.circular_w1_level_median(u, v).numpy()
    assert rel(level_median, g[f"bsc_p1_{tag}_f32"]) > 1e-4


@pytest.mark.parametrize("tag", G3B_WEIGHTED)
def test_g3b_weighted_bisection_at_p1(golden, tag):
    g, u, v = _g3b_rows(golden, tag)
    wu, wv = T(g[f"wu_{tag}"]), T(g[f"wv_{tag}"])
    a, b = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
    got = ref_mirror.circular_ot_bisect(a, b, p=1, u_weights=wu, v_weights=wv)
    got.sum().backward()
    assert rel(got.detach().numpy(), g[f"bsc_p1_w_{tag}_f32"]) < 2e-6
    assert np.abs(a.grad.numpy() - g[f"bsc_p1_w_{tag}_gu"]).max() < 1e-6
    assert np.abs(b.grad.numpy() - g[f"bsc_p1_w_{tag}_gv"]).max() < 1e-6


@pytest.mark.parametrize("tag", ["64x64", "100x100", "256x256", "96x96", "1200x1200"])
def test_g3b_min_over_shifts_equals_the_bisection_at_p1(golden, tag):
    """Row A8 holds at p = 1 too (the cost is convex piecewise linear in the cut, its minimum sits on a kink): for equal
    sizes and uniform weights the equal-size HIP kernels (min over cyclic shifts, |.|^1) serve binary_search_circle(p=1)."""
    g, u, v = _g3b_rows(golden, tag)
    cost, _ = exact_shift.circular_ot_equal(u.numpy(), v.numpy(), p=1)
    assert rel(cost, g[f"bsc_p1_{tag}_f64"]) < 1e-9
    assert rel(cost, g[f"bsc_p1_{tag}_f32"]) < 2e-6


# ---------------------------------------------------------------- G10: the notebooks' call shape (round 3)
@pytest.mark.parametrize("target", ["cube", "sphere"])
@pytest.mark.parametrize("p", [1, 2])
def test_g10_notebook_shape_values_and_gradients(golden, target, p):
    """Flow_cube.ipynb:1381 -- N = 1200, L = 100, un-normalised cube-surface evolving cloud: the restatement against the
    real sliced_cost (value, per-slice costs, d loss / d evolving)."""
    g = golden("g10_notebook_flow.npz")
    x = T(g["source"]).clone().requires_grad_(True)
    y = T(g["target" if target == "cube" else "sphere"])
    U = T(g["U"])
    val = ref_mirror.sliced_cost(x, y, U, p=p)
    val.backward()
    assert rel(val.detach().numpy(), g[f"loss_{target}_p{p}"]) < 2e-6
    per = ref_mirror.per_slice_costs(x.detach(), y, U, p=p).numpy()
    assert rel(per, g[f"per_slice_{target}_p{p}"]) < 1e-5
    scale = np.abs(g[f"g_evolving_{target}_p{p}"]).max()
    assert np.abs(x.grad.numpy() - g[f"g_evolving_{target}_p{p}"]).max() < 1e-4 * scale


def test_g10_min_over_shifts_on_the_notebook_shape(golden):
    g = golden("g10_notebook_flow.npz")
    for target in ("target", "sphere"):
        tag = "cube" if target == "target" else "sphere"
        val = exact_shift.ssw_pair(g["source"], g[target], g["U"], p=2)
        assert abs(val - float(g[f"loss_{tag}_p2"])) < 2e-6 * val


# ---------------------------------------------------------------- G4: edges
def test_g4_identical_clouds_are_exactly_zero(golden):
    g = golden("g4_edges.npz")
    x, U = T(g["x"]), T(g["U"])
    for p in (1, 2):
        assert float(g[f"identical_p{p}"]) == 0.0
        assert float(ref_mirror.sliced_cost(x, x.clone(), U, p=p)) == 0.0


@pytest.mark.parametrize("p", [1, 2])
def test_g4_zero_target(golden, p):
    g = golden("g4_edges.npz")
    x, U = T(g["x"]).requires_grad_(True), T(g["U"])
    val = ref_mirror.sliced_cost(x, torch.zeros(256, 3), U, p=p)
    val.backward()
    assert rel(val.detach().numpy(), g[f"zero_target_p{p}"]) < 1e-6
    scale = np.abs(g[f"zero_target_gx_p{p}"]).max()
    assert np.abs(x.grad.numpy() - g[f"zero_target_gx_p{p}"]).max() < 1e-4 * scale
    if p == 2:
        assert abs(exact_shift.ssw_pair(g["x"], np.zeros((256, 3)), g["U"], 2) - float(g["zero_target_p2"])) \
            < 1e-5 * float(g["zero_target_p2"])


@pytest.mark.parametrize("p", [1, 2])
def test_g4_unnormalised_cube(golden, p):
    g = golden("g4_edges.npz")
    a, b, U = T(g["cube"]).requires_grad_(True), T(g["blob"]).requires_grad_(True), T(g["U"])
    per = ref_mirror.per_slice_costs(a, b, U, p=p)
    per.mean().backward()
    assert rel(per.mean().detach().numpy(), g[f"cube_loss_p{p}"]) < 1e-6
    assert np.allclose(per.detach().numpy(), g[f"cube_per_slice_p{p}"], rtol=2e-5, atol=1e-9)
    scale = np.abs(g[f"cube_gx_p{p}"]).max()
    assert np.abs(a.grad.numpy() - g[f"cube_gx_p{p}"]).max() < 1e-4 * scale
    assert np.abs(b.grad.numpy() - g[f"cube_gy_p{p}"]).max() < 1e-4 * np.abs(g[f"cube_gy_p{p}"]).max()


@pytest.mark.parametrize("p", [1, 2])
def test_g4_unequal_sizes_and_weights(golden, p):
    g = golden("g4_edges.npz")
    x, y, U = T(g["x"]), T(g["y200"]), T(g["U"])
    per = ref_mirror.per_slice_costs(x, y, U, p=p)
    assert np.allclose(per.numpy(), g[f"n256_m200_per_slice_p{p}"], rtol=2e-5, atol=1e-9)
    assert rel(per.mean().numpy(), g[f"n256_m200_loss_p{p}"]) < 1e-6
    w = ref_mirror.sliced_cost(T(g["x128"]), T(g["y128"]), U, p=p, u_weights=T(g["wu"]), v_weights=T(g["wv"]))
    assert rel(w.numpy(), g[f"weighted_loss_p{p}"]) < 1e-6


# ---------------------------------------------------------------- G5: RNG stream parity
def test_g5_direction_sampling_consumes_generator_like_reference(golden):
    g = golden("g5_rng.npz")
    torch.manual_seed(int(g["seed"]))
    U = ref_mirror.draw_directions(24)
    assert np.array_equal(U.numpy(), g["U_pair"])
    torch.manual_seed(int(g["seed"]))
    val = ref_mirror.sliced_wasserstein_sphere(T(g["x"]), T(g["y"]), 24, "cpu", p=2)
    assert rel(val.numpy(), g["value_pair"]) < 1e-6
    torch.manual_seed(int(g["seed"]))
    Ub = ref_mirror.draw_directions(12, batch=3)
    assert np.array_equal(Ub.numpy(), g["U_batched"])
    torch.manual_seed(int(g["seed"]))
    valb = ref_mirror.sliced_wasserstein_sphere_fast(T(g["xb"]), T(g["yb"]), 12, "cpu", p=2)
    assert rel(valb.numpy(), g["value_batched"]) < 1e-6


# ---------------------------------------------------------------- G6: headline shapes
@pytest.mark.parametrize("tag", ["c2", "c3"])
def test_g6_headline_shapes(golden, tag):
    g = golden("g6_headline_shapes.npz")
    x, y, U = T(g[f"x_{tag}"]), T(g[f"y_{tag}"]), T(g[f"U_{tag}"])
    val = ref_mirror.sliced_cost_batched(x, y, U, p=2)
    assert rel(val.numpy(), g[f"value_{tag}_p2"]) < 1e-6
    for b in range(x.shape[0]):
        for p in (1, 2):
            per = ref_mirror.per_slice_costs(x[b], y[b], U[b], p=p).numpy()
            assert np.allclose(per, g[f"per_slice_{tag}_p{p}"][b], rtol=2e-5, atol=1e-9)


# ---------------------------------------------------------------- G7: log-domain Sinkhorn
@pytest.mark.parametrize("eps,iters", [(0.05, 60), (0.01, 100)])
def test_g7_sinkhorn_mirror(golden, eps, iters):
    from oracle import sinkhorn_mirror
    g = golden("g7_sinkhorn.npz")
    cost, P, C, _ = sinkhorn_mirror.sinkhorn_costs(T(g["x"]), T(g["y"]), eps, iters)
    tag = f"eps{eps}_it{iters}"
    assert rel(cost.numpy(), g[f"cost_{tag}"]) < 2e-5
    assert np.allclose(P.sum(-1).numpy(), g[f"P_rowsum_{tag}"], rtol=1e-4, atol=1e-7)
    assert np.allclose(C[0, 0].numpy(), g["C_first_row"], rtol=1e-6)
    l1, _, _, _ = sinkhorn_mirror.sinkhorn_costs(T(g["x"]), T(g["y"]), 0.05, 60, norm_p=1)
    assert rel(l1.sum().numpy(), g["cost_L1_sum"]) < 2e-5
    n2, _, _, _ = sinkhorn_mirror.sinkhorn_costs(T(g["x"]), T(g["y"]), 0.05, 60, cost_pow=2)
    assert rel(n2.pow(0.5).mean().numpy(), g["cost_N2_mean"]) < 2e-5


# ---------------------------------------------------------------- G8: phi-max wrappers (host logic of the package)
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


@pytest.mark.parametrize("tag", ["pair", "fast"])
@pytest.mark.parametrize("mode", ["train", "test"])
def test_g8_phi_max_wrappers_host_logic_with_the_cpu_oracle_as_ssw(golden, tag, mode):
    """The package's wrappers (modules.py) are device-agnostic host code: driven on CPU with an SSW callable that
    evaluates the CPU oracle on the fixture's fixed directions they must retrace what the REAL wrappers
    (max_spherical_sliced_w.py:498-536, _fast.py:346-380) did with the real sliced_cost: the per-iteration ssw
    values the reference printed, the phi weights after max_iter Adam ascent steps, the returned value and maps."""
    import shw_amd
    g = golden("g8_phi_max.npz")
    U = T(g["U_pair"] if tag == "pair" else g["U_batch"])
    if tag == "pair":
        cls = shw_amd.max_spherical_wassersten_distance
        ssw = lambda a, b, L, device, p=2: ref_mirror.sliced_cost(a, b, U, p=p)           # noqa: E731
    else:
        cls = shw_amd.max_spherical_wassersten_distance_fast
        ssw = lambda a, b, L, device, p=2: ref_mirror.sliced_cost_batched(a, b, U, p=p)   # noqa: E731
    phi = LinearSphereMap(g["W0"], g["b0"])
    opt = torch.optim.Adam(phi.parameters(), lr=float(g["lr"]))
    crit = cls(U.shape[-3], phi, ssw, opt, p=2, max_iter=int(g["max_iter"]), device="cpu")
    trace = []
    crit.on_inner_value = trace.append
    a, b = T(g["first"]).requires_grad_(True), T(g["second"]).requires_grad_(True)
    val, fa, fb = crit(a, b, train_or_test=mode)
    key = f"{tag}_{mode}"
    assert rel(val.detach().numpy().reshape(-1), g[f"{key}_ssw"]) < 1e-5
    assert len(trace) == len(g[f"{key}_trace"])
    if trace:
        assert rel(np.array(trace), g[f"{key}_trace"]) < 1e-5
    assert np.abs(phi.lin.weight.detach().numpy() - g[f"{key}_W"]).max() < 1e-5
    assert np.abs(phi.lin.bias.detach().numpy() - g[f"{key}_b"]).max() < 1e-5
    assert np.abs(fa.detach().numpy() - g[f"{key}_phi_first"]).max() < 1e-5
    opt.zero_grad()
    val.sum().backward()
    scale = np.abs(g[f"{key}_g_first"]).max()
    assert np.abs(a.grad.numpy() - g[f"{key}_g_first"]).max() < 1e-3 * scale
    assert np.abs(b.grad.numpy() - g[f"{key}_g_second"]).max() < 1e-3 * scale


# ---------------------------------------------------------------- G9: the notebook's Euclidean sliced-W cell
@pytest.mark.parametrize("tag", ["n200", "n1200"])
@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("L", [1, 50])
def test_g9_euclidean_restatement_against_the_notebook_cell(golden, tag, p, L):
    from oracle import euclid_sw
    g = golden("g9_notebook_esw.npz")
    key = f"{tag}_p{p}_L{L}"
    a = T(g[f"first_{tag}"]).requires_grad_(True)
    val = euclid_sw.sliced_wasserstein_distance(a, T(g[f"second_{tag}"]), T(g[f"swd_theta_{key}"]), p=p)
    val.backward()
    assert rel(val.detach().numpy(), g[f"swd_{key}"]) < 2e-6
    scale = np.abs(g[f"swd_gfirst_{key}"]).max()
    assert np.abs(a.grad.numpy() - g[f"swd_gfirst_{key}"]).max() < 1e-4 * scale
    # the directions the cell drew are what rand_projections draws from the same seed (generator parity)
    import shw_amd
    torch.manual_seed(int(g[f"swd_seed_{key}"]))
    assert np.array_equal(shw_amd.rand_projections(3, L).numpy(), g[f"swd_theta_{key}"])


# ---------------------------------------------------------------- the live trainer criterion (host logic)
@pytest.mark.parametrize("mode", ["train", "test"])
def test_live_criterion_mirror_against_its_restatement(golden, mode):
    """modules.max_cos_disimilarity_wassersten_distance (mirror of s2_wasserstein.py:211-262) is host code: on CPU,
    with a CSW built from the CPU oracle, it must retrace the statement-by-statement restatement in
    oracle/phi_max_mirror.py (parity unpinned by fixtures: the reference module needs POT to import)."""
    import shw_amd
    from oracle import phi_max_mirror
    g = golden("g8_phi_max.npz")
    U = T(g["U_batch"])

    def csw(a, b):
        pair = torch.stack([ref_mirror.per_slice_costs(a[i], b[i], U[i], p=2).mean() for i in range(a.shape[0])])
        return phi_max_mirror.csw_from_pair_losses(pair, 2)

    results = []
    for impl in ("mirror", "restatement"):
        phi = LinearSphereMap(g["W0"], g["b0"])
        opt = torch.optim.Adam(phi.parameters(), lr=0.05)
        a, b = T(g["first"]).requires_grad_(True), T(g["second"])
        if impl == "mirror":
            crit = shw_amd.max_cos_disimilarity_wassersten_distance(phi, csw, "cpu", opt, max_iter=3, lam=0.1)
            val, fa, fb = crit(a, b, train_or_test=mode)
        else:
            val, fa, fb, _ = phi_max_mirror.criterion_forward(phi, csw, opt, a, b, 3, 0.1, mode)
        val.backward()
        results.append((val.item(), phi.lin.weight.detach().clone(), fa.detach().clone(), a.grad.clone()))
    assert results[0][0] == results[1][0]
    for x, y in zip(results[0][1:], results[1][1:]):
        assert torch.equal(x, y)


# ---------------------------------------------------------------- G7b: Sinkhorn gradients (the unrolled iterations)
@pytest.mark.parametrize("tag,eps,iters,norm_p,cost_pow,red", [("eps0.05_it60", 0.05, 60, 2, 1, "sum"),
                                                               ("eps0.1_it5", 0.1, 5, 2, 1, "mean"),
                                                               ("L1_eps0.05_it30", 0.05, 30, 1, 1, "sum"),
                                                               ("N2_eps0.05_it30", 0.05, 30, 2, 2, "mean")])
def test_g7b_sinkhorn_restatement_autograd(golden, tag, eps, iters, norm_p, cost_pow, red):
    from oracle import sinkhorn_mirror
    g = golden("g7b_sinkhorn_grad.npz")
    x, y = T(g["x"]).requires_grad_(True), T(g["y"]).requires_grad_(True)
    cost = sinkhorn_mirror.sinkhorn_costs(x, y, eps, iters, norm_p=norm_p, cost_pow=cost_pow)[0]
    if cost_pow != 1:
        cost = cost.pow(1.0 / cost_pow)
    total = cost.sum() if red == "sum" else cost.mean()
    total.backward()
    assert rel(total.detach().numpy(), g[f"cost_{tag}"]) < 2e-5
    assert np.abs(x.grad.numpy() - g[f"gx_{tag}"]).max() < 1e-4 * np.abs(g[f"gx_{tag}"]).max()
    assert np.abs(y.grad.numpy() - g[f"gy_{tag}"]).max() < 1e-4 * np.abs(g[f"gy_{tag}"]).max()
