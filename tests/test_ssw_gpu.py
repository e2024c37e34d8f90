"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the C ABI via
the package's Python mirror of the reference signatures, against
  (a) golden vectors captured from the real reference (tests/golden, made by oracle/make_golden.py),
  (b) the CPU oracle (oracle/ref_mirror.py, oracle/exact_shift.py) on seeded inputs,
  (c) size-independent properties at the full BASELINE sizes (B=64, N=2048, L=512).

Tolerances (fp32 path, north_star asks for 1e-5 relative on the loss):
  loss / per-pair values ......... 1e-5 relative
  per-slice costs ................ 2e-5 relative (a single slice has no averaging; the reference's own
                                   fp32-vs-fp64 noise is up to 5e-6 at non-power-of-two n)
  gradients ...................... see grad_close(): the loss is piecewise smooth in the inputs -- its
                                   gradient JUMPS when two points of a cloud swap places in a slice's sort.
                                   Two fp32 evaluations of the coordinates (torch's atan2 there, the kernel's
                                   polynomial here) order near-equal coordinates (|du| ~ 1e-7) differently in a
                                   few slices, which moves the affected entries by ~(target gap)/(n L).  So:
                                   every entry within 2e-2 of the largest entry, and all but 0.15 % of the
                                   entries (or 8 entries) within 2e-4 of it; the fixture cases G1/G2/G4 pin the
                                   COUNT of entries outside 2e-4 at one swapped pair (6; measured: 0, once 4);
                                   cases without near-ties (small n) are held to 2e-4 everywhere.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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


from helpers.compare import grad_close  # noqa: E402  (count of entries outside the strict bound is recorded)


def unit_cloud(gen, *shape):
    return torch.nn.functional.normalize(torch.randn(*shape, 3, generator=gen), dim=-1)


def directions(gen, *shape):
    return torch.linalg.qr(torch.randn(*shape, 3, 2, generator=gen))[0]


# ------------------------------------------------------------------------------ golden: G1
@pytest.mark.parametrize("deg", [90, 135, 180])
def test_g1_config1_loss_slices_grads_p2(shw, golden, deg):
    g = golden("g1_config1.npz")
    x, y, U = dev(g["x"]).requires_grad_(True), dev(g[f"y_{deg}"]).requires_grad_(True), dev(g["U"])
    pair, cost, _ = shw.ssw_pair_losses(x.unsqueeze(0), y.unsqueeze(0), U, p=2, return_slices=True)
    loss = shw.sliced_cost(x, y, U, p=2)
    assert loss.dim() == 0
    loss.backward()
    assert rel(loss.item(), g[f"loss_{deg}_p2"]) < 1e-5
    assert rel(pair[0].item(), g[f"loss_{deg}_p2"]) < 1e-5
    assert np.allclose(cost[0].cpu().numpy(), g[f"per_slice_{deg}_p2"], rtol=2e-5, atol=1e-10)
    grad_close(x.grad.cpu().numpy(), g[f"gx_{deg}_p2"], max_outside=6)
    grad_close(y.grad.cpu().numpy(), g[f"gy_{deg}_p2"], max_outside=6)


@pytest.mark.parametrize("deg", [90, 135, 180])
def test_g1_config1_loss_slices_grads_p1(shw, golden, deg):
    g = golden("g1_config1.npz")
    x, y, U = dev(g["x"]).requires_grad_(True), dev(g[f"y_{deg}"]).requires_grad_(True), dev(g["U"])
    pair, cost, _ = shw.ssw_pair_losses(x.unsqueeze(0), y.unsqueeze(0), U, p=1, return_slices=True)
    pair.sum().backward()
    assert rel(pair[0].item(), g[f"loss_{deg}_p1"]) < 1e-5
    assert np.allclose(cost[0].cpu().numpy(), g[f"per_slice_{deg}_p1"], rtol=2e-5, atol=1e-10)
    grad_close(x.grad.cpu().numpy(), g[f"gx_{deg}_p1"], max_outside=6)
    grad_close(y.grad.cpu().numpy(), g[f"gy_{deg}_p1"], max_outside=6)


# ------------------------------------------------------------------------------ golden: G2
@pytest.mark.parametrize("p", [2, 3])
def test_g2_batched_is_sum_over_pairs(shw, golden, p):
    g = golden("g2_batched.npz")
    x, y, U = dev(g["x"]).requires_grad_(True), dev(g["y"]).requires_grad_(True), dev(g["U"])
    val = shw.sliced_cost(x, y, U, p=p)
    assert tuple(val.shape) == (1,)
    val.backward()
    assert rel(val.item(), g[f"value_p{p}"]) < 1e-5
    pair = shw.ssw_pair_losses(x.detach(), y.detach(), U, p=p)
    assert np.allclose(pair.cpu().numpy(), g[f"per_pair_p{p}"], rtol=1e-5)
    grad_close(x.grad.cpu().numpy(), g[f"gx_p{p}"], max_outside=6)
    grad_close(y.grad.cpu().numpy(), g[f"gy_p{p}"], max_outside=6)


# ------------------------------------------------------------------------------ golden: G3 (circle level, through
# a planar embedding: points (cos 2 pi c, sin 2 pi c, 0) projected on the frame (e1, e2) have coordinate
# (atan2(-sin, -cos) + pi) / 2 pi = c up to rounding)
@pytest.mark.parametrize("tag,tol", [("64x64", 2e-5), ("100x100", 3e-5), ("256x256", 2e-5)])
@pytest.mark.parametrize("p", [2, 3])
def test_g3_circle_rows_through_planar_embedding(shw, golden, tag, tol, p):
    g = golden("g3_circle.npz")
    u, v = g[f"u_{tag}"].astype(np.float64), g[f"v_{tag}"].astype(np.float64)

    def embed(c):
        ang = 2 * np.pi * c
        return np.stack([np.cos(ang), np.sin(ang), np.zeros_like(ang)], -1).astype(np.float32)

    U = np.zeros((1, 3, 2), dtype=np.float32)
    U[0, 0, 0] = U[0, 1, 1] = 1.0
    _, cost, _ = shw.ssw_pair_losses(dev(embed(u)), dev(embed(v)), dev(U), p=p, return_slices=True)
    # embedding + re-projection perturbs each coordinate by ~1e-7, i.e. ~1e-5 relative on a row cost
    assert rel(cost[:, 0].cpu().numpy(), g[f"bsc_p{p}_{tag}_f64"]) < 5 * tol


@pytest.mark.parametrize("tag", ["64x64", "100x100", "256x256", "128x100"])
def test_g3_level_median_rows_through_planar_embedding(shw, golden, tag):
    g = golden("g3_circle.npz")
    u, v = g[f"u_{tag}"].astype(np.float64), g[f"v_{tag}"].astype(np.float64)

    def embed(c):
        ang = 2 * np.pi * c
        return np.stack([np.cos(ang), np.sin(ang), np.zeros_like(ang)], -1).astype(np.float32)

    U = np.zeros((1, 3, 2), dtype=np.float32)
    U[0, 0, 0] = U[0, 1, 1] = 1.0
    _, cost, _ = shw.ssw_pair_losses(dev(embed(u)), dev(embed(v)), dev(U), p=1, return_slices=True)
    assert rel(cost[:, 0].cpu().numpy(), g[f"emd1_{tag}_f64"]) < 1e-4


# ------------------------------------------------------------------------------ golden: G4 edges
def test_g4_identical_clouds_exactly_zero(shw, golden):
    g = golden("g4_edges.npz")
    x, U = dev(g["x"]), dev(g["U"])
    assert float(shw.sliced_cost(x, x.clone(), U, p=2)) == 0.0
    assert float(shw.sliced_cost(x, x.clone(), U, p=1)) == 0.0


@pytest.mark.parametrize("p", [1, 2])
def test_g4_zero_target(shw, golden, p):
    g = golden("g4_edges.npz")
    x, U = dev(g["x"]).requires_grad_(True), dev(g["U"])
    z = torch.zeros(256, 3, device="cuda")
    val = shw.sliced_cost(x, z, U, p=p)
    val.backward()
    assert rel(val.item(), g[f"zero_target_p{p}"]) < 1e-5
    grad_close(x.grad.cpu().numpy(), g[f"zero_target_gx_p{p}"], max_outside=6)


@pytest.mark.parametrize("p", [1, 2])
def test_g4_unnormalised_cube(shw, golden, p):
    g = golden("g4_edges.npz")
    a, b, U = dev(g["cube"]).requires_grad_(True), dev(g["blob"]).requires_grad_(True), dev(g["U"])
    pair, cost, _ = shw.ssw_pair_losses(a.unsqueeze(0), b.unsqueeze(0), U, p=p, return_slices=True)
    pair.sum().backward()
    assert rel(pair[0].item(), g[f"cube_loss_p{p}"]) < 1e-5
    assert np.allclose(cost[0].cpu().numpy(), g[f"cube_per_slice_p{p}"], rtol=2e-5, atol=1e-10)
    grad_close(a.grad.cpu().numpy(), g[f"cube_gx_p{p}"], max_outside=6)
    grad_close(b.grad.cpu().numpy(), g[f"cube_gy_p{p}"], max_outside=6)


def test_g4_unequal_sizes_p1(shw, golden):
    """n != m is supported by the p = 1 kernel (p != 1 with n != m is rejected, see below)."""
    g = golden("g4_edges.npz")
    x, y, U = dev(g["x"]).requires_grad_(True), dev(g["y200"]).requires_grad_(True), dev(g["U"])
    pair, cost, _ = shw.ssw_pair_losses(x.unsqueeze(0), y.unsqueeze(0), U, p=1, return_slices=True)
    pair.sum().backward()
    assert rel(pair[0].item(), g["n256_m200_loss_p1"]) < 1e-5
    assert np.allclose(cost[0].cpu().numpy(), g["n256_m200_per_slice_p1"], rtol=2e-5, atol=1e-10)
    grad_close(x.grad.cpu().numpy(), g["n256_m200_gx_p1"], max_outside=6)
    grad_close(y.grad.cpu().numpy(), g["n256_m200_gy_p1"], max_outside=6)


# ------------------------------------------------------------------------------ general solver: n != m, weights
def test_g4_unequal_sizes_p2_general_solver(shw, golden):
    g = golden("g4_edges.npz")
    x, y, U = dev(g["x"]).requires_grad_(True), dev(g["y200"]).requires_grad_(True), dev(g["U"])
    pair, cost, _ = shw.ssw_pair_losses(x.unsqueeze(0), y.unsqueeze(0), U, p=2, return_slices=True)
    pair.sum().backward()
    assert rel(pair[0].item(), g["n256_m200_loss_p2"]) < 1e-5
    assert np.allclose(cost[0].cpu().numpy(), g["n256_m200_per_slice_p2"], rtol=3e-5, atol=1e-10)
    grad_close(x.grad.cpu().numpy(), g["n256_m200_gx_p2"], max_outside=6)
    grad_close(y.grad.cpu().numpy(), g["n256_m200_gy_p2"], max_outside=6)


def test_g4_weighted_p2_general_solver(shw, golden):
    g = golden("g4_edges.npz")
    x, y, U = dev(g["x128"]), dev(g["y128"]), dev(g["U"])
    val = shw.sliced_cost(x, y, U, p=2, u_weights=dev(g["wu"]), v_weights=dev(g["wv"]))
    assert val.dim() == 0
    assert rel(val.item(), g["weighted_loss_p2"]) < 1e-5
    val1 = shw.sliced_cost(x, y, U, p=1, u_weights=dev(g["wu"]), v_weights=dev(g["wv"]))
    assert rel(val1.item(), g["weighted_loss_p1"]) < 1e-5


@pytest.mark.parametrize("p", [2, 3])
def test_g3_unequal_rows_through_planar_embedding(shw, golden, p):
    g = golden("g3_circle.npz")
    u, v = g["u_128x100"].astype(np.float64), g["v_128x100"].astype(np.float64)

    def embed(c):
        ang = 2 * np.pi * c
        return np.stack([np.cos(ang), np.sin(ang), np.zeros_like(ang)], -1).astype(np.float32)

    U = np.zeros((1, 3, 2), dtype=np.float32)
    U[0, 0, 0] = U[0, 1, 1] = 1.0
    _, cost, _ = shw.ssw_pair_losses(dev(embed(u)), dev(embed(v)), dev(U), p=p, return_slices=True)
    assert rel(cost[:, 0].cpu().numpy(), g[f"bsc_p{p}_128x100_f64"]) < 1e-4


@pytest.mark.parametrize("n,m,weighted", [(64, 64, True), (100, 37, False), (100, 37, True), (300, 512, False),
                                          (1000, 1024, True), (2048, 1500, False), (1, 5, False), (7, 1, True)])
@pytest.mark.parametrize("p", [2, 3])
def test_general_solver_against_cpu_oracle(shw, n, m, weighted, p):
    """Loss and gradients of the general path against the torch restatement of the reference's bisection
    (oracle/ref_mirror.py, pinned by the golden vectors), evaluated in float64 on the fp32 inputs."""
    from oracle import ref_mirror
    g = torch.Generator().manual_seed(5000 + n + 3 * m + p)
    B, L = 2, 6
    x, y, U = unit_cloud(g, B, n), unit_cloud(g, B, m), directions(g, B, L)
    wu = wv = None
    if weighted:
        wu = torch.rand(n, generator=g) + 0.05
        wv = torch.rand(m, generator=g) + 0.05
        wu, wv = wu / wu.sum(), wv / wv.sum()
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, _ = shw.ssw_pair_losses(xs, ys, U.cuda(), p=p, return_slices=True,
                                        u_weights=None if wu is None else wu.cuda(),
                                        v_weights=None if wv is None else wv.cuda())
    wts = torch.tensor([1.0, -0.5], device="cuda")
    (pair * wts).sum().backward()
    for b in range(B):
        xd, yd = x[b].double().requires_grad_(True), y[b].double().requires_grad_(True)
        ref = ref_mirror.per_slice_costs(xd, yd, U[b].double(), p=p,
                                         u_weights=None if wu is None else wu.double(),
                                         v_weights=None if wv is None else wv.double())
        (ref.mean() * wts[b].item()).backward()
        assert np.allclose(cost[b].detach().cpu().numpy(), ref.detach().numpy(), rtol=5e-5, atol=1e-9)
        if min(n, m) >= 7:
            grad_close(xs.grad[b].cpu().numpy(), xd.grad.numpy(), exact=(max(n, m) <= 128))
            grad_close(ys.grad[b].cpu().numpy(), yd.grad.numpy(), exact=(max(n, m) <= 128))


def test_general_solver_reduces_to_fast_path_on_equal_uniform_input(shw):
    """n == m with explicitly uniform weights goes through the general solver and must agree with the
    equal-size kernel (two different algorithms, same minimum)."""
    g = torch.Generator().manual_seed(91)
    x, y, U = unit_cloud(g, 2, 200).cuda(), unit_cloud(g, 2, 200).cuda(), directions(g, 2, 8).cuda()
    w = torch.full((200,), 1.0 / 200, device="cuda")
    fast = shw.ssw_pair_losses(x, y, U, p=2)
    gen = shw.ssw_pair_losses(x, y, U, p=2, u_weights=w, v_weights=w)
    assert torch.allclose(fast, gen, rtol=2e-5)


# ------------------------------------------------------------------------------ golden: G6 headline shapes
@pytest.mark.parametrize("tag", ["c2", "c3"])
def test_g6_headline_shapes(shw, golden, tag):
    g = golden("g6_headline_shapes.npz")
    x, y, U = dev(g[f"x_{tag}"]), dev(g[f"y_{tag}"]), dev(g[f"U_{tag}"])
    for p in (1, 2):
        pair, cost, _ = shw.ssw_pair_losses(x, y, U, p=p, return_slices=True)
        assert np.allclose(cost.cpu().numpy(), g[f"per_slice_{tag}_p{p}"], rtol=2e-5, atol=1e-10)
    assert rel(shw.sliced_cost(x, y, U, p=2).item(), g[f"value_{tag}_p2"]) < 1e-5


# ------------------------------------------------------------------------------ RNG stream parity (G5 on device)
def test_direction_sampling_consumes_generator_like_reference(shw, golden):
    """Same generator consumption as the reference (:307, _fast.py:317); the frames come from the HIP
    Householder kernel and must equal LAPACK's reduced QR of the same Gaussian matrices (sign convention
    included) to fp32 rounding."""
    torch.manual_seed(777)
    Z = torch.randn((24, 3, 2), device="cuda")                                # reference :307
    torch.manual_seed(777)
    got = shw.draw_directions(24, "cuda")
    expect = torch.linalg.qr(Z.cpu().double())[0]
    assert np.abs(got.cpu().double().numpy() - expect.numpy()).max() < 1e-6
    gtg = torch.einsum("ldk,ldj->lkj", got, got).cpu()
    assert np.abs(gtg.numpy() - np.eye(2)).max() < 1e-6                     # orthonormal columns
    # fixture G5 (CPU generator): the kernel reproduces the frames the reference drew from the same Gaussians
    g = golden("g5_rng.npz")
    torch.manual_seed(int(g["seed"]))
    Zc = torch.randn((24, 3, 2))
    assert np.abs(shw.stiefel_frames(Zc.cuda()).cpu().numpy() - g["U_pair"]).max() < 1e-6
    gen = torch.Generator().manual_seed(5)
    x, y = unit_cloud(gen, 128).cuda(), unit_cloud(gen, 128).cuda()
    torch.manual_seed(777)
    val = shw.sliced_wasserstein_sphere(x, y, 24, "cuda", p=2)
    assert rel(val.item(), shw.sliced_cost(x, y, got, p=2).item()) == 0.0
    xb, yb = unit_cloud(gen, 3, 64).cuda(), unit_cloud(gen, 3, 64).cuda()
    torch.manual_seed(778)
    expect_b = shw.stiefel_frames(torch.randn((3, 12, 3, 2), device="cuda"))     # _fast.py:317-318
    torch.manual_seed(778)
    valb = shw.sliced_wasserstein_sphere_fast(xb, yb, 12, "cuda", p=2)
    assert tuple(valb.shape) == (1,)
    assert rel(valb.item(), shw.sliced_cost(xb, yb, expect_b, p=2).item()) == 0.0
    # degenerate columns: zero matrix, first column already on e1 -> still finite frames
    Zd = torch.zeros(2, 3, 2, device="cuda")
    Zd[1, 0, 0] = 2.0
    Zd[1, 1, 1] = -3.0
    Ud = shw.stiefel_frames(Zd)
    assert torch.isfinite(Ud).all()
    assert np.abs(Ud[1].cpu().numpy() - torch.linalg.qr(Zd[1].cpu())[0].numpy()).max() < 1e-6


# ------------------------------------------------------------------------------ oracle on seeded inputs
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 100, 200, 257, 1000, 1024, 2048])
@pytest.mark.parametrize("p", [2, 3, 2.5])
def test_sizes_against_cpu_oracle(shw, n, p):
    from oracle import exact_shift, ref_mirror
    g = torch.Generator().manual_seed(1000 + n)
    L = 8 if n >= 1000 else 16
    x, y, U = unit_cloud(g, n), unit_cloud(g, n), directions(g, L)
    pair, cost, shift = shw.ssw_pair_losses(x.cuda().unsqueeze(0), y.cuda().unsqueeze(0), U.cuda(), p=p,
                                            return_slices=True)
    ref64, k64 = exact_shift.circular_ot_equal(exact_shift.circle_coords(x.numpy(), U.numpy()),
                                               exact_shift.circle_coords(y.numpy(), U.numpy()), p=p)
    tol = 2e-5 if p == 2 else 4e-5
    assert np.allclose(cost[0].cpu().numpy(), ref64, rtol=tol, atol=1e-9)
    if n <= 257:   # the torch restatement of the reference's bisection, fp32
        mirror = ref_mirror.per_slice_costs(x, y, U, p=p).numpy()
        assert np.allclose(cost[0].cpu().numpy(), mirror, rtol=5e-5, atol=1e-9)


@pytest.mark.parametrize("n", [3000, 4096, 5000, 8192])
def test_large_sizes_against_cpu_oracle(shw, n):
    """Size classes above the headline one (64 and 128 keys per lane), loss and gradients."""
    from oracle import exact_shift
    g = torch.Generator().manual_seed(4000 + n)
    L = 3
    x, y, U = unit_cloud(g, n), unit_cloud(g, n), directions(g, L)
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, _ = shw.ssw_pair_losses(xs.unsqueeze(0), ys.unsqueeze(0), U.cuda(), p=2, return_slices=True)
    pair.sum().backward()
    cu = exact_shift.circle_coords(x.numpy(), U.numpy())
    cv = exact_shift.circle_coords(y.numpy(), U.numpy())
    ref64, _ = exact_shift.circular_ot_equal(cu, cv, p=2)
    assert np.allclose(cost[0].detach().cpu().numpy(), ref64, rtol=3e-5, atol=1e-10)
    nograd = shw.ssw_pair_losses(x.cuda().unsqueeze(0), y.cuda().unsqueeze(0), U.cuda(), p=2)
    assert abs(nograd.item() - pair.item()) < 1e-6 * pair.item()      # loss-only and training kernels agree
    gx, gy = exact_shift.ssw_pair_grad(x.numpy(), y.numpy(), U.numpy(), p=2)
    grad_close(xs.grad.cpu().numpy(), gx)
    grad_close(ys.grad.cpu().numpy(), gy)
    ref1 = np.array([exact_shift.w1_level_median(cu[l], cv[l]) for l in range(L)])
    _, cost1, _ = shw.ssw_pair_losses(x.cuda().unsqueeze(0), y.cuda().unsqueeze(0), U.cuda(), p=1, return_slices=True)
    assert np.allclose(cost1[0].cpu().numpy(), ref1, rtol=5e-5, atol=1e-10)


def test_sizes_beyond_the_limit_are_rejected(shw):
    x = torch.zeros(1, 8193, 3, device="cuda")
    U = torch.zeros(1, 2, 3, 2, device="cuda")
    with pytest.raises(RuntimeError):
        shw.ssw_pair_losses(x, x, U, p=2)


@pytest.mark.parametrize("n,m", [(1, 1), (2, 3), (63, 65), (64, 64), (100, 100), (200, 256), (1000, 777),
                                 (1024, 1024), (2048, 2048)])
def test_sizes_p1_against_cpu_oracle(shw, n, m):
    from oracle import exact_shift
    g = torch.Generator().manual_seed(3000 + n + m)
    L = 8
    x, y, U = unit_cloud(g, n), unit_cloud(g, m), directions(g, L)
    _, cost, _ = shw.ssw_pair_losses(x.cuda().unsqueeze(0), y.cuda().unsqueeze(0), U.cuda(), p=1, return_slices=True)
    cu = exact_shift.circle_coords(x.numpy(), U.numpy())
    cv = exact_shift.circle_coords(y.numpy(), U.numpy())
    ref = np.array([exact_shift.w1_level_median(cu[l], cv[l]) for l in range(L)])
    assert np.allclose(cost[0].cpu().numpy(), ref, rtol=3e-5, atol=1e-9)


def test_batched_p1_is_pairwise_extension(shw):
    """The reference's batched p == 1 raises; here it is the sum of the per-pair p == 1 values."""
    g = torch.Generator().manual_seed(55)
    x, y, U = unit_cloud(g, 3, 128).cuda(), unit_cloud(g, 3, 128).cuda(), directions(g, 3, 16).cuda()
    val = shw.sliced_cost(x, y, U, p=1)
    each = sum(shw.sliced_cost(x[b], y[b], U[b], p=1).item() for b in range(3))
    assert tuple(val.shape) == (1,) and abs(val.item() - each) < 1e-6 * each


@pytest.mark.parametrize("n", [64, 100, 256, 1000])
@pytest.mark.parametrize("p", [2, 3])
def test_gradients_against_cpu_oracle(shw, n, p):
    from oracle import exact_shift
    g = torch.Generator().manual_seed(2000 + n)
    L = 12
    x, y, U = unit_cloud(g, 2, n), unit_cloud(g, 2, n), directions(g, 2, L)
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    w = torch.tensor([0.7, -1.3], device="cuda")
    (shw.ssw_pair_losses(xs, ys, U.cuda(), p=p) * w).sum().backward()
    for b in range(2):
        gx, gy = exact_shift.ssw_pair_grad(x[b].numpy(), y[b].numpy(), U[b].numpy(), p=p)
        gx, gy = gx * w[b].item(), gy * w[b].item()
        grad_close(xs.grad[b].cpu().numpy(), gx, exact=(n <= 256))
        grad_close(ys.grad[b].cpu().numpy(), gy, exact=(n <= 256))


def test_shared_directions_equal_per_pair_directions(shw):
    g = torch.Generator().manual_seed(31)
    x, y, U = unit_cloud(g, 3, 256).cuda(), unit_cloud(g, 3, 256).cuda(), directions(g, 16).cuda()
    a = shw.ssw_pair_losses(x, y, U, p=2)
    b = shw.ssw_pair_losses(x, y, U.unsqueeze(0).expand(3, -1, -1, -1).contiguous(), p=2)
    assert torch.equal(a, b)


def test_duplicate_points_and_ties(shw):
    """Clouds padded by repeating points: ties in every slice; value must match the oracle and the
    gradient of duplicated points must follow the stable (original index) order of torch.sort."""
    from oracle import exact_shift
    g = torch.Generator().manual_seed(77)
    base_x, base_y = unit_cloud(g, 96), unit_cloud(g, 96)
    x = torch.cat([base_x, base_x[:32]], 0)
    y = torch.cat([base_y, base_y[:32]], 0)
    U = directions(g, 8)
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    val = shw.sliced_cost(xs, ys, U.cuda(), p=2)
    val.backward()
    assert abs(val.item() - exact_shift.ssw_pair(x.numpy(), y.numpy(), U.numpy(), 2)) < 1e-5 * val.item()
    gx, gy = exact_shift.ssw_pair_grad(x.numpy(), y.numpy(), U.numpy(), 2)
    grad_close(xs.grad.cpu().numpy(), gx, exact=True)
    grad_close(ys.grad.cpu().numpy(), gy, exact=True)


def test_rejects_what_is_not_implemented_instead_of_falling_back(shw):
    x = torch.zeros(8, 3, device="cuda")
    U = torch.zeros(2, 3, 2, device="cuda")
    with pytest.raises(RuntimeError):
        shw.sliced_cost(x.cpu(), x.cpu(), U.cpu())
    with pytest.raises(ValueError):
        shw.sliced_cost(x, x, U, p=1, u_weights=torch.ones(7, device="cuda") / 7)      # wrong length
    with pytest.raises(TypeError):
        shw.sliced_cost(x.double(), x.double(), U.double())


# ------------------------------------------------------------------------------ full BASELINE size: properties
def test_full_size_properties_config3(shw):
    B, N, L = 64, 2048, 512
    g = torch.Generator().manual_seed(1234)
    x = unit_cloud(g, B, N).cuda()
    y = unit_cloud(g, B, N).cuda()
    U = directions(g, B, L).cuda()
    pair, cost, shift = shw.ssw_pair_losses(x, y, U, p=2, return_slices=True)
    assert torch.isfinite(pair).all() and (cost >= 0).all()
    # (1) the per-pair value is the mean of its slices
    assert torch.allclose(pair, cost.double().mean(1).float(), rtol=2e-6)
    # (2) symmetry W(u,v) = W(v,u), optimal shifts negate
    pair_t, cost_t, shift_t = shw.ssw_pair_losses(y, x, U, p=2, return_slices=True)
    assert torch.allclose(cost, cost_t, rtol=2e-5, atol=1e-10)
    # (3) identical clouds cost exactly zero in every slice
    _, cost0, shift0 = shw.ssw_pair_losses(x, x.clone(), U, p=2, return_slices=True)
    assert float(cost0.abs().max()) == 0.0 and int(shift0.abs().max()) == 0
    # (4) rotating both clouds AND the frames by the same rotation changes nothing (up to rounding)
    R = torch.linalg.qr(torch.randn(3, 3, generator=g))[0].cuda()
    pair_r = shw.ssw_pair_losses(x @ R.T, y @ R.T, torch.einsum("ij,bljk->blik", R, U).contiguous(), p=2)
    assert torch.allclose(pair, pair_r, rtol=1e-4)
    # (5) slice sharding: two halves of the slices average to the whole (the multi-GPU decomposition)
    h = L // 2
    pa = shw.ssw_pair_losses(x, y, U[:, :h].contiguous(), p=2)
    pb = shw.ssw_pair_losses(x, y, U[:, h:].contiguous(), p=2)
    assert torch.allclose(pair, 0.5 * (pa + pb), rtol=2e-6)
    # (6) batched entry = sum over pairs
    assert abs(shw.sliced_cost(x, y, U, p=2).item() - pair.double().sum().item()) < 1e-5 * pair.sum().item()
    # (7) a sample of slices against the float64 exhaustive-shift oracle at full N
    from oracle import exact_shift
    for b, l in [(0, 0), (17, 300), (63, 511)]:
        cu = exact_shift.circle_coords(x[b].cpu().numpy(), U[b, l:l + 1].cpu().numpy())
        cv = exact_shift.circle_coords(y[b].cpu().numpy(), U[b, l:l + 1].cpu().numpy())
        c64, k64 = exact_shift.circular_ot_equal(cu, cv, 2)
        assert abs(cost[b, l].item() - c64[0]) < 2e-5 * c64[0]


def test_full_size_gradient_is_consistent_config3_subset(shw):
    """Gradient at N=2048: directional derivative check against a central finite difference of the HIP
    loss itself along a smooth random direction (size-independent property)."""
    B, N, L = 4, 2048, 128
    g = torch.Generator().manual_seed(4321)
    x, y, U = unit_cloud(g, B, N).cuda(), unit_cloud(g, B, N).cuda(), directions(g, B, L).cuda()
    xs = x.clone().requires_grad_(True)
    shw.ssw_pair_losses(xs, y, U, p=2).sum().backward()
    d = torch.randn(B, N, 3, generator=g).cuda()
    eps = 1e-3
    fp = shw.ssw_pair_losses(x + eps * d, y, U, p=2).double().sum()
    fm = shw.ssw_pair_losses(x - eps * d, y, U, p=2).double().sum()
    fd = ((fp - fm) / (2 * eps)).item()
    an = (xs.grad.double() * d.double()).sum().item()
    assert abs(fd - an) < 2e-2 * abs(an) + 1e-7


# ------------------------------------------------------------------------------ Chamfer
@pytest.mark.parametrize("n,m", [(1, 1), (7, 13), (256, 256), (1000, 300), (2048, 2048)])
def test_chamfer_against_numpy_oracle(shw, n, m):
    from oracle import exact_shift
    g = torch.Generator().manual_seed(9 + n + m)
    B = 3
    x, y = torch.randn(B, n, 3, generator=g), torch.randn(B, m, 3, generator=g) * 0.8 + 0.1
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    loss, extra = shw.chamfer_distance(xs, ys)
    assert extra is None
    ref = exact_shift.chamfer(x.numpy(), y.numpy(), "mean")
    assert abs(loss.item() - ref) < 1e-5 * abs(ref)
    assert abs(shw.chamfer_distance(xs, ys, batch_reduction="sum")[0].item() - exact_shift.chamfer(x.numpy(), y.numpy(), "sum")) < 1e-5 * abs(ref) * B
    loss.backward()
    # autograd of the same definition in float64 on the CPU
    xd, yd = x.double().requires_grad_(True), y.double().requires_grad_(True)
    dmat = torch.cdist(xd, yd) ** 2
    (dmat.min(2).values.mean(1) + dmat.min(1).values.mean(1)).mean().backward()
    assert np.abs(xs.grad.cpu().numpy() - xd.grad.numpy()).max() < 1e-4 * np.abs(xd.grad.numpy()).max()
    assert np.abs(ys.grad.cpu().numpy() - yd.grad.numpy()).max() < 1e-4 * np.abs(yd.grad.numpy()).max()


# ------------------------------------------------------------------------------ module-level call shapes (8b)
def test_csw_slot_module_and_criteria(shw):
    g = torch.Generator().manual_seed(12)
    x, y = unit_cloud(g, 4, 256).cuda().requires_grad_(True), unit_cloud(g, 4, 256).cuda()
    csw = shw.SlicedSphereW("cuda", p=2, num_projections=32)
    torch.manual_seed(3)
    val = csw(x, y)
    assert val.dim() == 0
    torch.manual_seed(3)
    U = shw.draw_directions(32, "cuda", batch=4)
    expect = shw.ssw_pair_losses(x.detach(), y, U, 2).sqrt().mean()
    assert abs(val.item() - expect.item()) < 1e-6 * expect.item()
    val.backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().max() > 0
    single = csw(x[0].detach(), y[0])
    assert single.dim() == 0
    loss, a, b = shw.SSWCriterion("cuda", 2, 16)(x.detach(), y, train_or_test="test")
    assert loss.dim() == 0 and a.shape == x.shape and b.shape == y.shape
    cd = shw.ChamferCriterion()(x.detach(), y)[0]
    assert cd.dim() == 0


def test_phi_max_wrappers_run_and_ascend(shw):
    g = torch.Generator().manual_seed(13)
    x, y = torch.randn(4, 128, 3, generator=g).cuda(), torch.randn(4, 128, 3, generator=g).cuda()

    class Sphere(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3, 3)

        def forward(self, t):
            return torch.nn.functional.normalize(self.lin(t), dim=-1)

    torch.manual_seed(0)
    for cls, fn in ((shw.max_spherical_wassersten_distance, shw.sliced_wasserstein_sphere),
                    (shw.max_spherical_wassersten_distance_fast, shw.sliced_wasserstein_sphere_fast)):
        phi = Sphere().cuda()
        op = torch.optim.Adam(phi.parameters(), lr=1e-2)
        wrap = cls(16, phi, fn, op, p=2, max_iter=3, device="cuda")
        before = [q.detach().clone() for q in phi.parameters()]
        ssw, a, b = wrap(x, y, train_or_test="train")
        assert torch.isfinite(ssw).all() and a.shape == x.shape and b.shape == y.shape
        assert any((q.detach() - q0).abs().max() > 0 for q, q0 in zip(phi.parameters(), before))
        ssw_t, _, _ = wrap(x, y, train_or_test="test")
        assert torch.isfinite(ssw_t).all()


def test_reduce_paths_agree_and_match_torch(shw):
    """C-ABI reduction: the fused (pairs <= 256) and the two-kernel (pairs > 256) forms use the same summation
    order; both agree with a float64 sum to fp32 rounding."""
    lib = shw._lib.load()
    g = torch.Generator().manual_seed(8)
    for pairs, slices in ((1, 1), (3, 70), (64, 512), (256, 33), (300, 129)):
        cost = torch.rand(pairs, slices, generator=g).cuda()
        pl = torch.empty(pairs, device="cuda")
        tot = torch.empty(2, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        shw._lib.check(lib.shw_ssw_reduce(cost.data_ptr(), pairs, slices, 1.0 / slices, pl.data_ptr(),
                                          tot.data_ptr(), st), "reduce")
        ref = cost.double().mean(1)
        assert torch.allclose(pl.double(), ref, rtol=1e-6)
        assert abs(tot[0].item() - ref.sum().item()) < 2e-6 * ref.sum().item()
        assert abs(tot[1].item() - ref.mean().item()) < 2e-6 * ref.mean().item()


# ------------------------------------------------------------------------------ Euclidean sliced-W (notebook SWD)
@pytest.mark.parametrize("n", [1, 50, 64, 1200, 2048, 4096])
@pytest.mark.parametrize("p", [1, 2, 3])
def test_euclidean_sliced_w_against_restatement(shw, n, p):
    from oracle import euclid_sw
    g = torch.Generator().manual_seed(600 + n + p)
    L = 10
    a, b = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g) * 0.7 + 0.2
    th = shw.rand_projections(3, L)
    xa, xb = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    sums = shw.esw_slice_sums(xa.unsqueeze(0), xb.unsqueeze(0), th.cuda(), p)
    ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = euclid_sw.slice_sums(ad, bd, th.double(), p)
    assert np.allclose(sums[0].detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=1e-9)
    w = torch.rand(L, generator=g) + 0.5
    (sums[0] * w.cuda()).sum().backward()
    (ref * w.double()).sum().backward()
    if p > 1 or n > 1:
        grad_close(xa.grad.cpu().numpy(), ad.grad.numpy(), strict=5e-4)
        grad_close(xb.grad.cpu().numpy(), bd.grad.numpy(), strict=5e-4)


def test_euclidean_sliced_w_call_shape_and_rng(shw):
    g = torch.Generator().manual_seed(7)
    a, b = torch.randn(300, 3, generator=g).cuda(), torch.randn(300, 3, generator=g).cuda()
    torch.manual_seed(11)
    val = shw.sliced_wasserstein_distance(a, b, num_projection=50, p=2, device="cuda")
    torch.manual_seed(11)
    th = shw.rand_projections(3, 50)              # CPU generator, like the notebook cell
    from oracle import euclid_sw
    ref = euclid_sw.sliced_wasserstein_distance(a.cpu().double(), b.cpu().double(), th.double(), 2)
    assert val.dim() == 0 and abs(val.item() - ref.item()) < 1e-5 * ref.item()


# ------------------------------------------------------------------------------ trainer-level drop-in (config 5 shape)
def _config5_module():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "config5_train_step.py")
    spec = importlib.util.spec_from_file_location("config5_train_step", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_config5_training_steps_reduce_the_loss(shw):
    """small shape, many steps: the loop trains (loss goes down) with every criterion variant"""
    mod = _config5_module()
    for criterion in ("plain", "csw", "ssw_fast"):
        losses, _ = mod.run(batch=8, points=1024, slices=128, steps=25, verbose=False, criterion=criterion)
        assert all(np.isfinite(losses)), criterion
        assert min(losses[-5:]) < losses[0], (criterion, losses)


def test_config5_at_its_stated_shape(shw, capsys):
    """BASELINE configs[4] at the stated size: PCRNet-shaped regressor (emb 1024, 5 FC, 8 refinement iterations),
    B=32, N=2048, L=512, phi-max criterion with the sliced loss in the CSW slot, forward + backward + Adam on one
    GPU.  Finite, the loss decreases over the steps, and the sliced-loss share of the step time is printed."""
    mod = _config5_module()
    losses, times = mod.run(batch=32, points=2048, slices=512, steps=12, verbose=False, criterion="csw",
                            phi_max_iter=1, iteration_num=8)
    assert all(np.isfinite(losses))
    assert min(losses[-4:]) < losses[0], losses
    med = sorted(times[2:])[len(times[2:]) // 2]
    share = mod.ssw_share(32, 2048, 512, evaluations=2)
    with capsys.disabled():
        print(f"\n[config5] B=32 N=2048 L=512 iters=8: median step {1e3 * med:.2f} ms, sliced-loss part "
              f"{1e3 * share:.2f} ms ({100 * share / med:.0f} %), loss {losses[0]:.5f} -> {losses[-1]:.5f}")
    assert share < med


# ------------------------------------------------------------------------------ hipGraph capture of the training step
def test_loss_and_backward_capture_into_a_hip_graph(shw):
    """Nothing on the path allocates through HIP, synchronises or reads back on the host, so forward + backward
    of the loss can be captured once and replayed on new data (static input buffers)."""
    g = torch.Generator().manual_seed(21)
    B, N, L = 4, 512, 64
    x_static = unit_cloud(g, B, N).cuda().requires_grad_(True)
    y_static = unit_cloud(g, B, N).cuda()
    U_static = shw.stiefel_frames(torch.randn(B, L, 3, 2, generator=g).cuda())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up on a side stream, as graph capture requires
        for _ in range(2):
            x_static.grad = None
            shw.sliced_cost(x_static, y_static, U_static, p=2).backward()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    x_static.grad = None
    with torch.cuda.graph(graph):
        loss_static = shw.sliced_cost(x_static, y_static, U_static, p=2)
        loss_static.backward()
    for seed in (1, 2):
        h = torch.Generator().manual_seed(seed)
        xn, yn = unit_cloud(h, B, N).cuda(), unit_cloud(h, B, N).cuda()
        with torch.no_grad():
            x_static.copy_(xn)
            y_static.copy_(yn)
        graph.replay()
        torch.cuda.synchronize()
        xe = xn.clone().requires_grad_(True)
        le = shw.sliced_cost(xe, yn, U_static, p=2)
        le.backward()
        assert torch.equal(loss_static, le)
        assert torch.equal(x_static.grad, xe.grad)


# ------------------------------------------------------------------------------ API robustness
def test_noncontiguous_inputs_partial_grads_and_shared_dirs(shw):
    from oracle import exact_shift
    g = torch.Generator().manual_seed(33)
    B, n, L = 3, 96, 10
    big = unit_cloud(g, B, 2 * n).cuda()
    x = big[:, ::2, :]                                   # non-contiguous view
    y = unit_cloud(g, B, n).cuda().requires_grad_(True)   # only the target requires grad
    U = directions(g, L).cuda()                           # shared across pairs
    assert not x.is_contiguous()
    pair = shw.ssw_pair_losses(x, y, U, p=2)
    pair.sum().backward()
    assert y.grad is not None and torch.isfinite(y.grad).all()
    for b in range(B):
        ref = exact_shift.ssw_pair(x[b].cpu().numpy(), y[b].detach().cpu().numpy(), U.cpu().numpy(), 2)
        assert abs(pair[b].item() - ref) < 1e-5 * ref
        _, gy = exact_shift.ssw_pair_grad(x[b].cpu().numpy(), y[b].detach().cpu().numpy(), U.cpu().numpy(), 2)
        grad_close(y.grad[b].cpu().numpy(), gy, exact=True)


def test_no_grad_mode_single_slice_single_pair_and_fractional_power(shw):
    from oracle import exact_shift
    g = torch.Generator().manual_seed(34)
    x, y, U = unit_cloud(g, 1, 300).cuda().requires_grad_(True), unit_cloud(g, 1, 300).cuda(), directions(g, 1, 1).cuda()
    with torch.no_grad():
        v = shw.sliced_cost(x, y, U, p=1.5)
    assert not v.requires_grad and tuple(v.shape) == (1,)
    ref = exact_shift.ssw_pair(x[0].detach().cpu().numpy(), y[0].cpu().numpy(), U[0].cpu().numpy(), 1.5)
    assert abs(v.item() - ref) < 2e-5 * ref
    with pytest.raises(ValueError):
        shw.sliced_cost(x, y, U, p=0.5)
    with pytest.raises(ValueError):
        shw.ssw_pair_losses(x, y, U[:, :0], p=2)


def test_many_slices_few_points_and_many_pairs(shw):
    g = torch.Generator().manual_seed(35)
    x, y, U = unit_cloud(g, 2, 16).cuda(), unit_cloud(g, 2, 16).cuda(), directions(g, 2, 4096).cuda()
    pair, cost, _ = shw.ssw_pair_losses(x, y, U, p=2, return_slices=True)
    assert torch.allclose(pair, cost.double().mean(1).float(), rtol=2e-6)
    xb, yb, Ub = unit_cloud(g, 700, 64).cuda(), unit_cloud(g, 700, 64).cuda(), directions(g, 4).cuda()
    pb, cb, _ = shw.ssw_pair_losses(xb, yb, Ub, p=2, return_slices=True)       # > 256 pairs: two-kernel reduction
    assert torch.allclose(pb, cb.double().mean(1).float(), rtol=2e-6)
    assert abs(shw.sliced_cost(xb, yb, Ub.unsqueeze(0).expand(700, -1, -1, -1).contiguous(), p=2).item()
               - pb.double().sum().item()) < 1e-5 * pb.sum().item()


def test_results_are_deterministic_run_to_run(shw):
    g = torch.Generator().manual_seed(36)
    x, y, U = unit_cloud(g, 8, 1000).cuda(), unit_cloud(g, 8, 1000).cuda(), directions(g, 8, 64).cuda()
    outs = []
    for _ in range(3):
        xs = x.clone().requires_grad_(True)
        v = shw.sliced_cost(xs, y, U, p=2)
        v.backward()
        outs.append((v.clone(), xs.grad.clone()))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])


# ------------------------------------------------------------------------------ the C ABI without Python
def test_standalone_c_consumer_of_the_abi(shw, tmp_path):
    """Compiles tests/capi/standalone.cpp with hipcc against include/shw.h + libshw_hip.so (system HIP runtime,
    no torch in the process) and compares its output with the Python mirror on the same inputs."""
    import os
    import shutil
    import struct
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(shw._lib.LIB_PATH)
    exe = str(tmp_path / "standalone")
    subprocess.run([hipcc, "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "capi", "standalone.cpp"),
                    "-L", libdir, "-lshw_hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True, capture_output=True)
    g = torch.Generator().manual_seed(404)
    B, n, L = 3, 500, 20
    x, y, U = unit_cloud(g, B, n), unit_cloud(g, B, n), directions(g, B, L)
    blob = str(tmp_path / "in.bin")
    with open(blob, "wb") as fh:
        fh.write(struct.pack("iii", B, n, L))
        for t in (x, y, U):
            fh.write(t.contiguous().numpy().astype(np.float32).tobytes())
    out = subprocess.run([exe, blob], check=True, capture_output=True, text=True).stdout.split("\n")
    pair = shw.ssw_pair_losses(x.cuda(), y.cuda(), U.cuda(), p=2).cpu().numpy()
    cham = shw.chamfer_pair_losses(x.cuda(), y.cuda()).cpu().numpy()
    tot = [float(v) for v in out[0].split()[1:]]
    assert abs(tot[0] - pair.astype(np.float64).sum()) < 2e-6 * pair.sum()
    for b in range(B):
        _, _, p_c, c_c = out[1 + b].split()
        assert float(p_c) == pytest.approx(float(pair[b]), rel=1e-7)
        assert float(c_c) == pytest.approx(float(cham[b]), rel=1e-7)


# ------------------------------------------------------------------------------ log-domain Sinkhorn (forward)
@pytest.mark.parametrize("eps,iters", [(0.05, 60), (0.01, 100)])
def test_g7_sinkhorn_against_reference_fixture(shw, golden, eps, iters):
    """eps divides every exponent, so fp32 rounding of the duals is amplified by 1/eps in the plan: the reference's
    own fp32-vs-fp64 difference is ~1e-5 at eps = 0.05 and ~1e-4 at eps = 0.01; tolerance 5e-4 on the cost."""
    g = golden("g7_sinkhorn.npz")
    crit = shw.log_Sinkhorn_Distance_Loss(eps=eps, max_iter=iters, batch_reduction="none", type_of_cost_norm="L2")
    cost, P, C = crit(dev(g["x"]), dev(g["y"]), "cuda")
    tag = f"eps{eps}_it{iters}"
    assert tuple(P.shape) == (2, 96, 80) and tuple(C.shape) == (2, 96, 80)
    assert rel(cost.cpu().numpy(), g[f"cost_{tag}"]) < 5e-4
    assert np.allclose(P.sum(-1).cpu().numpy(), g[f"P_rowsum_{tag}"], rtol=2e-3, atol=1e-6)
    assert np.allclose(P.sum(-2).cpu().numpy(), g[f"P_colsum_{tag}"], rtol=2e-3, atol=1e-6)
    assert np.allclose(C[0, 0].cpu().numpy(), g["C_first_row"], rtol=1e-6)
    assert abs((P * C).sum((-2, -1))[0].item() - cost[0].item()) < 1e-5 * cost[0].item()


def test_g7_sinkhorn_variants_and_call_shapes(shw, golden):
    g = golden("g7_sinkhorn.npz")
    x, y = dev(g["x"]), dev(g["y"])
    l1 = shw.log_Sinkhorn_Distance_Loss(0.05, 60, batch_reduction="sum", type_of_cost_norm="L1", return_plan=False)
    val, P, C = l1(x, y, "cuda")
    assert P is None and C is None and val.dim() == 0
    assert rel(val.item(), g["cost_L1_sum"]) < 5e-4
    n2 = shw.log_N_Sinkhorn_Distance_Loss(0.05, 60, batch_reduction="mean", type_of_cost_norm="L2",
                                          type_of_Wasserstein_N="2", return_plan=False)
    assert rel(n2(x, y, "cuda")[0].item(), g["cost_N2_mean"]) < 5e-4
    single = shw.log_Sinkhorn_Distance_Loss(0.05, 10)(x[0], y[0], "cuda")
    assert single[0].dim() == 0 and tuple(single[1].shape) == (96, 80)
    # round 2: differentiable like the reference (gradient parity: tests/test_r2_gpu.py, fixture G7b)
    xr = x.clone().requires_grad_(True)
    shw.log_Sinkhorn_Distance_Loss(0.05, 10, batch_reduction="sum")(xr, y, "cuda")[0].backward()
    assert torch.isfinite(xr.grad).all() and xr.grad.abs().max() > 0


@pytest.mark.parametrize("n,m", [(1, 1), (5, 700), (300, 257), (1024, 1024)])
def test_sinkhorn_sizes_against_restatement(shw, n, m):
    from oracle import sinkhorn_mirror
    g = torch.Generator().manual_seed(70 + n + m)
    B = 2
    x, y = unit_cloud(g, B, n), unit_cloud(g, B, m) * 0.9 + 0.05
    iters = 30
    cost, _, _ = shw.sinkhorn_pair_costs(x.cuda(), y.cuda(), 0.05, iters)
    ref, _, _, _ = sinkhorn_mirror.sinkhorn_costs(x.double(), y.double(), 0.05, iters)
    assert rel(cost.cpu().numpy(), ref.numpy()) < 5e-4


def test_sinkhorn_convergence_flag_stops_the_iteration(shw):
    """Identical tiny clouds converge immediately; with a loose threshold the device-side flag must freeze the
    duals exactly where the reference's `break` would (same result as running fewer iterations)."""
    g = torch.Generator().manual_seed(3)
    x = unit_cloud(g, 2, 32).cuda()
    from oracle import sinkhorn_mirror
    ref, _, _, its = sinkhorn_mirror.sinkhorn_costs(x.cpu().double(), x.cpu().double(), 0.5, 200, thresh=1e-3)
    assert its < 200
    got, _, _ = shw.sinkhorn_pair_costs(x, x.clone(), 0.5, 200, thresh=1e-3)
    assert rel(got.cpu().numpy(), ref.numpy()) < 5e-4


def test_euclidean_sliced_w_direction_gradient_and_max_sw(shw):
    from oracle import euclid_sw
    g = torch.Generator().manual_seed(808)
    n, L = 400, 6
    a, b = torch.randn(2, n, 3, generator=g), torch.randn(2, n, 3, generator=g) * 0.6 + 0.3
    th = shw.rand_projections(3, L)
    tg = th.cuda().requires_grad_(True)
    sums = shw.esw_slice_sums(a.cuda(), b.cuda(), tg, 2)
    w = torch.rand(2, L, generator=g) + 0.5
    (sums * w.cuda()).sum().backward()
    td = th.double().requires_grad_(True)
    ref = torch.stack([euclid_sw.slice_sums(a[k].double(), b[k].double(), td, 2) for k in range(2)])
    (ref * w.double()).sum().backward()
    assert np.allclose(sums.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5)
    grad_close(tg.grad.cpu().numpy(), td.grad.numpy(), strict=1e-3, exact=True)
    # max-sliced-W: the same Adam ascent (notebook cell :294-323) run on the CPU restatement in float64 from the
    # same starting direction must land on the same value
    x, y = a[0].cuda(), b[0].cuda()
    torch.manual_seed(5)
    d1 = shw.max_sliced_wasserstein_distance(x, y, p=2, max_iter=25, device="cuda").item()
    torch.manual_seed(5)
    proj = shw.rand_projections(3, 1).double().requires_grad_(True)
    opt = torch.optim.Adam([proj], lr=0.005, betas=(0.999, 0.999))
    for _ in range(25):
        d = torch.pow(euclid_sw.slice_sums(a[0].double(), b[0].double(), proj, 2).mean(), 0.5)
        opt.zero_grad()
        (-d).backward()
        opt.step()
        proj.data = proj.data / torch.sqrt(torch.sum(proj.data ** 2, dim=1))
    ref_d = torch.pow(euclid_sw.slice_sums(a[0].double(), b[0].double(), proj.detach(), 2).mean(), 0.5).item()
    assert abs(d1 - ref_d) < 1e-3 * ref_d


# ------------------------------------------------------------------------------ bench.py contract
def _run_bench(extra, torchrun=False):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable]
    if torchrun:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29533"]
    cmd += [os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"] + extra
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, cwd=root).stdout.strip().split("\n")
    assert len(out) == 1, out                      # exactly ONE line on stdout
    return json.loads(out[0])


def test_bench_prints_one_json_line_with_the_contract_keys():
    d = _run_bench(["--cpu-sample-pairs", "1"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["value"] > 1e3 * c["value"]          # sanity: orders of magnitude above the CPU restatement


def test_bench_under_torchrun_uses_rccl_and_still_prints_one_line():
    d = _run_bench(["--no-cpu-baseline"], torchrun=True)
    assert d["n_gpus"] == 1 and "roofline" in d and "cpu_baseline" not in d


def test_dist_module_over_rccl_with_the_hip_evaluator():
    """dist.py with its default (HIP) local evaluator over the nccl backend, world size 1 (one GPU per box; the
    world_size-2 logic is covered by tests/test_dist_gloo.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29544", os.path.join(root, "tests", "helpers", "dist_nccl_worker.py")]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=root)
    assert res.returncode == 0 and "DIST_NCCL_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.parametrize("n,m", [(64, 64), (100, 37), (300, 512), (1000, 1024), (3, 1)])
def test_weighted_level_median_against_cpu_oracle(shw, n, m):
    """p == 1 with weights (and unequal sizes): loss and gradients against the torch restatement of emd1D_circle
    in float64; also the per-pair (B, n) weight layout and agreement with the uniform kernel on uniform weights."""
    from oracle import ref_mirror
    g = torch.Generator().manual_seed(8000 + n + 7 * m)
    B, L = 2, 6
    x, y, U = unit_cloud(g, B, n), unit_cloud(g, B, m), directions(g, B, L)
    wu = torch.rand(B, n, generator=g) + 0.05
    wv = torch.rand(m, generator=g) + 0.05
    wu, wv = wu / wu.sum(1, keepdim=True), wv / wv.sum()
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, _ = shw.ssw_pair_losses(xs, ys, U.cuda(), p=1, return_slices=True, u_weights=wu.cuda(), v_weights=wv.cuda())
    wts = torch.tensor([0.8, -1.1], device="cuda")
    (pair * wts).sum().backward()
    for b in range(B):
        xd, yd = x[b].double().requires_grad_(True), y[b].double().requires_grad_(True)
        ref = ref_mirror.per_slice_costs(xd, yd, U[b].double(), p=1, u_weights=wu[b].double(), v_weights=wv.double())
        (ref.mean() * wts[b].item()).backward()
        assert np.allclose(cost[b].detach().cpu().numpy(), ref.detach().numpy(), rtol=5e-5, atol=1e-9)
        if min(n, m) >= 30:
            # p = 1 coefficients are piecewise CONSTANT (+-weight): a near-tie that is ordered differently in fp32
            # and fp64 next to the median level flips one entry by its full size, hence the wide per-entry bound;
            # all but 0.5 % of the entries must still agree to 2e-4 of the largest
            grad_close(xs.grad[b].cpu().numpy(), xd.grad.numpy(), loose=0.2, exact=(max(n, m) <= 128))
            grad_close(ys.grad[b].cpu().numpy(), yd.grad.numpy(), loose=0.2, exact=(max(n, m) <= 128))
    if n == m:
        un = torch.full((n,), 1.0 / n, device="cuda")
        a = shw.ssw_pair_losses(x.cuda(), y.cuda(), U.cuda(), p=1)
        bb = shw.ssw_pair_losses(x.cuda(), y.cuda(), U.cuda(), p=1, u_weights=un, v_weights=un)
        assert torch.allclose(a, bb, rtol=2e-5)


# ------------------------------------------------------------------------------ adversarial inputs vs the float64 oracle
def _adversarial_clouds(kind, n, gen):
    if kind == "clustered":            # two tight clusters: thousands of near-equal coordinates per slice
        c = torch.nn.functional.normalize(torch.randn(2, 3, generator=gen), dim=-1)
        x = c[torch.randint(0, 2, (n,), generator=gen)] + 1e-3 * torch.randn(n, 3, generator=gen)
        y = c[torch.randint(0, 2, (n,), generator=gen)] + 1e-3 * torch.randn(n, 3, generator=gen)
    elif kind == "great_circle":       # all points on one great circle: some slices see them edge-on
        t = torch.rand(n, generator=gen) * 6.2831853
        x = torch.stack([torch.cos(t), torch.sin(t), torch.zeros(n)], 1)
        t2 = torch.rand(n, generator=gen) * 6.2831853
        y = torch.stack([torch.cos(t2), torch.zeros(n), torch.sin(t2)], 1)
    elif kind == "antipodal":          # source near a pole, target near the opposite pole: optimal shift ~ n/2
        x = torch.nn.functional.normalize(torch.tensor([0., 0., 1.]) + 0.2 * torch.randn(n, 3, generator=gen), dim=-1)
        y = torch.nn.functional.normalize(torch.tensor([0., 0., -1.]) + 0.2 * torch.randn(n, 3, generator=gen), dim=-1)
    elif kind == "duplicates":         # every point appears four times
        b = torch.nn.functional.normalize(torch.randn(n // 4 + 1, 3, generator=gen), dim=-1)
        x = b.repeat(4, 1)[:n]
        y = torch.nn.functional.normalize(torch.randn(n // 4 + 1, 3, generator=gen), dim=-1).repeat(4, 1)[:n]
    elif kind == "scales":             # un-normalised, wildly different magnitudes (the angle ignores them)
        x = torch.randn(n, 3, generator=gen) * torch.logspace(-6, 6, n).unsqueeze(1)
        y = torch.randn(n, 3, generator=gen) * 1e-4
    elif kind == "wrap":               # coordinates hugging 0 / 1: the cut sits on the seam of the circle
        t = (torch.rand(n, generator=gen) - 0.5) * 0.02
        x = torch.stack([-torch.cos(t), -torch.sin(t), torch.zeros(n)], 1)
        t2 = (torch.rand(n, generator=gen) - 0.5) * 0.02 + 0.01
        y = torch.stack([-torch.cos(t2), -torch.sin(t2), torch.zeros(n)], 1)
    else:
        raise ValueError(kind)
    return x.contiguous(), y.contiguous()


@pytest.mark.parametrize("kind", ["clustered", "great_circle", "antipodal", "duplicates", "scales", "wrap"])
@pytest.mark.parametrize("n", [200, 1024])
def test_adversarial_clouds_against_float64_oracle(shw, kind, n):
    from oracle import exact_shift
    gen = torch.Generator().manual_seed(sum(map(ord, kind)) + n)      # stable across processes
    x, y = _adversarial_clouds(kind, n, gen)
    U = directions(gen, 6)
    if kind == "wrap":                 # put the first slice in the plane of the arcs so the seam case is hit exactly
        U[0] = torch.tensor([[1., 0.], [0., 1.], [0., 0.]])
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, cost, shift = shw.ssw_pair_losses(xs[None], ys[None], U.cuda(), p=2, return_slices=True)
    pair.sum().backward()
    cu = exact_shift.circle_coords(x.numpy(), U.numpy())
    cv = exact_shift.circle_coords(y.numpy(), U.numpy())
    ref, _ = exact_shift.circular_ot_equal(cu, cv, 2)
    got = cost[0].detach().cpu().numpy()
    # absolute floor: a slice whose cost is ~1e-9 (tight clusters) is dominated by fp32 coordinate rounding (6e-8)^2
    assert np.all(np.abs(got - ref) <= 3e-5 * ref + 2e-9), (got, ref)
    assert torch.isfinite(xs.grad).all() and torch.isfinite(ys.grad).all()
    ref1 = np.array([exact_shift.w1_level_median(cu[l], cv[l]) for l in range(U.shape[0])])
    _, cost1, _ = shw.ssw_pair_losses(x.cuda()[None], y.cuda()[None], U.cuda(), p=1, return_slices=True)
    g1 = cost1[0].cpu().numpy()
    if kind not in ("clustered", "duplicates", "great_circle"):
        # (the p = 1 value is discontinuous in the ORDER of near-equal coordinates through the omitted first
        #  segment and the median pick; those three families are near-tie dominated by construction)
        assert np.all(np.abs(g1 - ref1) <= 1e-4 * ref1 + 1e-7), (g1, ref1)
    assert np.all(np.isfinite(g1))


@pytest.mark.parametrize("n", [64, 256, 1024, 2048, 4096])
@pytest.mark.parametrize("which", ["source", "target", "both"])
def test_last_point_in_top_quantisation_cell_of_a_full_row(shw, n, which):
    """Regression: in the training kernel's packed sort the atom with the LAST original index and a coordinate
    within 2^-QBITS of 1 packs to the all-ones word, which used to be taken for a padding slot (-> +inf cost,
    NaN gradients; seen once in ~1e5 slices of a training run).  Costs with gradients must equal the
    forward-only kernel's, which sorts plain float keys."""
    g = torch.Generator().manual_seed(n)
    x, y = unit_cloud(g, n), unit_cloud(g, n)
    U = directions(g, 8)
    edge = 0.5 * U[3, :, 0] - 1e-7 * U[3, :, 1]          # coordinate 1 - 3e-8 on slice 3
    if which in ("source", "both"):
        x[n - 1] = edge
    if which in ("target", "both"):
        y[n - 1] = edge * 1.5
    _, want, want_k = shw.ssw_pair_losses(x.cuda()[None], y.cuda()[None], U.cuda(), p=2, return_slices=True)
    xs, ys = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    pair, got, got_k = shw.ssw_pair_losses(xs[None], ys[None], U.cuda(), p=2, return_slices=True)
    pair.sum().backward()
    assert torch.isfinite(got).all() and torch.isfinite(xs.grad).all() and torch.isfinite(ys.grad).all()
    assert torch.equal(got_k, want_k)
    assert torch.allclose(got, want, rtol=1e-6, atol=0)
    if n <= 1024:
        from oracle import exact_shift
        gx, gy = exact_shift.ssw_pair_grad(x.numpy(), y.numpy(), U.numpy(), 2)
        grad_close(xs.grad.cpu().numpy(), gx)
        grad_close(ys.grad.cpu().numpy(), gy)
    # weighted (general) path sorts through the same packed helper
    w = torch.full((n,), 1.0 / n, device="cuda")
    if n <= 2048:
        _, got_w, _ = shw.ssw_pair_losses(x.cuda()[None], y.cuda()[None], U.cuda(), p=2, return_slices=True, u_weights=w, v_weights=w)
        assert torch.isfinite(got_w).all()
        assert torch.allclose(got_w, want, rtol=2e-4, atol=1e-7)


def test_short_soak_of_every_kernel_family(shw):
    """Ten seconds of tools/soak.py: random batches of several shapes and families through the forward, training,
    p = 1, general, Chamfer and Euclidean kernels with NaN-poisoned allocator memory; paths that must agree are
    compared with each other (~5e6 slices per path).  The four-minute form of the same run is how the packed-sort
    pad collision was characterised."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak.py")
    spec = importlib.util.spec_from_file_location("shw_soak", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(budget=10.0) == 0


@pytest.mark.parametrize("n,m", [(2048, 2048), (1024, 1024), (2000, 2000), (700, 1300), (2048, 1), (1, 1), (64, 64),
                                 (65, 63), (130, 2048), (1500, 1000), (5, 3)])
def test_p1_merge_kernel_agrees_with_the_search_kernel_and_the_oracle(shw, n, m):
    """p = 1, loss only: two waves per slice, merged by the sorting network (shw_ssw_p1_merge.hip); with gradients
    requested: the one-wave search kernel (exact coordinates).  Same closed form, so the per-slice costs must agree
    to fp32 rounding (the merge kernel clears one mantissa bit of every coordinate: <= 1 ulp), and both must match
    the float64 restatement of the reference's emd1D_circle."""
    from oracle import exact_shift
    g = torch.Generator().manual_seed(1000 * n + m)
    x, y = unit_cloud(g, n), unit_cloud(g, m)
    U = directions(g, 12)
    _, merged, _ = shw.ssw_pair_losses(x.cuda()[None], y.cuda()[None], U.cuda(), p=1, return_slices=True)
    xs = x.cuda().requires_grad_(True)
    _, searched, _ = shw.ssw_pair_losses(xs[None], y.cuda()[None], U.cuda(), p=1, return_slices=True)
    merged, searched = merged[0].cpu().numpy(), searched[0].detach().cpu().numpy()
    assert np.all(np.isfinite(merged))
    assert np.all(np.abs(merged - searched) <= 2e-5 * searched + 2e-7), (merged, searched)
    cu = exact_shift.circle_coords(x.numpy(), U.numpy())
    cv = exact_shift.circle_coords(y.numpy(), U.numpy())
    ref = np.array([exact_shift.w1_level_median(cu[l], cv[l]) for l in range(U.shape[0])])
    assert np.all(np.abs(merged - ref) <= 1e-4 * ref + 2e-7), (merged, ref)


@pytest.mark.parametrize("n,m", [(1000, 1000), (2048, 2048), (700, 1300), (64, 64), (3, 5)])
def test_p1_training_kernel_gradients_against_torch_autograd_of_the_restatement(shw, n, m):
    """p = 1 with gradients (two-wave merge kernel with indices): loss and d loss / d points against torch autograd
    through the CPU restatement of emd1D_circle (oracle/ref_mirror.py, pinned by the G1/G4 p=1 gradient fixtures).
    The p = 1 gradient is piecewise constant in the ORDER of the merged atoms and jumps with the median level:
    entries are compared with the near-tie allowance of grad_close."""
    from oracle import ref_mirror
    g = torch.Generator().manual_seed(77 * n + m)
    x, y = unit_cloud(g, n), unit_cloud(g, m)
    U = directions(g, 10)
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


@pytest.mark.parametrize("poison", ["nan", "inf", "zero_cloud"])
def test_non_finite_and_degenerate_inputs_terminate_on_every_path(shw, poison):
    """Every kernel loop has a constant bound: garbage in must give garbage out, never a hang or a fault.  NaN / inf
    coordinates and an all-zero cloud through the equal-size, p = 1, unequal-size, weighted, Euclidean and Chamfer
    paths, forward and backward; afterwards the same paths must still give finite values on clean inputs."""
    g = torch.Generator().manual_seed(99)
    n, m = 300, 257
    x, y, y2 = unit_cloud(g, 2, n), unit_cloud(g, 2, n), unit_cloud(g, 2, m)
    if poison == "nan":
        x[0, 5] = float("nan"); y[1, 7, 1] = float("nan"); y2[0, 0] = float("nan")
    elif poison == "inf":
        x[0, 5] = float("inf"); y[1, 7, 1] = -float("inf"); y2[0, 0, 2] = float("inf")
    else:
        x[0] = 0.0; y2[1] = 0.0
    U = directions(g, 2, 16).cuda()
    w1 = torch.full((n,), 1.0 / n, device="cuda")
    w2 = torch.rand(m, generator=g).cuda() + 0.1
    w2 = w2 / w2.sum()

    def run_all(a, b, b2):
        for p in (2, 1, 3):
            for second, kw in ((b, {}), (b2, {}), (b2, {"u_weights": w1, "v_weights": w2})):
                aa, bb = a.clone().requires_grad_(True), second.clone().requires_grad_(True)
                out = shw.ssw_pair_losses(aa, bb, U, p, **kw)
                out.sum().backward()
        aa, bb = a.clone().requires_grad_(True), b2.clone().requires_grad_(True)
        shw.chamfer_distance(aa, bb)[0].backward()
        shw.sliced_wasserstein_distance(a[0], b[0], num_projection=8, p=2, device="cuda")
        torch.cuda.synchronize()
        return out

    run_all(x.cuda(), y.cuda(), y2.cuda())
    clean = run_all(unit_cloud(g, 2, n).cuda(), unit_cloud(g, 2, n).cuda(), unit_cloud(g, 2, m).cuda())
    assert torch.isfinite(clean).all()
