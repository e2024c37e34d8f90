"""Euclidean sliced-Wasserstein, the notebooks' SWD baseline
(/root/reference/Wasserstein_flow_problem/Flow_cube.ipynb:275-292: `rand_projections`,
`sliced_wasserstein_distance`), on the HIP path.  Same call shape:

    sliced_wasserstein_distance(first_samples (n,3), second_samples (n,3), num_projection=100, p=2, device='cuda')

Notes on the reference cell: it draws the directions on the CPU generator and moves them to `device`
(`torch.randn((L, dim))` then `.to(device)`), which this mirror does too; it reads a *global* `num_projections`
instead of its own `num_projection` argument (a notebook slip) -- here the argument is used.  The notebook cannot
be imported (its `datas` / `losses` modules are not shipped); the cell itself is self-contained torch code and fixture
G9 (tests/golden/g9_notebook_esw.npz, oracle/make_golden.py) holds its outputs -- values and gradients -- exec'd from
the .ipynb JSON: the family is pinned by the reference (round 2).
`max_sliced_wasserstein_distance` (:294-323: Adam ascent on ONE direction, then the distance along it) is mirrored
too; the op is differentiable w.r.t. the clouds and the directions."""
from __future__ import annotations

import torch

from . import _lib
from .ssw import _check_cloud, _stream_ptr


def rand_projections(dim, num_projections=100):
    projections = torch.randn((num_projections, dim))
    return projections / torch.sqrt(torch.sum(projections ** 2, dim=1, keepdim=True))


class _SliceSums(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Xs, Xt, thetas, p, need_grad=True):
        lib = _lib.load()
        B, n, _ = Xs.shape
        L = thetas.shape[-2]
        dev = Xs.device
        xs, xt, th = Xs.contiguous(), Xt.contiguous(), thetas.contiguous()
        stride = 0 if th.dim() == 2 else L * 3
        sums = torch.empty(B * L, dtype=torch.float32, device=dev)
        need = need_grad
        cs = ct = None
        if need:
            cs = torch.empty(B * L * n, dtype=torch.float32, device=dev)
            ct = torch.empty(B * L * n, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.shw_esw_forward(xs.data_ptr(), xt.data_ptr(), th.data_ptr(), B, n, L, stride, float(p),
                                           sums.data_ptr(), cs.data_ptr() if need else None,
                                           ct.data_ptr() if need else None, _stream_ptr(dev)), "shw_esw_forward")
        if need:
            ctx.save_for_backward(th, cs, ct, xs, xt)
            ctx.dims = (B, n, L, stride)
            ctx.theta_shape = tuple(thetas.shape)
        return sums.view(B, L)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        th, cs, ct, xs, xt = ctx.saved_tensors
        B, n, L, stride = ctx.dims
        dev = th.device
        gxs = gxt = gth = None
        w = g.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                gxs = torch.empty(B, n, 3, dtype=torch.float32, device=dev)
                gxt = torch.empty(B, n, 3, dtype=torch.float32, device=dev)
                _lib.check(lib.shw_esw_backward_points(th.data_ptr(), cs.data_ptr(), ct.data_ptr(), w.data_ptr(), B, n,
                                                       L, stride, gxs.data_ptr(), gxt.data_ptr(), _stream_ptr(dev)),
                           "shw_esw_backward_points")
            if ctx.needs_input_grad[2]:
                gth = torch.empty(B, L, 3, dtype=torch.float32, device=dev)
                _lib.check(lib.shw_esw_backward_dirs(xs.data_ptr(), xt.data_ptr(), cs.data_ptr(), ct.data_ptr(),
                                                     w.data_ptr(), B, n, L, gth.data_ptr(), _stream_ptr(dev)),
                           "shw_esw_backward_dirs")
                if len(ctx.theta_shape) == 2:          # directions shared by the pairs
                    gth = gth.sum(0)
        return gxs, gxt, gth, None, None


def esw_slice_sums(Xs, Xt, thetas, p=2):
    """(B,n,3), (B,n,3), directions (L,3) or (B,L,3) -> (B,L) per-slice sums of |sorted difference|^p."""
    _check_cloud("Xs", Xs)
    _check_cloud("Xt", Xt)
    if Xs.shape != Xt.shape or Xs.dim() != 3:
        raise ValueError("the Euclidean sliced distance needs two (B,n,3) clouds of equal size")
    if not thetas.is_cuda or thetas.dtype != torch.float32 or thetas.shape[-1] != 3:
        raise TypeError("thetas must be a float32 device tensor (L,3) or (B,L,3)")
    need = torch.is_grad_enabled() and (Xs.requires_grad or Xt.requires_grad or thetas.requires_grad)
    return _SliceSums.apply(Xs, Xt, thetas, float(p), need)


def sliced_wasserstein_distance(first_samples, second_samples, num_projection=100, p=2, device="cuda"):
    dim = second_samples.size(1)
    projections = rand_projections(dim, num_projection).to(device)
    sums = esw_slice_sums(first_samples.unsqueeze(0), second_samples.unsqueeze(0), projections, p)
    return torch.pow(sums.mean(), 1.0 / p)


def max_sliced_wasserstein_distance(first_samples, second_samples, num_projection=100, p=2, max_iter=10, device="cuda"):
    """Flow_cube.ipynb:294-323: one direction, `max_iter` Adam ascent steps on it (lr 0.005, betas (0.999, 0.999),
    re-normalised after every step) against the detached clouds, then the distance along the final direction."""
    dim = second_samples.size(1)
    first_d, second_d = first_samples.detach().unsqueeze(0), second_samples.detach().unsqueeze(0)
    projections = rand_projections(dim, 1).to(device)
    projections.requires_grad_()
    optimizer = torch.optim.Adam([projections], lr=0.005, betas=(0.999, 0.999))
    for _ in range(max_iter):
        dist_l = torch.pow(esw_slice_sums(first_d, second_d, projections, p).mean(), 1.0 / p)
        optimizer.zero_grad()
        (-dist_l).backward()
        optimizer.step()
        projections.data = projections.data / torch.sqrt(torch.sum(projections.data ** 2, dim=1))
    sums = esw_slice_sums(first_samples.unsqueeze(0), second_samples.unsqueeze(0), projections.detach(), p)
    return torch.pow(sums.mean(), 1.0 / p)
