"""Log-domain Sinkhorn distance with the reference's class interface
(/root/reference/Comparison_Wasserstein_with_Chamfer_distance/losses/sinkhorn.py:3-87 and :92-186; used at
main_rotation.py:207-211):

    criteria = log_Sinkhorn_Distance_Loss(eps, max_iter, batch_reduction='sum', type_of_cost_norm='L2')
    loss, P, C = criteria(template, source, device)

Differentiable like the reference (its forward is autograd-visible through the unrolled iterations, :35-49): when a
cloud requires grad the iterations keep the trajectory of the duals and `backward` walks it (shw_sinkhorn_backward;
round 2).  The dense plan P and cost matrix C the
reference returns are produced on request by the SAME solve (`return_plan=True`, the default, keeps the call a drop-in;
`return_plan=False` returns (cost, None, None) and never allocates the two (B, n, m) tensors).  In training P and C are
plain values marked non-differentiable (the reference's are autograd-visible): the gradient is that of the cost."""
from __future__ import annotations

import torch

from . import _lib
from .ssw import _check_cloud, _stream_ptr


class _SinkhornCosts(torch.autograd.Function):
    """(B,n,3), (B,m,3) -> (B,) costs with the gradient through the unrolled iterations."""

    @staticmethod
    def forward(ctx, x, y, eps, max_iter, norm_p, cost_pow, thresh, return_plan):
        lib = _lib.load()
        B, n, _ = x.shape
        m = y.shape[1]
        dev = x.device
        xc, yc = x.contiguous(), y.contiguous()
        ws = torch.empty(lib.shw_sinkhorn_train_workspace_bytes(B, n, m, max_iter), dtype=torch.uint8, device=dev)
        cost = torch.empty(B, dtype=torch.float32, device=dev)
        # the dense outputs of the reference, from the SAME solve (a second solve doubled the step)
        P = torch.empty(B, n, m, dtype=torch.float32, device=dev) if return_plan else None
        C = torch.empty(B, n, m, dtype=torch.float32, device=dev) if return_plan else None
        with torch.cuda.device(dev):
            _lib.check(lib.shw_sinkhorn_forward_train(xc.data_ptr(), yc.data_ptr(), B, n, m, eps, max_iter, norm_p,
                                                      cost_pow, thresh, ws.data_ptr(), cost.data_ptr(),
                                                      P.data_ptr() if return_plan else None,
                                                      C.data_ptr() if return_plan else None, _stream_ptr(dev)),
                       "shw_sinkhorn_forward_train")
        ctx.save_for_backward(xc, yc, ws)
        ctx.cfg = (B, n, m, eps, max_iter, norm_p, cost_pow)
        if return_plan:
            ctx.mark_non_differentiable(P, C)
        return cost, P, C

    @staticmethod
    def backward(ctx, g, _gP=None, _gC=None):
        lib = _lib.load()
        xc, yc, ws = ctx.saved_tensors
        B, n, m, eps, max_iter, norm_p, cost_pow = ctx.cfg
        dev = xc.device
        gx, gy = torch.empty_like(xc), torch.empty_like(yc)
        gc = g.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            _lib.check(lib.shw_sinkhorn_backward(xc.data_ptr(), yc.data_ptr(), B, n, m, eps, max_iter, norm_p, cost_pow,
                                                 ws.data_ptr(), gc.data_ptr(), gx.data_ptr(), gy.data_ptr(),
                                                 _stream_ptr(dev)), "shw_sinkhorn_backward")
        return gx, gy, None, None, None, None, None, None


def sinkhorn_pair_costs(x, y, eps, max_iter, norm_p=2, cost_pow=1, thresh=1e-9, return_plan=False):
    """(B,n,3), (B,m,3) -> (B,) transport costs sum_ij P_ij C_ij [, P, C]."""
    _check_cloud("x", x)
    _check_cloud("y", y)
    if x.dim() != 3 or y.dim() != 3 or x.shape[0] != y.shape[0]:
        raise ValueError("x and y must be (B,n,3) and (B,m,3) with the same B")
    if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad):
        # P and C come out of the same solve as plain tensors: they are marked non-differentiable (the reference's P and C
        # are autograd-visible; a loss built on them gets no gradient here -- use the cost)
        return _SinkhornCosts.apply(x, y, float(eps), int(max_iter), int(norm_p), int(cost_pow), float(thresh),
                                    bool(return_plan))
    lib = _lib.load()
    B, n, _ = x.shape
    m = y.shape[1]
    dev = x.device
    xc, yc = x.contiguous(), y.contiguous()
    ws = torch.empty(lib.shw_sinkhorn_workspace_bytes(B, n, m), dtype=torch.uint8, device=dev)
    cost = torch.empty(B, dtype=torch.float32, device=dev)
    P = C = None
    if return_plan:
        P = torch.empty(B, n, m, dtype=torch.float32, device=dev)
        C = torch.empty(B, n, m, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.shw_sinkhorn_forward(xc.data_ptr(), yc.data_ptr(), B, n, m, float(eps), int(max_iter),
                                            int(norm_p), int(cost_pow), float(thresh), ws.data_ptr(), cost.data_ptr(),
                                            P.data_ptr() if return_plan else None,
                                            C.data_ptr() if return_plan else None, _stream_ptr(dev)),
                   "shw_sinkhorn_forward")
    return cost, P, C


class log_Sinkhorn_Distance_Loss(torch.nn.Module):
    """Reference signature (sinkhorn.py:7, :14)."""

    cost_pow = 1

    def __init__(self, eps, max_iter, batch_reduction="none", type_of_cost_norm="L2", return_plan=True):
        super().__init__()
        self.eps = eps
        self.max_iter = max_iter
        self.batch_reduction = batch_reduction
        self.p = int(type_of_cost_norm[-1])
        self.return_plan = return_plan

    def forward(self, x, y, device=None):
        single = x.dim() == 2
        if single:
            x, y = x.unsqueeze(0), y.unsqueeze(0)
        cost, P, C = sinkhorn_pair_costs(x, y, self.eps, self.max_iter, self.p, self.cost_pow, 1e-9, self.return_plan)
        if self.cost_pow != 1:
            cost = torch.pow(cost, 1.0 / self.cost_pow)
        if single:
            cost = cost[0]
            P = None if P is None else P[0]
            C = None if C is None else C[0]
        if self.batch_reduction == "mean":
            cost = cost.mean()
        elif self.batch_reduction == "sum":
            cost = cost.sum()
        return cost, P, C


class log_N_Sinkhorn_Distance_Loss(log_Sinkhorn_Distance_Loss):
    """Reference signature (sinkhorn.py:96): cost matrix raised to the power N, result to the power 1/N."""

    def __init__(self, eps, max_iter, batch_reduction="none", type_of_cost_norm="L2", type_of_Wasserstein_N="2",
                 return_plan=True):
        super().__init__(eps, max_iter, batch_reduction, type_of_cost_norm, return_plan)
        self.N = int(type_of_Wasserstein_N)
        self.cost_pow = self.N
