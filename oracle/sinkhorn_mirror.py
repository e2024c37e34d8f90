"""oracle/sinkhorn_mirror.py -- CPU restatement (torch, float64-capable) of the reference's log-domain Sinkhorn
(/root/reference/Comparison_Wasserstein_with_Chamfer_distance/losses/sinkhorn.py:14-63, :104-157).
TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/g7_sinkhorn.npz (outputs of the real class, make_golden.py)."""
import torch


def sinkhorn_costs(x, y, eps, max_iter, norm_p=2, cost_pow=1, thresh=1e-9):
    """(B,n,3), (B,m,3) -> (cost (B,), P (B,n,m), C (B,n,m), iterations run)"""
    C = torch.sum(torch.abs(x.unsqueeze(-2) - y.unsqueeze(-3)) ** norm_p, -1) ** cost_pow
    B, n, m = C.shape
    log_a = torch.log(torch.full((B, n), 1.0 / n, dtype=torch.float32) + 1e-8).to(C.dtype)
    log_b = torch.log(torch.full((B, m), 1.0 / m, dtype=torch.float32) + 1e-8).to(C.dtype)
    u = torch.zeros(B, n, dtype=C.dtype)
    v = torch.zeros(B, m, dtype=C.dtype)

    def M(u, v):
        return (-C + u.unsqueeze(-1) + v.unsqueeze(-2)) / eps

    it = 0
    for it in range(1, max_iter + 1):
        u_prev = u
        u = eps * (log_a - torch.logsumexp(M(u, v), dim=-1)) + u
        v = eps * (log_b - torch.logsumexp(M(u, v).transpose(-2, -1), dim=-1)) + v
        if (u - u_prev).abs().sum(-1).mean().item() < thresh:
            break
    P = torch.exp(M(u, v))
    return torch.sum(P * C, dim=(-2, -1)), P, C, it
