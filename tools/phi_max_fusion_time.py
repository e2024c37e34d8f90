"""Before / after table of the phi-max fusion: one trainer-level call of
max_spherical_wassersten_distance_fast at B=32, N=2048, L=512, max_iter=10 -- eager inner loop vs hipGraph replay
(GraphedAscent), with a small planar-flow phi.   python tools/phi_max_fusion_time.py  (on the GPU box)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402


class PlanarFlow(torch.nn.Module):
    def __init__(self, n_layers=3, dim=3):
        super().__init__()
        self.u = torch.nn.Parameter(0.1 * torch.randn(n_layers, dim))
        self.w = torch.nn.Parameter(0.5 * torch.randn(n_layers, dim))
        self.b = torch.nn.Parameter(torch.zeros(n_layers))

    def forward(self, x):
        for u, w, b in zip(self.u, self.w, self.b):
            x = x + u * torch.tanh((x * w).sum(-1, keepdim=True) + b)   # (x @ w would go to a rocBLAS gemv: 0.5 ms per call)
        return x


def main(B=32, N=2048, L=512, iters=10, reps=20):
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    x = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1).to(dev)
    y = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1).to(dev)
    rows = []
    for graph in (False, True):
        torch.manual_seed(0)
        phi = PlanarFlow().to(dev)
        opt = torch.optim.Adam(phi.parameters(), lr=1e-4, capturable=True)
        crit = shw.max_spherical_wassersten_distance_fast(L, phi, shw.sliced_wasserstein_sphere_fast, opt, p=2,
                                                          max_iter=iters, device=dev, graph=graph)
        for _ in range(3):
            crit(x, y, train_or_test="train")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            val, _, _ = crit(x, y, train_or_test="train")
        torch.cuda.synchronize()
        rows.append(("hipGraph replay" if graph else "eager", 1e3 * (time.perf_counter() - t0) / reps, float(val.sum())))
    print("phi-max wrapper, B=%d N=%d L=%d max_iter=%d (one trainer-level call = %d loss evaluations with gradient)"
          % (B, N, L, iters, iters + 1))
    for name, ms, val in rows:
        print("  %-16s %8.3f ms per call   %7.3f ms per inner iteration   (ssw %.6f)" % (name, ms, ms / (iters + 1), val))
    print("  speed-up %.2fx" % (rows[0][1] / rows[1][1]))


if __name__ == "__main__":
    main()
