"""The notebooks' gradient-flow step (Flow_cube.ipynb:1372-1395): evolving cloud of 1200 points, 100 projections,
loss = sliced_wasserstein_sphere(evolving, target, 100, device, p=2); loss.backward(); Adam step.
The saved notebook outputs record 0.531 s per 5 such steps (unstated hardware, includes one
ot.emd2 probe).  This script times the same step on the HIP path:
    eager        the notebook's loop as written (torch.optim.Adam defaults)
    graph        the same step captured once and replayed (shw.GraphedStep; Adam(capturable=True))
    graph-fused  ... with Adam(capturable=True, fused=True)
usage: python tools/notebook_flow_time.py [eager|graph|graph-fused|all] [p]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
p = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
dev = "cuda"
N, L, K = 1200, 100, 400


def run(kind):
    g = torch.Generator().manual_seed(0)
    target = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
    evolving = (torch.randn(N, 3, generator=g) * 0.5).to(dev).requires_grad_(True)

    def loss_fn():
        return shw.sliced_wasserstein_sphere(evolving, target, L, device=dev, p=p)
    if kind == "eager":
        opt = torch.optim.Adam([evolving], lr=1e-2)

        def step():
            opt.zero_grad()
            loss = loss_fn()
            loss.backward()
            opt.step()
            return loss
    else:
        opt = torch.optim.Adam([evolving], lr=1e-2, capturable=True, fused=(kind == "graph-fused"))
        step = shw.GraphedStep(loss_fn, opt)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    print(f"gradient-flow step N={N} L={L} p={p:g} [{kind}]: {1e3 * dt:.4f} ms/step ({5 * dt:.5f} s per 5 steps; notebook "
          f"record 0.531 s), loss {float(loss):.6f}", flush=True)


for kind in (("eager", "graph", "graph-fused") if mode == "all" else (mode,)):
    run(kind)
