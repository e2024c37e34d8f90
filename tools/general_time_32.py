"""Developer aid: the general path at the 2048-point class only (variant libraries built with SHW_DEV_ONLY_EPT=32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402
from tools.general_time import timed  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
tag = os.path.basename(os.environ.get("SHW_LIB_PATH", "default"))
for (B, n, m, L, p, weighted) in [(64, 2048, 2048, 512, 2, True), (64, 2048, 1536, 512, 2, False), (64, 1800, 1100, 512, 2, False),
                                  (64, 2048, 2048, 512, 1, True)]:
    x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).to(dev)
    y = torch.nn.functional.normalize(torch.randn(B, m, 3, generator=g), dim=-1).to(dev)
    U = shw.draw_directions(L, dev, batch=B, d=3)
    wu = wv = None
    if weighted:
        wu = torch.rand(B, n, generator=g).to(dev) + 0.1
        wu = wu / wu.sum(1, keepdim=True)
        wv = torch.rand(B, m, generator=g).to(dev) + 0.1
        wv = wv / wv.sum(1, keepdim=True)
    fwd = timed(lambda: shw.ssw_pair_losses(x, y, U, p, u_weights=wu, v_weights=wv))
    xs, ys = x.clone().requires_grad_(True), y.clone().requires_grad_(True)

    def step():
        xs.grad = None
        ys.grad = None
        shw.ssw_pair_losses(xs, ys, U, p, u_weights=wu, v_weights=wv).sum().backward()
    print(f"{tag} n={n} m={m} p={p} weighted={weighted}: loss {fwd:.3f} ms, training step {timed(step):.3f} ms", flush=True)
