"""Developer aid: loss-only time of the weighted general path at VERDICT's shape (B=64, n=m=2048, L=512)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402
from tools.general_time import timed  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
B, n, m, L = 64, 2048, 2048, 512
x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).to(dev)
y = torch.nn.functional.normalize(torch.randn(B, m, 3, generator=g), dim=-1).to(dev)
U = shw.draw_directions(L, dev, batch=B, d=3)
wu = torch.rand(B, n, generator=g).to(dev) + 0.1
wu = wu / wu.sum(1, keepdim=True)
wv = torch.rand(B, m, generator=g).to(dev) + 0.1
wv = wv / wv.sum(1, keepdim=True)
for p in (2, 1):
    t = timed(lambda: shw.ssw_pair_losses(x, y, U, p, u_weights=wu, v_weights=wv))
    xs, ys = x.clone().requires_grad_(True), y.clone().requires_grad_(True)

    def step():
        xs.grad = None
        ys.grad = None
        shw.ssw_pair_losses(xs, ys, U, p, u_weights=wu, v_weights=wv).sum().backward()
    print(f"{os.environ.get('SHW_LIB_PATH', 'default')[-12:]} p={p}: loss {t:.3f} ms, training step {timed(step):.3f} ms", flush=True)
