"""Developer aid: trace of the weighted cut solve of the first slices (library built with -DSHW_DBG_TRACE)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
B, n, m, L, p = 1, 2048, 2048, 4, 2
x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).to(dev)
y = torch.nn.functional.normalize(torch.randn(B, m, 3, generator=g), dim=-1).to(dev)
U = shw.draw_directions(L, dev, batch=B, d=3)
wu = torch.rand(B, n, generator=g).to(dev) + 0.1
wu = wu / wu.sum(1, keepdim=True)
wv = torch.rand(B, m, generator=g).to(dev) + 0.1
wv = wv / wv.sum(1, keepdim=True)
print(shw.ssw_pair_losses(x, y, U, p, u_weights=wu, v_weights=wv))
torch.cuda.synchronize()
