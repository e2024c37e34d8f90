"""Developer aid: a few evaluations of one general-path problem, for rocprofv3 --kernel-trace --stats.
usage: python tools/general_prof.py [weighted|unequal] [train|loss] [p]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "weighted"
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
p = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
B, n, L = 64, 2048, 512
m = 2048 if kind == "weighted" else 1536
x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).to(dev)
y = torch.nn.functional.normalize(torch.randn(B, m, 3, generator=g), dim=-1).to(dev)
U = shw.draw_directions(L, dev, batch=B, d=3)
wu = wv = None
if kind == "weighted":
    wu = torch.rand(B, n, generator=g).to(dev) + 0.1
    wu = wu / wu.sum(1, keepdim=True)
    wv = torch.rand(B, m, generator=g).to(dev) + 0.1
    wv = wv / wv.sum(1, keepdim=True)
xs, ys = x.clone().requires_grad_(mode == "train"), y.clone().requires_grad_(mode == "train")
for _ in range(12):
    out = shw.ssw_pair_losses(xs, ys, U, p, u_weights=wu, v_weights=wv)
    if mode == "train":
        xs.grad = None
        ys.grad = None
        out.sum().backward()
torch.cuda.synchronize()
print("done", float(out.sum()))
