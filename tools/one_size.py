"""Developer aid: loss-only launches of one size (for rocprofv3 --kernel-trace --stats): python tools/one_size.py N [B] [p]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, min(64, 131072 // n))
p = float(sys.argv[3]) if len(sys.argv) > 3 else 2
g = torch.Generator().manual_seed(1)
x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).to(dev)
y = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).to(dev)
U = shw.draw_directions(256, dev, batch=B, d=3)
for _ in range(30):
    shw.ssw_pair_losses(x, y, U, p)
torch.cuda.synchronize()
