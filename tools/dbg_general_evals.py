"""Developer aid: evaluations of the cut solve per slice (library built with -DSHW_DBG_EVALS, SHW_LIB_PATH set)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402
from shw_amd import ssw as _ssw  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
for (B, n, m, L, p) in [(8, 2048, 2048, 512, 2), (8, 2048, 2048, 512, 1), (8, 2048, 1536, 512, 2), (8, 500, 700, 128, 3)]:
    x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).to(dev)
    y = torch.nn.functional.normalize(torch.randn(B, m, 3, generator=g), dim=-1).to(dev)
    U = shw.draw_directions(L, dev, batch=B, d=3)
    wu = torch.rand(B, n, generator=g).to(dev) + 0.1
    wu = wu / wu.sum(1, keepdim=True)
    wv = torch.rand(B, m, generator=g).to(dev) + 0.1
    wv = wv / wv.sum(1, keepdim=True)
    shw.ssw_pair_losses(x, y, U, p, u_weights=wu, v_weights=wv)
    torch.cuda.synchronize()
    ws = [w for w in _ssw.SSWWorkspace._pools.values()]
    print(B, n, m, L, p, "pools", len(ws))
    for pool in ws:
        for w in (pool if isinstance(pool, (list, tuple)) else [pool]):
            if w.slice_aux.numel() == B * L:
                v = w.slice_aux.view(torch.float32)[: B * L].cpu()
                ev = (v % 100)
                br = (v // 100)
                print("  evals: mean %.2f max %d ; before bracket mean %.2f max %d" % (ev.mean(), ev.max(), br.mean(), br.max()))
                print("  hist", torch.bincount(ev.long()).tolist())
