"""Timing of the general circular-OT path (n != m and / or weights) on the GPU box: ms per loss evaluation and per
training step (loss + input gradients), HIP-event timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, warm=5, reps=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    g = torch.Generator().manual_seed(5)
    for (B, n, m, L, p, weighted) in [(64, 2048, 2048, 512, 2, True), (64, 2048, 1536, 512, 2, False), (64, 1024, 768, 256, 2, False),
                                      (64, 2048, 1536, 512, 1, False), (64, 2048, 2048, 512, 1, True), (64, 717, 1024, 256, 2, False)]:
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
        trn = timed(step)
        print(f"B={B} n={n} m={m} L={L} p={p} weighted={weighted}: loss {fwd:.3f} ms, training step {trn:.3f} ms "
              f"({B * max(n, m) * L / fwd * 1e3:.3g} point-pairs/s)", flush=True)


if __name__ == "__main__":
    main()
