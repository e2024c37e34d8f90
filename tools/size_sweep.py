"""Size sweep of the loss (forward) and training (forward + input gradients) paths through the drop-in Python call:
point-pairs/s per size, relative to the next power of two.   python tools/size_sweep.py   (GPU box)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402


def rate(N, L, B, train, p=2, reps=40):
    g = torch.Generator().manual_seed(N)
    x = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1).cuda()
    y = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1).cuda()
    U = shw.stiefel_frames(torch.randn(B, L, 3, 2, generator=g).cuda())
    if train:
        x.requires_grad_(True)

    def step():
        if train:
            x.grad = None
            shw.sliced_cost(x, y, U, p=p).backward()
        else:
            shw.sliced_cost(x, y, U, p=p)
    best = float('inf')
    for _trial in range(3):                      # best of three timed blocks after a warm-up block
        for _ in range(15):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    dt = best
    del x, y, U
    shw.ssw.SSWWorkspace.clear()
    torch.cuda.empty_cache()
    return B * N * L / dt, 1e3 * dt


def main():
    sizes = [256, 500, 512, 1000, 1024, 1200, 1500, 2000, 2048, 3000, 4096, 5000, 8192]
    print("%6s %5s | %-24s | %-24s" % ("N", "B", "loss only", "loss + input gradients"))
    ps = tuple(int(a) for a in sys.argv[1].split(',')) if len(sys.argv) > 1 else (2, 1)
    for p in ps:
        print("p = %d" % p)
        for N in sizes:
            B = max(8, min(64, 131072 // N))
            L = 512
            f, fm = rate(N, L, B, False, p)
            t, tm = rate(N, L, B, True, p)
            print("%6d %5d | %9.3e pp/s %7.3f ms | %9.3e pp/s %7.3f ms" % (N, B, f, fm, t, tm))


if __name__ == "__main__":
    main()
