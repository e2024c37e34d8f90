"""the distribution sort assumes ~2 keys per bin and sends a slice to the bitonic network when any bin holds
more than 24 (round 3: 40) keys -- kernel time is data dependent.  This times the headline loss kernel (B=64, N=2048, L=512, p=2) on
cloud families that are NOT uniform in angle, and -- with a library built with -DSHW_DBG_RUNLEN (slice_shift then reports
the longest equal-bin run of the slice's two sorts) -- the distribution of that run length and the share of slices that
took the fallback.
    python tools/nonuniform_time.py            (GPU box; SHW_LIB_PATH may point at the SHW_DBG_RUNLEN variant)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402
from tools.general_time import timed  # noqa: E402

dev = torch.device("cuda", 0)
B, N, L = 64, 2048, 512
g = torch.Generator().manual_seed(11)


def family(name):
    x = torch.randn(B, N, 3, generator=g)
    if name == "gaussian sphere (bench.py)":
        return torch.nn.functional.normalize(x, dim=-1)
    if name == "cube surface (the notebooks)":
        face = torch.randint(0, 3, (B, N), generator=g)
        pts = torch.rand(B, N, 3, generator=g)
        pts.scatter_(2, face.unsqueeze(-1), torch.randint(0, 2, (B, N, 1), generator=g).float())
        return pts
    if name == "CAD-like: 6 planes + 2 cylinders, centred, unit scale":
        kind = torch.randint(0, 8, (B, N), generator=g)
        uv = torch.rand(B, N, 2, generator=g) * 2 - 1
        pts = torch.zeros(B, N, 3)
        for k in range(6):
            sel = kind == k
            ax = k // 2
            p = torch.zeros(B, N, 3)
            p[..., ax] = 1.0 if k % 2 else -1.0
            p[..., (ax + 1) % 3] = uv[..., 0]
            p[..., (ax + 2) % 3] = uv[..., 1]
            pts[sel] = p[sel] * torch.tensor([1.0, 0.6, 0.3])
        for k in (6, 7):
            sel = kind == k
            ang = uv[..., 0] * math.pi
            p = torch.stack([0.4 * torch.cos(ang), 0.4 * torch.sin(ang), uv[..., 1]], -1)
            pts[sel] = (p + (0.5 if k == 6 else -0.5) * torch.tensor([1.0, 0.0, 0.0]))[sel]
        pts = pts - pts.mean(1, keepdim=True)
        return pts / pts.norm(dim=-1).amax(1, keepdim=True).unsqueeze(-1)
    if name == "16 tight clusters on the sphere (sigma 0.02)":
        centres = torch.nn.functional.normalize(torch.randn(B, 16, 3, generator=g), dim=-1)
        pick = torch.randint(0, 16, (B, N), generator=g)
        return torch.nn.functional.normalize(torch.gather(centres, 1, pick.unsqueeze(-1).expand(B, N, 3)) + 0.02 * x, dim=-1)
    if name == "great circle band (|z| < 0.02)":
        x[..., 2] *= 0.02
        return torch.nn.functional.normalize(x, dim=-1)
    if name == "64 distinct points, each 32 times":
        base = torch.nn.functional.normalize(torch.randn(B, 64, 3, generator=g), dim=-1)
        return base.repeat(1, N // 64, 1)
    raise ValueError(name)


names = ["gaussian sphere (bench.py)", "cube surface (the notebooks)", "CAD-like: 6 planes + 2 cylinders, centred, unit scale",
         "16 tight clusters on the sphere (sigma 0.02)", "great circle band (|z| < 0.02)", "64 distinct points, each 32 times"]
MAX_RUN = int(os.environ.get("SHW_BINSORT_MAX_RUN", "40"))      # csrc/bin_sort.hpp (a compile-time constant of the library)
U = shw.draw_directions(L, dev, batch=B, d=3)
print("library:", os.path.basename(os.environ.get("SHW_LIB_PATH", "default")))
for name in names:
    x, y = family(name).to(dev), family(name).to(dev)
    t = timed(lambda: shw.ssw_pair_losses(x, y, U, 2), warm=10, reps=50)
    _, cost, aux = shw.ssw_pair_losses(x, y, U, 2, return_slices=True)
    line = f"{name:58s} loss {t:.3f} ms"
    if os.environ.get("SHW_RUNLEN") == "1":
        run = aux.flatten().float()
        q = torch.quantile(run, torch.tensor([0.5, 0.9, 0.99], device=dev))
        line += (f" | longest run: median {q[0]:.0f}, p90 {q[1]:.0f}, p99 {q[2]:.0f}, max {run.max():.0f}; "
                 f"network fallback (run > {MAX_RUN}, SHW_BINSORT_MAX_RUN of this build) {100 * (run > MAX_RUN).float().mean():.2f} % of slices"
                 f" (run > 24, round 2's threshold: {100 * (run > 24).float().mean():.2f} %)")
    print(line, flush=True)
