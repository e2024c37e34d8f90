#!/usr/bin/env python3
"""BASELINE config 5 in miniature: the trainer-level drop-in.  A PointNet-style pose regressor written here (the
reference's PCRNet, models/pcrnet.py:7-62, is out of scope -- stock nn layers) is trained on ModelNet-SHAPED
synthetic clouds (no dataset offline) with the sliced loss in the `criteria(template, transformed_source,
train_or_test=...)` slot of train_W_COS.py:171, forward + backward + Adam on one GPU.

    python examples/config5_train_step.py [--batch 32] [--points 2048] [--slices 512] [--steps 20]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402


def synthetic_batch(B, N, gen, device):
    """Unit-scale random surfaces (points on randomly stretched ellipsoids) + a random rigid motion <= 45 deg."""
    pts = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=gen), dim=-1)
    pts = pts * (0.4 + 0.6 * torch.rand(B, 1, 3, generator=gen))
    ang = (torch.rand(B, generator=gen) - 0.5) * (3.14159 / 2)
    c, s = torch.cos(ang), torch.sin(ang)
    R = torch.zeros(B, 3, 3)
    R[:, 0, 0] = 1
    R[:, 1, 1], R[:, 1, 2], R[:, 2, 1], R[:, 2, 2] = c, -s, s, c
    src = pts @ R.transpose(1, 2) + 0.02 * torch.randn(B, N, 3, generator=gen) + 0.1 * torch.randn(B, 1, 3, generator=gen)
    return pts.to(device), src.to(device)


class TinyRegistrar(nn.Module):
    """shared point MLP -> max pool -> FC -> (quaternion, translation); applies the estimated motion to the source."""

    def __init__(self, emb=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Conv1d(3, 64, 1), nn.ReLU(), nn.Conv1d(64, 128, 1), nn.ReLU(), nn.Conv1d(128, emb, 1))
        self.fc = nn.Sequential(nn.Linear(2 * emb, 256), nn.ReLU(), nn.Linear(256, 7))

    def embed(self, x):
        return self.mlp(x.transpose(1, 2)).max(dim=2).values

    def forward(self, template, source):
        pose = self.fc(torch.cat([self.embed(template), self.embed(source)], dim=1))
        q = torch.nn.functional.normalize(pose[:, :4] + torch.tensor([1.0, 0, 0, 0], device=pose.device), dim=1)
        w, x, y, z = q.unbind(1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)
        return source @ R.transpose(1, 2) + pose[:, None, 4:]


def run(batch=32, points=2048, slices=512, steps=20, seed=0, verbose=True):
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = TinyRegistrar().to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    criteria = shw.SSWCriterion(dev, p=2, num_projections=slices)
    template, source = synthetic_batch(batch, points, gen, dev)
    template = template - template.mean(1, keepdim=True)          # train_W_COS.py:167-168
    source = source - source.mean(1, keepdim=True)
    losses, times = [], []
    for it in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        moved = model(template, source)
        loss, _, _ = criteria(template, moved, train_or_test="train")
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        losses.append(loss.item())
        if verbose:
            print(f"step {it:3d}  loss {losses[-1]:.6f}  {1e3 * times[-1]:.2f} ms")
    return losses, times


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--slices", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    losses, times = run(a.batch, a.points, a.slices, a.steps)
    print(f"median step {1e3 * sorted(times)[len(times) // 2]:.2f} ms; loss {losses[0]:.5f} -> {losses[-1]:.5f}")
