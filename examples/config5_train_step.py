#!/usr/bin/env python3
"""BASELINE config 5 at its stated shape: "train_W_COS.py end-to-end: ModelNet40-shaped synthetic clouds, N=2048,
registration network forward+backward on 1 GPU".

The loop is train_W_COS.py:155-175 (`train_one_epoch`): mean-centre both clouds, `model(template, source, 8)`,
`criteria(template, output['transformed_source'], train_or_test="train")`, `loss.backward()`, Adam step.

* network: a PCRNet-SHAPED iterative pose regressor written here from the description of models/pcrnet.py:7-62 and
  models/mlp_architecture.py (the reference's model is out of scope -- stock nn layers -- but the workload must have
  its shape): shared point MLP Conv1d 3->64->64->64->128->1024 + ReLU, max-pool over points, 5 FC layers
  2048->1024->1024->512->512->256 and a 7-d pose head (quaternion + translation), `iteration_num = 8` refinements,
  each one re-embedding the moved source;
* criterion: the live trainer criterion's shape (s2_wasserstein.py:234-262): phi-max inner loop (`phi_max_iter`
  ascent steps with the |norm - 1| regulariser) and the final distance, with the sliced loss `SlicedSphereW` in the
  CSW slot (`--criterion csw`), or the dormant batched wrapper `max_spherical_wassersten_distance_fast`
  (`--criterion ssw_fast`, _fast.py:346-380);
* phi: a small planar flow (three x + u tanh(w.x + b) layers; the reference's flows come from a vendored package);
* data: ModelNet-shaped synthetic clouds (no dataset offline): unit-scale random surfaces, random rigid motion
  <= 45 degrees, sigma 0.02 noise (train_W_COS.py:291-295 defaults).

    python examples/config5_train_step.py [--batch 32] [--points 2048] [--slices 512] [--steps 10]
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


def quat_rotate(q, v):
    """v (B,N,3) rotated by unit quaternions q (B,4), (w, x, y, z):  v + 2 (w (u x v) + u x (u x v))."""
    u = q[:, None, 1:].expand_as(v)
    uv = torch.cross(u, v, dim=-1)
    return v + 2 * (q[:, None, :1] * uv + torch.cross(u, uv, dim=-1))


class PointMLP(nn.Module):
    """shared per-point MLP 3 -> 64 -> 64 -> 64 -> 128 -> emb, ReLU after every layer (PointNet trunk)"""

    def __init__(self, emb=1024):
        super().__init__()
        dims = [3, 64, 64, 64, 128, emb]
        layers = []
        for a, b in zip(dims, dims[1:]):
            layers += [nn.Conv1d(a, b, 1), nn.ReLU()]
        self.net = nn.Sequential(*layers)
        self.emb = emb

    def forward(self, x):                       # (B,N,3) -> (B,emb): max-pool over the points
        return self.net(x.transpose(1, 2)).max(dim=2).values


class IterativePoseRegressor(nn.Module):
    """PCRNet-shaped: global features of template and (moved) source -> 5 FC layers -> 7-d pose; the estimated
    motion is applied to the source and the step repeats `iteration_num` times."""

    def __init__(self, emb=1024):
        super().__init__()
        self.features = PointMLP(emb)
        dims = [2 * emb, 1024, 1024, 512, 512, 256]
        layers = []
        for a, b in zip(dims, dims[1:]):
            layers += [nn.Linear(a, b), nn.ReLU()]
        layers.append(nn.Linear(256, 7))
        self.head = nn.Sequential(*layers)

    def forward(self, template, source, iteration_num=8):
        ft = self.features(template)
        for _ in range(iteration_num):
            pose = self.head(torch.cat([ft, self.features(source)], dim=1))
            q = torch.nn.functional.normalize(pose[:, :4], dim=1)
            source = quat_rotate(q, source) + pose[:, None, 4:]
        return {"transformed_source": source}


class PlanarFlow(nn.Module):
    """phi: n planar layers x + u tanh(w.x + b) on R^3 (the shape of normflows' Planar flow)"""

    def __init__(self, n_layers=3, dim=3):
        super().__init__()
        self.u = nn.Parameter(0.1 * torch.randn(n_layers, dim))
        self.w = nn.Parameter(0.5 * torch.randn(n_layers, dim))
        self.b = nn.Parameter(torch.zeros(n_layers))

    def forward(self, x):
        for u, w, b in zip(self.u, self.w, self.b):
            x = x + u * torch.tanh((x * w).sum(-1, keepdim=True) + b)   # (x @ w would go to a rocBLAS gemv: 0.5 ms per call)
        return x


def build(slices=512, criterion="csw", phi_max_iter=1, seed=0, dev=None, lr=1e-3):
    dev = dev or torch.device("cuda", 0)
    torch.manual_seed(seed)
    model = IterativePoseRegressor().to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1.4e-8)
    phi = PlanarFlow().to(dev)
    phi_op = torch.optim.Adam(phi.parameters(), lr=9.2e-5, weight_decay=1.4e-8)
    if criterion == "csw":      # train_W_COS.py:393,404 with the sliced loss in the CSW slot
        crit = shw.max_cos_disimilarity_wassersten_distance(phi=phi, CSW=shw.SlicedSphereW(dev, p=2, num_projections=slices),
                                                            phi_op=phi_op, lam=1.3e-5, max_iter=phi_max_iter, device=dev)
    elif criterion == "ssw_fast":
        crit = shw.max_spherical_wassersten_distance_fast(slices, phi, shw.sliced_wasserstein_sphere_fast, phi_op, p=2,
                                                          max_iter=phi_max_iter, device=dev)
    elif criterion == "plain":  # no phi: the sliced loss directly on the clouds
        crit = shw.SSWCriterion(dev, p=2, num_projections=slices)
    else:
        raise ValueError(criterion)
    return model, opt, crit


def train_step(model, opt, crit, template, source, iteration_num=8):
    """train_W_COS.py:163-174"""
    opt.zero_grad()
    source = source - source.mean(1, keepdim=True)
    template = template - template.mean(1, keepdim=True)
    out = model(template, source, iteration_num)
    loss, _, _ = crit(template, out["transformed_source"], train_or_test="train")
    loss = loss.sum()
    loss.backward()
    opt.step()
    return loss


def run(batch=32, points=2048, slices=512, steps=10, seed=0, verbose=True, criterion="csw", phi_max_iter=1,
        iteration_num=8, fresh_batches=False):
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(seed)
    model, opt, crit = build(slices, criterion, phi_max_iter, seed, dev)
    template, source = synthetic_batch(batch, points, gen, dev)
    losses, times = [], []
    for it in range(steps):
        if fresh_batches and it:
            template, source = synthetic_batch(batch, points, gen, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = train_step(model, opt, crit, template, source, iteration_num)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        losses.append(loss.item())
        if verbose:
            print(f"step {it:3d}  loss {losses[-1]:.6f}  {1e3 * times[-1]:.2f} ms")
    return losses, times


def ssw_share(batch, points, slices, evaluations, reps=10):
    """Time of the sliced-loss part of one step in isolation: `evaluations` forward+backward passes of
    SlicedSphereW at the step's shape (phi_max_iter inner evaluations + the final one)."""
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(1)
    a, b = synthetic_batch(batch, points, gen, dev)
    csw = shw.SlicedSphereW(dev, p=2, num_projections=slices)
    a.requires_grad_(True)
    for _ in range(3):
        csw(a, b).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for _ in range(evaluations):
            csw(a, b).backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--slices", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--criterion", default="csw", choices=["csw", "ssw_fast", "plain"])
    ap.add_argument("--phi-max-iter", type=int, default=1)
    a = ap.parse_args()
    losses, times = run(a.batch, a.points, a.slices, a.steps, criterion=a.criterion, phi_max_iter=a.phi_max_iter)
    med = sorted(times[2:] or times)[len(times[2:] or times) // 2]
    share = ssw_share(a.batch, a.points, a.slices, a.phi_max_iter + 1)
    print(f"median step {1e3 * med:.2f} ms (sliced-loss part {1e3 * share:.2f} ms = {100 * share / med:.0f} %); "
          f"loss {losses[0]:.5f} -> {losses[-1]:.5f}")
