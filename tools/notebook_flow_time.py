"""The notebooks' gradient-flow step (Flow_cube.ipynb:1372-1395): evolving cloud of 1200 points, 100 projections,
loss = sliced_wasserstein_sphere(evolving, target, 100, device, p=2); loss.backward(); Adam step.
The saved notebook outputs record 0.531 s per 5 such steps (unstated hardware, includes one
ot.emd2 probe).  This script times the same step on the HIP path."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw

dev = "cuda"
g = torch.Generator().manual_seed(0)
N, L = 1200, 100
target = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
evolving = (torch.randn(N, 3, generator=g) * 0.5).to(dev).requires_grad_(True)
opt = torch.optim.Adam([evolving], lr=1e-2)
def step():
    loss = shw.sliced_wasserstein_sphere(evolving, target, L, device=dev, p=2)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss
for _ in range(20): step()
torch.cuda.synchronize(); t = time.perf_counter()
K = 200
for _ in range(K): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
print(f"gradient-flow step N={N} L={L}: {1e3*dt:.3f} ms/step ({5*dt:.4f} s per 5 steps; notebook record 0.531 s), loss {l.item():.6f}")
