import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw
dev = "cuda"
g = torch.Generator().manual_seed(0)
N, L = 1200, 100
target = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
ev = (torch.randn(N, 3, generator=g) * 0.5).to(dev).requires_grad_(True)
U = shw.draw_directions(L, dev)
def t(fn, K=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / K * 1e6
print("draw_directions            %.1f us" % t(lambda: shw.draw_directions(L, dev)))
print("forward (no grad)          %.1f us" % t(lambda: shw.sliced_cost(ev.detach(), target, U, p=2)))
print("forward (grad-enabled)     %.1f us" % t(lambda: shw.sliced_cost(ev, target, U, p=2)))
def fb():
    ev.grad = None
    shw.sliced_cost(ev, target, U, p=2).backward()
print("forward + backward         %.1f us" % t(fb))
opt = torch.optim.Adam([ev], lr=1e-2)
def full():
    loss = shw.sliced_wasserstein_sphere(ev, target, L, device=dev, p=2)
    opt.zero_grad(); loss.backward(); opt.step()
print("full step                  %.1f us" % t(full))
