import sys, time, torch, cProfile, pstats, io
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.randn(1200, 3, generator=g).to(dev).requires_grad_(True)
y = torch.nn.functional.normalize(torch.randn(1200, 3, generator=g), dim=-1).to(dev)
opt = torch.optim.Adam([x], lr=1e-3)
def t(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("draw_directions      %.4f ms" % t(lambda: shw.draw_directions(100, dev, d=3)))
U = shw.draw_directions(100, dev, d=3)
with torch.no_grad():
    print("sliced_cost no grad  %.4f ms" % t(lambda: shw.sliced_cost(x, y, U, p=2)))
print("sliced_cost grad fwd %.4f ms" % t(lambda: shw.sliced_cost(x, y, U, p=2)))
def fb():
    x.grad = None
    shw.sliced_cost(x, y, U, p=2).backward()
print("fwd+bwd              %.4f ms" % t(fb))
def step():
    opt.zero_grad()
    loss = shw.sliced_wasserstein_sphere(x, y, 100, dev, p=2)
    loss.backward(retain_graph=True)
    opt.step()
print("full step            %.4f ms" % t(step))
print("adam step only       %.4f ms" % t(lambda: opt.step()))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
