import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw
g = torch.Generator().manual_seed(0)
for (B, N) in ((64, 256), (64, 1024), (64, 2048)):
    x = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1).cuda()
    y = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1).cuda()
    shw.sinkhorn_pair_costs(x, y, 0.01, 100); torch.cuda.synchronize()
    t = time.perf_counter(); c, _, _ = shw.sinkhorn_pair_costs(x, y, 0.01, 100); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"B={B} N=M={N} eps=0.01 100 iterations: {1e3*dt:.2f} ms  ({B*N*N*200/dt:.3e} pair-updates/s)  cost[0]={c[0].item():.6f}")
