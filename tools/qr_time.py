import torch, time
dev="cuda"
for shape in [(512,3,2),(32,512,3,2),(64,512,3,2)]:
    Z=torch.randn(shape,device=dev)
    torch.linalg.qr(Z); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(3): Q,_=torch.linalg.qr(Z)
    torch.cuda.synchronize(); print(shape,"gpu qr ms",(time.perf_counter()-t)/3*1e3)
    Zc=Z.cpu(); t=time.perf_counter(); Qc,_=torch.linalg.qr(Zc); print(shape,"cpu qr ms",(time.perf_counter()-t)*1e3, "max diff gpu-cpu", (Q.cpu()-Qc).abs().max().item())
