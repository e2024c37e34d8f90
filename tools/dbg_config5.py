"""Development aid: repeat the config-5 miniature with NaN-poisoned allocator memory and stage-by-stage finiteness
checks, to localise an intermittent non-finite value."""
import importlib.util
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("c5", "examples/config5_train_step.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
import shw_amd as shw  # noqa: E402

dev = torch.device("cuda", 0)
os.makedirs("gpurun_out", exist_ok=True)


def poison():
    blocks = [torch.full((s,), float("nan"), device=dev) for s in (1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10, 256, 64, 8) for _ in range(4)]
    del blocks


def fin(t):
    return bool(torch.isfinite(t).all())


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
found = 0
for rep in range(reps):
    gen = torch.Generator().manual_seed(rep)
    torch.manual_seed(rep)
    model = mod.TinyRegistrar().to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    template, source = mod.synthetic_batch(8, 1024, gen, dev)
    template = template - template.mean(1, keepdim=True)
    source = source - source.mean(1, keepdim=True)
    for it in range(25):
        poison()
        opt.zero_grad()
        moved = model(template, source)
        moved.retain_grad()
        U = shw.draw_directions(128, dev, batch=8, d=3)
        pair, cost, shift = shw.ssw_pair_losses(template, moved, U, 2, return_slices=True)
        loss = torch.pow(pair, 0.5).sum() / 8
        loss.backward()
        g = moved.grad
        pbad = [n for n, p in model.named_parameters() if p.grad is not None and not fin(p.grad)]
        ok = (fin(moved), fin(U), fin(cost), fin(pair), fin(loss), fin(g), not pbad)
        if not all(ok):
            found += 1
            print("rep", rep, "it", it, "moved/U/cost/pair/loss/grad/params finite:", ok, pbad, "pair", pair.tolist(), flush=True)
            torch.save({"template": template.cpu(), "moved": moved.detach().cpu(), "U": U.cpu(), "cost": cost.cpu(),
                        "pair": pair.detach().cpu(), "grad": g.cpu()}, f"gpurun_out/dbg5_bad{found}.pt")
            break
        opt.step()
        pb = [n for n, p in model.named_parameters() if not fin(p)]
        if pb:
            found += 1
            print("rep", rep, "it", it, "params non-finite after Adam step", pb, "grad absmax", float(g.abs().max()), flush=True)
            break
    if found >= 3:
        break
print("reps", rep + 1, "found", found)
