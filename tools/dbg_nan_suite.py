"""Development aid: run the GPU suite in-process with the autograd Function of the sliced loss wrapped so that a
non-finite output on finite inputs dumps every operand to gpurun_out/nan_dump_*.pt."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw  # noqa: E402

ssw = sys.modules[shw.__name__ + ".ssw"] if (shw.__name__ + ".ssw") in sys.modules else __import__("importlib").import_module(shw.__name__ + ".ssw")
F = ssw._PairLosses
orig_fwd, orig_bwd = F.forward, F.backward
count = [0]


def finite(*ts):
    if torch.cuda.is_current_stream_capturing():
        return True
    return all(t is None or bool(torch.isfinite(t).all()) for t in ts)


def fwd(ctx, Xs, Xt, Us, p, shared, wu, wv):
    out = orig_fwd(ctx, Xs, Xt, Us, p, shared, wu, wv)
    if finite(Xs, Xt, Us, wu, wv) and not finite(out[0], out[1]):
        count[0] += 1
        print("NONFINITE FORWARD", Xs.shape, Xt.shape, Us.shape, p, flush=True)
        torch.save({"Xs": Xs.cpu(), "Xt": Xt.cpu(), "Us": Us.cpu(), "p": p, "loss": out[0].cpu(), "cost": out[1].cpu()},
                   f"gpurun_out/nan_dump_fwd{count[0]}.pt")
    return out


def bwd(ctx, g, a, b):
    saved = ctx.saved_tensors
    out = orig_bwd(ctx, g, a, b)
    if finite(saved[0], saved[1], saved[2], g) and not finite(out[0], out[1]) and float(saved[1].abs().sum()) != 0.0 and float(saved[0].abs().sum()) != 0.0:
        count[0] += 1
        print("NONFINITE BACKWARD dims", ctx.dims, "coef finite", finite(saved[3]), finite(saved[4]), flush=True)
        torch.save({"Xs": saved[0].cpu(), "Xt": saved[1].cpu(), "Us": saved[2].cpu(), "coef_s": saved[3].cpu(),
                    "coef_t": saved[4].cpu(), "g": g.cpu(), "gxs": out[0].cpu(), "gxt": out[1].cpu(), "dims": ctx.dims},
                   f"gpurun_out/nan_dump_bwd{count[0]}.pt")
    return out


F.forward = staticmethod(fwd)
F.backward = staticmethod(bwd)
os.makedirs("gpurun_out", exist_ok=True)
rc = pytest.main(["tests", "-m", "gpu", "-q"] + sys.argv[1:])
print("dumps:", count[0], "rc", rc)
