"""Runs under torch.distributed.run on the GPU box (world size 1: one GPU per box): exercises
shw_amd.dist with its DEFAULT evaluator (the HIP op) over the nccl (= RCCL) backend."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import shw_amd as shw  # noqa: E402


def main():
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl")
    g = torch.Generator().manual_seed(17)
    B, n, L = 4, 256, 16
    x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).cuda()
    y = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).cuda()
    U = shw.stiefel_frames(torch.randn(B, L, 3, 2, generator=g).cuda())
    ref_x = x.clone().requires_grad_(True)
    ref = shw.ssw_pair_losses(ref_x, y, U, 2)
    ref.sum().backward()
    for mode in ("pairs", "slices"):
        xs = x.clone().requires_grad_(True)
        got = shw.dist.sharded_pair_losses(xs, y, U, 2, mode=mode)
        got.sum().backward()
        if dist.get_world_size() == 1:
            # one rank owns every (pair, slice): the sharded evaluation IS the single-process one, bit for bit
            assert torch.equal(got, ref), mode
            assert torch.equal(xs.grad, ref_x.grad), mode
        assert torch.allclose(got, ref, rtol=1e-6), mode
        assert torch.allclose(xs.grad, ref_x.grad, rtol=1e-5, atol=1e-9), mode
    tot = shw.dist.sharded_sliced_cost(x, y, U, 2)
    own = shw.dist.local_data_loss(x, y, U, 2)
    assert tuple(tot.shape) == (1,) and torch.allclose(tot, ref.sum().reshape(1), rtol=1e-6)
    assert torch.allclose(own, tot, rtol=1e-6)
    dist.destroy_process_group()
    print("DIST_NCCL_OK")


if __name__ == "__main__":
    main()
