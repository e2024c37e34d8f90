"""world_size-2 `gloo` tests (CPU) of the multi-GPU decomposition: partition bounds, the single sum
all-reduce of per-pair partials, gradient bookkeeping.  The CPU oracle stands in for the HIP kernels as the
injected local evaluator -- the product's default evaluator is the HIP op and is exercised by the gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_local(Xs, Xt, Us, p):
    from oracle import ref_mirror
    shared = Us.dim() == 3
    if Xs.shape[0] == 0:
        return Xs.new_zeros(0)
    return torch.stack([ref_mirror.per_slice_costs(Xs[b], Xt[b], Us if shared else Us[b], p).mean()
                        for b in range(Xs.shape[0])])


def _worker(rank, world, port, mode, shared, out_dir, B=3, L=6):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import shw_amd
    from shw_amd import dist as sd
    g = torch.Generator().manual_seed(99)
    n = 64
    x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).requires_grad_(True)
    y = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).requires_grad_(True)
    U = torch.linalg.qr(torch.randn(*((L,) if shared else (B, L)), 3, 2, generator=g))[0]
    w = torch.tensor([1.0, -2.0, 0.5])[:B]
    pair = sd.sharded_pair_losses(x, y, U, 2, mode=mode, local_fn=_oracle_local)
    (pair * w).sum().backward()
    total = sd.sharded_sliced_cost(x.detach(), y.detach(), U, 2, mode=mode, local_fn=_oracle_local)
    # weak-scaling form: each rank brings its own pair
    lo, hi = sd.shard_bounds(B, world, rank)
    own = sd.local_data_loss(x.detach()[lo:hi], y.detach()[lo:hi], U if shared else U[lo:hi], 2, local_fn=_oracle_local)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pair=pair.detach().numpy(), gx=x.grad.numpy(),
             gy=y.grad.numpy(), total=total.numpy(), own=own.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["pairs", "slices"])
@pytest.mark.parametrize("shared", [False, True])
def test_two_rank_sharding_matches_single_process(tmp_path, mode, shared):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, shared, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(99)
    B, n, L = 3, 64, 6
    x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).requires_grad_(True)
    y = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).requires_grad_(True)
    U = torch.linalg.qr(torch.randn(*((L,) if shared else (B, L)), 3, 2, generator=g))[0]
    w = torch.tensor([1.0, -2.0, 0.5])
    ref = _oracle_local(x, y, U, 2)
    (ref * w).sum().backward()
    for rank in range(world):
        r = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert np.allclose(r["pair"], ref.detach().numpy(), rtol=2e-6)
        assert np.allclose(r["total"], ref.detach().sum().numpy(), rtol=2e-6)
        assert np.allclose(r["own"], ref.detach().sum().numpy(), rtol=2e-6)
        assert np.allclose(r["gx"], x.grad.numpy(), rtol=1e-4, atol=1e-9)
        assert np.allclose(r["gy"], y.grad.numpy(), rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("mode", ["pairs", "slices"])
def test_a_rank_without_work_still_joins_the_backward_collectives(tmp_path, mode):
    """world 3 with B = 2 pairs ("pairs" mode) resp. L = 2 slices ("slices" mode): rank 2 owns nothing.  Its
    backward must still enter the input-gradient all-reduce (it used to skip it and the other
    ranks hung until the collective timed out); every rank ends with the full single-process gradient."""
    world, B, L = 3, 2, 2
    mp.spawn(_worker, args=(world, _free_port(), mode, False, str(tmp_path), B, L), nprocs=world, join=True)
    g = torch.Generator().manual_seed(99)
    n = 64
    x = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).requires_grad_(True)
    y = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1).requires_grad_(True)
    U = torch.linalg.qr(torch.randn(B, L, 3, 2, generator=g))[0]
    ref = _oracle_local(x, y, U, 2)
    (ref * torch.tensor([1.0, -2.0])).sum().backward()
    for rank in range(world):
        r = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert np.allclose(r["pair"], ref.detach().numpy(), rtol=2e-6)
        assert np.allclose(r["gx"], x.grad.numpy(), rtol=1e-4, atol=1e-9)
        assert np.allclose(r["gy"], y.grad.numpy(), rtol=1e-4, atol=1e-9)


def test_shard_bounds_partition_everything_once():
    sys.path.insert(0, ROOT)
    from shw_amd.dist import shard_bounds
    for total in (0, 1, 5, 64, 513):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_bounds(total, world, r)
                assert 0 <= lo <= hi <= total
                cover += list(range(lo, hi))
            assert cover == list(range(total))
            sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
