"""Multi-GPU evaluation of the sliced loss: one process per GPU, `torch.distributed` ("nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for the tests).

Every (pair, slice) is independent until the final mean / sum, so the path shards with
NO data-path collective.  What is exchanged is one sum all-reduce of the (B,) per-pair partial losses
(B floats: latency-bound on any fabric) and -- only when the inputs are replicated and the caller wants
the full gradient on every rank -- one sum all-reduce of the two (B,n,3) gradient tensors.

Two partitions of the flattened (pair x slice) work:
  "pairs"  : rank r owns a contiguous block of pairs and all of their slices.
  "slices" : rank r owns a contiguous block of slice indices of every pair (BASELINE config 4).
The local evaluator is injectable so that the partition / collective logic is testable on CPU with the
oracle standing in for the HIP kernels; the default evaluator is the HIP op and nothing else.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous, balanced partition of range(total): the first (total % world) ranks get one extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class _SumAcrossRanks(torch.autograd.Function):
    """y = sum over ranks of x.  Every rank then derives the same replicated loss from y, so the gradient of
    that loss w.r.t. the local x is the upstream gradient unchanged."""

    @staticmethod
    def forward(ctx, x, group):
        y = x.clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


class _ReplicatedInput(torch.autograd.Function):
    """Identity on a tensor that every rank holds identically; its gradient is the SUM of the ranks' partial
    gradients (each rank only back-propagates through the work it owns)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def _default_local_fn(Xs, Xt, Us, p):
    from .ssw import ssw_pair_losses
    return ssw_pair_losses(Xs, Xt, Us, p)


def sharded_pair_losses(Xs, Xt, Us, p=2, mode="pairs", group=None, sync_input_grads=True, local_fn=None):
    """Replicated inputs Xs (B,n,3), Xt (B,m,3), Us (B,L,3,2) or (L,3,2) on every rank -> (B,) per-pair losses
    on every rank, identical to the single-process result.

    `local_fn(Xs, Xt, Us, p) -> (b,)` evaluates the mean over the given slices for the given pairs."""
    if local_fn is None:
        local_fn = _default_local_fn
    if not dist.is_initialized():
        return local_fn(Xs, Xt, Us, p)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = Xs.shape[0]
    shared = Us.dim() == 3
    L = Us.shape[-3]
    if sync_input_grads:
        if Xs.requires_grad:
            Xs = _ReplicatedInput.apply(Xs, group)
        if Xt.requires_grad:
            Xt = _ReplicatedInput.apply(Xt, group)
    partial = torch.zeros(B, dtype=Xs.dtype, device=Xs.device)
    if mode == "pairs":
        lo, hi = shard_bounds(B, world, rank)
        if hi > lo:
            mine = local_fn(Xs[lo:hi], Xt[lo:hi], Us if shared else Us[lo:hi], p)
            partial = torch.cat([partial[:lo], mine, partial[hi:]])
    elif mode == "slices":
        lo, hi = shard_bounds(L, world, rank)
        if hi > lo:
            Ul = (Us[lo:hi] if shared else Us[:, lo:hi]).contiguous()
            partial = local_fn(Xs, Xt, Ul, p) * ((hi - lo) / L)
    else:
        raise ValueError("mode must be 'pairs' or 'slices'")
    if hi == lo:
        # A rank that owns no work (world > B in "pairs" mode, world > L in "slices" mode) must still take part in
        # the backward all-reduces of _ReplicatedInput, or the other ranks wait for it until the RCCL timeout:
        # a zero-valued dependency on the wrapped inputs keeps it in the graph with a zero gradient.
        partial = partial + (Xs[:0].sum() + Xt[:0].sum())
    return _SumAcrossRanks.apply(partial, group)


def sharded_sliced_cost(Xs, Xt, Us, p=2, mode="pairs", group=None, sync_input_grads=True, local_fn=None):
    """Batched reference value (sum over pairs of the per-pair slice mean, shape [1]; _fast.py:291-293) with the
    work sharded across the ranks of `group`."""
    return sharded_pair_losses(Xs, Xt, Us, p, mode, group, sync_input_grads, local_fn).sum().reshape(1)


def local_data_loss(Xs_local, Xt_local, Us_local, p=2, group=None, local_fn=None):
    """Data-parallel form used by bench.py and by a DDP trainer: every rank holds ITS OWN pairs (weak scaling).
    Returns the global batched loss (sum over all ranks' pairs, shape [1]); the only collective is one
    all-reduce of that scalar.  Gradients stay local (each rank owns its pairs' gradients)."""
    if local_fn is None:
        local_fn = _default_local_fn
    mine = local_fn(Xs_local, Xt_local, Us_local, p).sum().reshape(1)
    if not dist.is_initialized():
        return mine
    return _SumAcrossRanks.apply(mine, group)
