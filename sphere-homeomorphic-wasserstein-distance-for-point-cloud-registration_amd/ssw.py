"""Host-side mirror of the reference's spherical sliced-Wasserstein call shapes on top
of the HIP C-ABI library.  Same names, argument order and return shapes as
/root/reference/Point_Cloud_Resistration/losses/max_spherical_sliced_w.py (per pair) and
max_spherical_sliced_w_fast.py (batched):

    sliced_wasserstein_sphere(Xs, Xt, num_projections, device, u_weights=None, v_weights=None, p=2)  :289-310
    sliced_wasserstein_sphere_fast(...)                                                      _fast.py:298-319
    sliced_cost(Xs, Xt, Us, p=2, u_weights=None, v_weights=None)                 :251-286 / _fast.py:258-295

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every number is produced by
the hand-written gfx950 kernels.  CPU tensors are rejected: there is no fallback path.
"""
from __future__ import annotations

import threading
from collections import OrderedDict

import torch

from . import _lib


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _check_cloud(name, t):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the MI355X HIP path needs device tensors (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if t.shape[-1] != 3:
        raise ValueError(f"{name} must have 3 coordinates per point, got shape {tuple(t.shape)}")


def draw_directions(num_projections, device, batch=None, d=3):
    """Uniform samples on the Stiefel manifold V_{d,2} from the GLOBAL torch generator, consumed exactly
    like the reference (randn of (L,d,2) resp. (B,L,d,2) on `device`, then reduced QR; :307-308,
    _fast.py:317-318)."""
    shape = (num_projections, d, 2) if batch is None else (batch, num_projections, d, 2)
    Z = torch.randn(shape, device=device)
    return stiefel_frames(Z)


def stiefel_frames(Z):
    """Reduced QR `U, _ = torch.linalg.qr(Z)` of (..., 3, 2) Gaussian matrices.  On the device this is the HIP
    kernel `shw_stiefel_frames` (LAPACK's Householder steps and sign convention, one thread per matrix):
    torch.linalg.qr itself takes ~0.94 s for the 32 768 frames of config 3 on MI355X, and differs from the CPU
    LAPACK result by up to 6e-6; the kernel takes microseconds and agrees with LAPACK to fp32 rounding.  CPU
    tensors (tests of the generator stream) go through torch.linalg.qr."""
    if not Z.is_cuda:
        return torch.linalg.qr(Z)[0]
    if Z.dtype != torch.float32 or tuple(Z.shape[-2:]) != (3, 2):
        raise TypeError("stiefel_frames expects float32 (..., 3, 2)")
    lib = _lib.load()
    Zc = Z.contiguous()
    U = torch.empty_like(Zc)
    with torch.cuda.device(Z.device):
        _lib.check(lib.shw_stiefel_frames(Zc.data_ptr(), Zc.numel() // 6, U.data_ptr(), _stream_ptr(Z.device)),
                   "shw_stiefel_frames")
    return U


class SSWWorkspace:
    """Persistent device buffers of one problem shape (B, n, m, L): the per-slice costs and shifts / cuts and --
    for training -- the two coefficient scratch rows that the slice kernel writes and the point-gradient kernel
    streams (2 x 268 MB at config 3).  Instances are leased from a small pool keyed by (device, stream,
    shape) and come back when the evaluation that used them is over: at once for a loss-only call, and when the
    autograd node of a training call dies (after its backward, or when its graph is dropped) -- see `_Lease`.
    A phi-max inner loop (max_iter + 1 evaluations of one shape per step, _fast.py:359-377) thus alternates
    between two sets of buffers instead of asking the allocator for five tensors per evaluation, and nothing is
    zero-filled.  The user-visible results (per-pair losses, total) are NOT pool memory: they are one fresh
    (B+2)-float tensor per call, so holding on to a loss value across later calls is safe.
    Work on the buffers is ordered by the stream they were leased on (the stream is part of the key).  While
    a HIP graph is being captured the pool is bypassed: buffers then belong to the graph's private memory.
    Idle buffers are capped per shape (2) and in total (MAX_IDLE_BYTES, least recently used shapes evicted first); the
    pool is guarded by a lock."""

    _pools: "OrderedDict" = OrderedDict()      # key -> idle workspaces, least recently used key first
    _lock = threading.Lock()                   # autograd may release leases from its own threads
    MAX_IDLE_PER_KEY = 2
    # bounded per key only, every distinct shape pinned up to 1 GB (a training workspace at config 3 is 537 MB)
    # that torch.cuda.empty_cache() cannot reclaim.  Idle buffers of ALL keys together stay below this many bytes: the
    # least recently used shapes go back to torch's caching allocator first.
    MAX_IDLE_BYTES = 4 << 30

    def __init__(self, key):
        dev, _stream, B, n, m, L, with_coef = key
        self.key = key
        self.pooled = True
        self.slice_cost = torch.empty(B * L, dtype=torch.float32, device=dev)
        self.slice_aux = torch.empty(B * L, dtype=torch.int32, device=dev)     # shift k* / median level / cut bits
        self.coef_s = torch.empty(B * L * n, dtype=torch.float32, device=dev) if with_coef else None
        self.coef_t = torch.empty(B * L * m, dtype=torch.float32, device=dev) if with_coef else None
        self.nbytes = 8 * B * L + (4 * B * L * (n + m) if with_coef else 0)

    @classmethod
    def lease(cls, dev, B, n, m, L, with_coef):
        key = (torch.device(dev), _stream_ptr(dev), B, n, m, L, bool(with_coef))
        if torch.cuda.is_current_stream_capturing():
            ws = cls(key)
            ws.pooled = False
            return ws
        with cls._lock:
            idle = cls._pools.get(key)
            if idle:
                cls._pools.move_to_end(key)
                return idle.pop()
        return cls(key)

    def release(self):
        if not self.pooled:
            return
        cls = SSWWorkspace
        with cls._lock:
            idle = cls._pools.setdefault(self.key, [])
            cls._pools.move_to_end(self.key)
            if len(idle) < cls.MAX_IDLE_PER_KEY and all(ws is not self for ws in idle):
                idle.append(self)
            total = sum(ws.nbytes for lst in cls._pools.values() for ws in lst)
            while total > cls.MAX_IDLE_BYTES and cls._pools:
                old_key = next(iter(cls._pools))                  # least recently used shape
                lst = cls._pools[old_key]
                if lst:
                    total -= lst.pop().nbytes                     # back to torch's caching allocator
                if not lst:
                    del cls._pools[old_key]

    @classmethod
    def idle_bytes(cls):
        with cls._lock:
            return sum(ws.nbytes for lst in cls._pools.values() for ws in lst)

    @classmethod
    def clear(cls):
        """Drop every idle buffer (hands the memory back to torch's caching allocator)."""
        with cls._lock:
            cls._pools.clear()


class _Lease:
    """Ties a leased workspace to the lifetime of the autograd node that reads it in backward: the node keeps the
    lease, and when the node is freed (CPython reference counting: after backward once the loss tensor is
    re-bound or deleted, or when a graph is dropped without backward) the buffers return to the pool.  A graph kept
    alive with retain_graph=True keeps its buffers -- a second backward through it reads intact coefficients."""

    __slots__ = ("ws",)

    def __init__(self, ws):
        self.ws = ws

    def __del__(self):
        try:
            self.ws.release()
        except Exception:        # interpreter shutdown
            pass


class _PairLosses(torch.autograd.Function):
    """(B,n,3), (B,m,3), dirs -> (B,) per-pair mean over slices of the circular OT cost W_p^p, their sum (1,), and
    on request the (B,L) per-slice costs and shifts."""

    @staticmethod
    def forward(ctx, Xs, Xt, Us, p, shared_dirs, wu, wv, need_grad, want_slices):
        lib = _lib.load()
        B, n, _ = Xs.shape
        m = Xt.shape[1]
        L = Us.shape[-3]
        dev = Xs.device
        Xs_c, Xt_c, Us_c = Xs.contiguous(), Xt.contiguous(), Us.contiguous()
        stride = 0 if shared_dirs else L * 6
        ws = SSWWorkspace.lease(dev, B, n, m, L, need_grad)
        out = torch.empty(B + 2, dtype=torch.float32, device=dev)        # [pair_loss (B) | total (2)]
        stream = _stream_ptr(dev)
        weighted = wu is not None or wv is not None
        general = weighted or (float(p) != 1.0 and n != m)      # p == 1 with uniform weights: level-median kernel
        cs = ws.coef_s.data_ptr() if need_grad else None
        ct = ws.coef_t.data_ptr() if need_grad else None
        with torch.cuda.device(dev):
            if general:
                # n != m and / or weights: the reference's bisection over the cut, followed step for step

                def wargs(w, cnt):
                    if w is None:
                        return None, 0
                    return w.data_ptr(), (0 if w.dim() == 1 else cnt)
                wu_p, wu_s = wargs(wu, n)
                wv_p, wv_s = wargs(wv, m)
                _lib.check(lib.shw_ssw_forward_general(
                    Xs_c.data_ptr(), Xt_c.data_ptr(), Us_c.data_ptr(), wu_p, wv_p, wu_s, wv_s, B, n, m, L, stride,
                    float(p), ws.slice_cost.data_ptr(), ws.slice_aux.data_ptr(), cs, ct, stream),
                    "shw_ssw_forward_general")
            elif need_grad:
                _lib.check(lib.shw_ssw_forward_grad(Xs_c.data_ptr(), Xt_c.data_ptr(), Us_c.data_ptr(), B, n, m, L,
                                                    stride, float(p), ws.slice_cost.data_ptr(),
                                                    ws.slice_aux.data_ptr(), cs, ct, stream),
                           "shw_ssw_forward_grad")
            else:
                _lib.check(lib.shw_ssw_forward(Xs_c.data_ptr(), Xt_c.data_ptr(), Us_c.data_ptr(), B, n, m, L, stride,
                                               float(p), ws.slice_cost.data_ptr(), ws.slice_aux.data_ptr(), stream),
                           "shw_ssw_forward")
            _lib.check(lib.shw_ssw_reduce(ws.slice_cost.data_ptr(), B, L, 1.0 / L, out.data_ptr(),
                                          out.data_ptr() + 4 * B, stream), "shw_ssw_reduce")
        cost2d = shift2d = None
        if want_slices:
            cost2d = ws.slice_cost.view(B, L).clone()
            shift2d = torch.zeros(B, L, dtype=torch.int32, device=dev) if general else ws.slice_aux.view(B, L).clone()
            ctx.mark_non_differentiable(cost2d, shift2d)
        if need_grad:
            ctx.save_for_backward(Xs_c, Xt_c, Us_c)
            ctx.lease = _Lease(ws)                  # the coefficients stay put until this node dies
            ctx.dims = (B, n, m, L, stride)
            ctx.set_materialize_grads(False)
        else:
            ws.release()                            # stream-ordered: the next lease is on the same stream
        # (the first pair's loss as a 0-dim OUTPUT: the per-pair call shape returns it, and indexing pair[0] outside would
        #  cost a SelectBackward node -- a zero fill and a copy per step, two of the notebooks' eleven launches)
        return out[:B], out[B:B + 1], cost2d, shift2d, out[0:1].view(())

    @staticmethod
    def backward(ctx, g_pair, g_total, _g_cost, _g_shift, g_first=None):
        lib = _lib.load()
        Xs_c, Xt_c, Us_c = ctx.saved_tensors
        ws = ctx.lease.ws
        B, n, m, L, stride = ctx.dims
        dev = Xs_c.device
        gxs = torch.empty_like(Xs_c)
        gxt = torch.empty_like(Xt_c)
        if g_pair is None and g_total is None and g_first is None:
            return (None,) * 9
        # the upstream gradients go to the kernel as they are: row b is scaled by g_pair[b] + g_total[0] there
        gp = g_pair.to(torch.float32).contiguous() if g_pair is not None else None
        if g_first is not None:                     # gradient of the 0-dim first-pair output: belongs to row 0
            if gp is None and B == 1:
                gp = g_first.to(torch.float32).reshape(1)
            else:
                gp = gp.clone() if gp is not None else torch.zeros(B, dtype=torch.float32, device=dev)
                gp[0] += g_first.to(torch.float32)
        gt = g_total.to(torch.float32).contiguous() if g_total is not None else None
        with torch.cuda.device(dev):
            _lib.check(lib.shw_ssw_backward_points(Xs_c.data_ptr(), Xt_c.data_ptr(), Us_c.data_ptr(),
                                                   ws.coef_s.data_ptr(), ws.coef_t.data_ptr(), B, n, m, L, stride,
                                                   1.0 / L, gp.data_ptr() if gp is not None else None,
                                                   gt.data_ptr() if gt is not None else None, gxs.data_ptr(),
                                                   gxt.data_ptr(), _stream_ptr(dev)),
                       "shw_ssw_backward_points")
        return gxs, gxt, None, None, None, None, None, None, None


def _check_weights(name, w, count, B, dev):
    if w is None:
        return None
    if not isinstance(w, torch.Tensor) or not w.is_cuda or w.dtype != torch.float32:
        raise TypeError(f"{name} must be a float32 device tensor")
    if tuple(w.shape) not in ((count,), (B, count)):
        raise ValueError(f"{name} must have shape ({count},) or ({B}, {count}), got {tuple(w.shape)}")
    return w.detach().contiguous()


def ssw_pair_losses(Xs, Xt, Us, p=2, return_slices=False, u_weights=None, v_weights=None, return_total=False,
                    return_first=False):
    """Core op: batched clouds (B,n,3), (B,m,3); directions (B,L,3,2) or shared (L,3,2); optional weights
    (n,) / (B,n) and (m,) / (B,m).
    Returns (B,) per-pair losses = mean over slices of W_p^p on the slice circle
    [optionally also the (B,L) per-slice costs and optimal shifts; or, with return_total, the shape-[1] sum over
    pairs that the reduction kernel produces in the same launch (the batched reference value, _fast.py:291-293)].
    Scratch (per-slice arrays, gradient coefficients) lives in a pooled SSWWorkspace; results are fresh tensors."""
    _check_cloud("Xs", Xs)
    _check_cloud("Xt", Xt)
    _check_cloud_dirs(Us)
    if Xs.dim() != 3 or Xt.dim() != 3 or Xs.shape[0] != Xt.shape[0]:
        raise ValueError("Xs and Xt must be (B,n,3) and (B,m,3) with the same B")
    shared = Us.dim() == 3
    if not shared and Us.shape[0] != Xs.shape[0]:
        raise ValueError("per-pair directions must be (B,L,3,2)")
    if not (float(p) >= 1.0):
        raise ValueError("p must be >= 1")
    if Us.shape[-3] < 1 or Xs.shape[1] < 1 or Xt.shape[1] < 1:
        raise ValueError("need at least one slice and one point per cloud")
    wu = _check_weights("u_weights", u_weights, Xs.shape[1], Xs.shape[0], Xs.device)
    wv = _check_weights("v_weights", v_weights, Xt.shape[1], Xs.shape[0], Xs.device)
    # evaluation under torch.no_grad() on inputs that still carry requires_grad must not run the
    # training kernel / allocate the coefficient scratch
    need_grad = torch.is_grad_enabled() and (Xs.requires_grad or Xt.requires_grad)
    pair, total, cost, shift, first = _PairLosses.apply(Xs, Xt, Us.detach(), float(p), shared, wu, wv, need_grad,
                                                        bool(return_slices))
    if return_first:                                # 0-dim loss of the first pair (the per-pair call shape)
        return first
    if return_total:
        return pair, total
    if return_slices:
        return pair, cost, shift
    return pair


def _check_cloud_dirs(Us):
    if not isinstance(Us, torch.Tensor) or not Us.is_cuda or Us.dtype != torch.float32:
        raise TypeError("Us must be a float32 device tensor")
    if Us.dim() not in (3, 4) or tuple(Us.shape[-2:]) != (3, 2):
        raise ValueError(f"Us must be (L,3,2) or (B,L,3,2), got {tuple(Us.shape)}")


def sliced_cost(Xs, Xt, Us, p=2, u_weights=None, v_weights=None):
    """Reference `sliced_cost`.  Per pair -- Xs (n,3), Xt (m,3), Us (L,3,2) -> 0-dim tensor, the mean
    over slices (:286).  Batched -- Xs (B,n,3), Xt (B,m,3), Us (B,L,3,2) -> shape-[1] tensor, the SUM
    over pairs of the per-pair means (_fast.py:291-293).  Batched p == 1, which raises in the
    reference, is evaluated pair-wise here (documented extension)."""
    if Xs.dim() == 2:
        return ssw_pair_losses(Xs.unsqueeze(0), Xt.unsqueeze(0), Us, p, u_weights=u_weights, v_weights=v_weights,
                               return_first=True)
    _, total = ssw_pair_losses(Xs, Xt, Us, p, u_weights=u_weights, v_weights=v_weights, return_total=True)
    return total


def sliced_wasserstein_sphere(Xs, Xt, num_projections, device, u_weights=None, v_weights=None, p=2):
    """Reference signature (:289).  Xs (n,3), Xt (m,3) on `device` -> 0-dim tensor."""
    U = draw_directions(num_projections, device, d=Xs.shape[1])
    return sliced_cost(Xs, Xt, U, p=p, u_weights=u_weights, v_weights=v_weights)


def sliced_wasserstein_sphere_fast(Xs, Xt, num_projections, device, u_weights=None, v_weights=None, p=2):
    """Reference batched signature (_fast.py:298).  Xs (B,n,3), Xt (B,m,3) -> shape-[1] tensor."""
    U = draw_directions(num_projections, device, batch=Xs.shape[0], d=Xs.shape[2])
    return sliced_cost(Xs, Xt, U, p=p, u_weights=u_weights, v_weights=v_weights)


# ------------------------------------------------------------------------------------------------------------------
# circle level: the reference's binary_search_circle / emd1D_circle on rows of circle coordinates
# ------------------------------------------------------------------------------------------------------------------
class _CircleOT(torch.autograd.Function):
    """(rows,n), (rows,m) circle coordinates -> (rows,) W_p^p.  One launch; the gradient rows come out of the same
    kernels as d cost / d coordinate (what the sliced path calls its coefficient rows)."""

    @staticmethod
    def forward(ctx, u, v, p, wu, wv, need_grad, method):
        lib = _lib.load()
        rows, n = u.shape
        m = v.shape[1]
        dev = u.device
        uc, vc = u.contiguous(), v.contiguous()
        cost = torch.empty(rows, dtype=torch.float32, device=dev)
        gu = torch.empty(rows, n, dtype=torch.float32, device=dev) if need_grad else None
        gv = torch.empty(rows, m, dtype=torch.float32, device=dev) if need_grad else None

        def wargs(w, cnt):
            if w is None:
                return None, 0
            return w.data_ptr(), (0 if w.dim() == 1 else cnt)
        wu_p, wu_s = wargs(wu, n)
        wv_p, wv_s = wargs(wv, m)
        with torch.cuda.device(dev):
            _lib.check(lib.shw_circle_ot(uc.data_ptr(), vc.data_ptr(), wu_p, wv_p, wu_s, wv_s, rows, n, m, float(p),
                                         int(method), cost.data_ptr(), None, gu.data_ptr() if need_grad else None,
                                         gv.data_ptr() if need_grad else None, _stream_ptr(dev)), "shw_circle_ot")
        if need_grad:
            ctx.save_for_backward(gu, gv)
        return cost

    @staticmethod
    def backward(ctx, g):
        gu, gv = ctx.saved_tensors
        w = g.to(torch.float32).unsqueeze(1)
        return gu * w, gv * w, None, None, None, None, None


def _check_rows(name, t):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32:
        raise TypeError(f"{name} must be a float32 device tensor (no CPU fallback)")
    if t.dim() not in (1, 2):
        raise ValueError(f"{name} must be (n,) or (rows, n) circle coordinates")
    return t.unsqueeze(0) if t.dim() == 1 else t


def _circle_ot(u_values, v_values, u_weights, v_weights, p, method):
    u, v = _check_rows("u_values", u_values), _check_rows("v_values", v_values)
    if u.shape[0] != v.shape[0]:
        raise ValueError("u_values and v_values need the same number of rows")
    if not (float(p) >= 1.0):
        raise ValueError("p must be >= 1")
    wu = _check_weights("u_weights", u_weights, u.shape[1], u.shape[0], u.device)
    wv = _check_weights("v_weights", v_weights, v.shape[1], u.shape[0], u.device)
    need_grad = torch.is_grad_enabled() and (u.requires_grad or v.requires_grad)
    return _CircleOT.apply(u, v, float(p), wu, wv, need_grad, method)


def binary_search_circle(u_values, v_values, u_weights=None, v_weights=None, p=1):
    """Reference `binary_search_circle` (max_spherical_sliced_w.py:117-207): circular OT cost W_p^p between rows of circle
    coordinates in [0, 1], (rows, n) and (rows, m) -> (rows,), by the bisection over the cut for EVERY p >= 1.  p = 1 is
    the reference's default and ends in `Cost`'s p == 1 branch (:107-108): the true circular W_1, up to 2.5 % away from
    what `emd1D_circle` returns (that formula leaves out the wrap segment; `sliced_cost` sends p == 1 there, :281-284).
    Differentiable w.r.t. the coordinates (the cut is detached as at :207)."""
    return _circle_ot(u_values, v_values, u_weights, v_weights, p, _lib.CIRCLE_BISECTION)


def emd1D_circle(u_values, v_values, u_weights=None, v_weights=None, p=1):
    """Reference `emd1D_circle` (:210-247): the level-median formula for p = 1, quirk included.
    The reference function has no branch for p != 1 (it falls off its end and returns None); that is an error here."""
    if float(p) != 1.0:
        raise ValueError("emd1D_circle is the p = 1 level-median formula (the reference returns None for p != 1); "
                         "use binary_search_circle for p != 1")
    return _circle_ot(u_values, v_values, u_weights, v_weights, 1.0, _lib.CIRCLE_LEVEL_MEDIAN)
