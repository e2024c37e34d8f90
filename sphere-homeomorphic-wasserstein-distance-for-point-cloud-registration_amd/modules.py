"""Module-level call shapes of the reference, host-side Python over the HIP op:

* `SlicedSphereW` -- drop-in for the `CSW` slot that train_W_COS.py:393 / train_Pseudo_W_COS.py:391 fill with
  `Cos_disimilarity_W(device, p)` (s2_wasserstein.py:13-66): ctor `(device, p)`, call `CSW(x, y) -> scalar`,
  x (B,N,3) or (N,3); batch-MEAN of the per-pair distance ** (1/p) (s2_wasserstein.py:41-48).
* `max_spherical_wassersten_distance` / `_fast` -- the adversarial phi-max wrappers
  (max_spherical_sliced_w.py:498-536, _fast.py:346-380): `forward(first, second, train_or_test) ->
  (ssw, phi(first), phi(second))`.
* `max_cos_disimilarity_wassersten_distance` -- the LIVE trainer criterion (s2_wasserstein.py:211-262, built at
  train_W_COS.py:404) with its injected `CSW` slot: phi-max inner loop with the |norm - 1| regulariser, then the
  distance on the non-detached clouds.  Host-side Python only; `SlicedSphereW` goes in the CSW slot.
* `SSWCriterion` / `ChamferCriterion` -- the trainer-level `criteria(template, transformed_source, ...)`
  slots (train_W_COS.py:133,171; train_CD.py:123,161).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .chamfer import chamfer_distance
from .ssw import draw_directions, sliced_wasserstein_sphere, sliced_wasserstein_sphere_fast, ssw_pair_losses


class SlicedSphereW(nn.Module):
    """`CSW`-slot module backed by the spherical sliced-Wasserstein HIP op."""

    def __init__(self, device, p=2, num_projections=512):
        super().__init__()
        self.device = device
        self.p = p
        self.num_projections = num_projections

    def forward(self, x, y):
        if x.dim() == 2:
            x, y = x.unsqueeze(0), y.unsqueeze(0)
        B = x.shape[0]
        U = draw_directions(self.num_projections, x.device, batch=B, d=x.shape[-1])
        pair = ssw_pair_losses(x, y, U, self.p)
        if B >= 2:
            return torch.pow(pair, 1.0 / self.p).sum() / int(B)      # s2_wasserstein.py:41-45
        return torch.pow(pair[0], 1.0 / self.p)                        # :47-48


def _sum_over_pairs(SSW, first, second, num_projections, device, p):
    """`ssw = sum_i SSW(first[i], second[i], ...)` (:518-519).  When SSW is this package's own per-pair
    function the B direction sets are drawn in the same order (same generator consumption) and the B pairs
    are evaluated by ONE batched launch instead of B."""
    if SSW is sliced_wasserstein_sphere:
        U = torch.stack([draw_directions(num_projections, device, d=first.shape[-1]) for _ in range(first.shape[0])])
        return ssw_pair_losses(first, second, U, p, return_total=True)[1].reshape(())
    ssw = 0
    for i in range(len(first)):
        ssw = ssw + SSW(first[i], second[i], num_projections, device, p=p)
    return ssw


class max_spherical_wassersten_distance(nn.Module):
    """phi-max wrapper, per-pair SSW (max_spherical_sliced_w.py:498-536)."""

    def __init__(self, num_projections, phi, SSW, phi_op, p=2, max_iter=10, device="cuda", verbose=False):
        super().__init__()
        self.num_projections, self.phi, self.SSW, self.phi_op = num_projections, phi, SSW, phi_op
        self.p, self.max_iter, self.device, self.verbose = p, max_iter, device, verbose
        self.on_inner_value = None       # optional observer of the per-iteration ssw (the reference prints it, :524)

    def _ssw(self, a, b):
        return _sum_over_pairs(self.SSW, a, b, self.num_projections, self.device, self.p)

    def forward(self, first_samples, second_samples, train_or_test="train"):
        first_detach, second_detach = first_samples.detach(), second_samples.detach()
        if train_or_test == "train":
            for _ in range(self.max_iter):
                ssw = self._ssw(self.phi(first_detach), self.phi(second_detach))
                loss = -ssw                                   # gradient ascent on phi (:521)
                self.phi_op.zero_grad()
                loss.backward(retain_graph=True)
                self.phi_op.step()
                if self.verbose:
                    print(ssw.item())
                if self.on_inner_value is not None:
                    self.on_inner_value(float(ssw.detach().sum()))
        elif train_or_test != "test":
            raise ValueError("train_or_test must be 'train' or 'test'")
        first_t, second_t = self.phi(first_samples), self.phi(second_samples)
        return self._ssw(first_t, second_t), first_t, second_t


class GraphedAscent:
    """hipGraph capture of ONE inner iteration of a phi-max loop -- phi forward on both (detached) clouds, the sliced
    loss forward + gradient kernels, phi backward, optimizer step -- replayed `max_iter` times per trainer step
    (the loop of _fast.py:359-372).  An inner iteration is ~40 small launches around three HIP
    kernels; replaying it removes the launch and Python overhead that dominates once the loss kernel is fast.

    Requirements of graph capture: static input buffers (the clouds are copied into them, two B*N*3 copies per
    trainer step) and an optimizer whose step is capturable (torch.optim.Adam(..., capturable=True)).  Directions
    drawn inside the loss (torch.randn on the device generator) stay random across replays: torch registers the
    generator with the graph.  The first call of a shape runs `warmup` eager iterations on a side stream (they
    count as iterations of the loop) and captures; every later iteration is a replay.  Replays execute the same
    kernels on the same buffers as the eager loop: results are bit-identical (tests/test_r2_gpu.py)."""

    def __init__(self, phi, phi_op, objective, warmup=2):
        self.phi, self.phi_op, self.objective, self.warmup = phi, phi_op, objective, warmup
        self._graphs = {}

    def _iteration(self, a, b):
        value = self.objective(self.phi(a), self.phi(b))
        self.phi_op.zero_grad(set_to_none=True)
        (-value).sum().backward()
        self.phi_op.step()
        return value.detach()

    def run(self, first, second, iters, observer=None):
        if iters <= 0:
            return
        if not first.is_cuda:
            raise RuntimeError("GraphedAscent needs device tensors")
        for group in self.phi_op.param_groups:
            if not group.get("capturable", False):
                raise RuntimeError("the phi optimizer must be created with capturable=True to be replayed in a hipGraph")
        key = (tuple(first.shape), tuple(second.shape), first.dtype, first.device)
        entry = self._graphs.get(key)
        done = 0
        if entry is None:
            a, b = first.detach().clone(), second.detach().clone()
            side = torch.cuda.Stream(first.device)
            side.wait_stream(torch.cuda.current_stream(first.device))
            with torch.cuda.stream(side):
                while done < min(self.warmup, iters):
                    v = self._iteration(a, b)
                    done += 1
                    if observer is not None:
                        observer(float(v.sum()))
            torch.cuda.current_stream(first.device).wait_stream(side)
            if done == iters:
                return                      # nothing left to replay this time: capture on the next call
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                value = self._iteration(a, b)
            entry = (graph, a, b, value)
            self._graphs[key] = entry
        graph, a, b, value = entry
        a.copy_(first.detach())
        b.copy_(second.detach())
        while done < iters:
            graph.replay()
            done += 1
            if observer is not None:
                observer(float(value.sum()))


class GraphedStep:
    """hipGraph capture of a WHOLE optimisation step -- zero_grad, loss, backward, optimizer step -- for callers whose step is
    launch-bound.  The notebooks' gradient flow (Wasserstein_flow_problem/Flow_cube.ipynb:1372-1395: one 1200-point pair,
    100 slices, `loss = sliced_wasserstein_sphere(evolving, target, 100, device, p=2); loss.backward(); optimizer.step()`) is
    ~15 launches of 2-20 us each: 58 us of GPU work inside 0.29 ms of launch and Python overhead per step (round 3,
    profiles/r03_notebook_*).  Replayed as one graph the step costs what the GPU work costs.

        step = shw.GraphedStep(lambda: shw.sliced_wasserstein_sphere(evolving, target, 100, device, p=2), optimizer)
        for i in range(400):
            loss = step()            # a static tensor that holds the loss of the step just replayed

    Requirements of graph capture: the tensors the closure reads and the optimizer's parameters keep their addresses
    (update them in place), and the optimizer's step is capturable (`torch.optim.Adam([...], capturable=True)`; add
    `fused=True` for one kernel instead of eight).  Directions drawn inside the loss stay random across replays: torch
    registers the device generator with the graph.  The first `warmup` calls run eagerly on a side stream (they are real
    steps), the next one captures, every later call is a replay of the same kernels on the same buffers -- bit-identical
    to the eager loop (tests/test_r3_gpu.py)."""

    def __init__(self, loss_fn, optimizer, warmup=3):
        self.loss_fn, self.optimizer, self.warmup = loss_fn, optimizer, warmup
        self._calls = 0
        self._graph = None
        self._loss = None
        self._ones = None
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise RuntimeError("the optimizer must be created with capturable=True to be replayed in a hipGraph")

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn()
        # the root gradient is a static tensor of ones (allocated on the first, eager call) and a 0-dim loss is not summed:
        # `loss.sum().backward()` costs a reduction kernel and a fill kernel per step, two launches of the notebooks' eleven
        if self._ones is None or self._ones.shape != loss.shape:
            self._ones = torch.ones_like(loss)
        torch.autograd.backward(loss, grad_tensors=self._ones)
        self.optimizer.step()
        return loss.detach()

    def __call__(self):
        if self._graph is not None:
            self._graph.replay()
            return self._loss
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        device = params[0].device
        if self._calls < self.warmup:
            self._calls += 1
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                loss = self._step()
            torch.cuda.current_stream(device).wait_stream(side)
            return loss
        self._graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self._graph):
            self._loss = self._step()
        self._graph.replay()                # capture records, it does not execute: this is the step of this call
        return self._loss


class max_spherical_wassersten_distance_fast(max_spherical_wassersten_distance):
    """phi-max wrapper, batched SSW (_fast.py:346-380).  `graph=True`: the inner ascent iterations are captured once
    per input shape and replayed (GraphedAscent); the returned value and maps are computed eagerly as in the
    reference, so the trainer's backward through them is unchanged."""

    def __init__(self, num_projections, phi, SSW, phi_op, p=2, max_iter=10, device="cuda", verbose=False, graph=False):
        super().__init__(num_projections, phi, SSW, phi_op, p=p, max_iter=max_iter, device=device, verbose=verbose)
        self._graphed = GraphedAscent(phi, phi_op, self._ssw) if graph else None

    def _ssw(self, a, b):
        return self.SSW(a, b, self.num_projections, self.device, p=self.p)

    def forward(self, first_samples, second_samples, train_or_test="train"):
        if self._graphed is None or train_or_test != "train":
            return super().forward(first_samples, second_samples, train_or_test)
        observer = None
        if self.verbose or self.on_inner_value is not None:
            def observer(v):
                if self.verbose:
                    print(v)
                if self.on_inner_value is not None:
                    self.on_inner_value(v)
        self._graphed.run(first_samples, second_samples, self.max_iter, observer)
        first_t, second_t = self.phi(first_samples), self.phi(second_samples)
        return self._ssw(first_t, second_t), first_t, second_t


class max_cos_disimilarity_wassersten_distance(nn.Module):
    """The criterion train_W_COS.py:404 builds (s2_wasserstein.py:211-262): `criteria(template, transformed_source,
    train_or_test=...) -> (cswd, phi(template), phi(source))`.  Same constructor arguments; `CSW` is any
    `CSW(x, y) -> scalar` module -- the reference passes its exact-EMD `Cos_disimilarity_W` (POT, out of scope),
    the drop-in passes `SlicedSphereW`.  The s2_wasserstein module itself cannot be imported without POT, so this
    mirror is checked against a restatement (oracle/phi_max_mirror.py): parity unpinned by fixtures."""

    def __init__(self, phi, CSW, device, phi_op, max_iter=10, lam=0.1, psi_minibatch_size=5, graph=False):
        super().__init__()
        self.phi, self.CSW, self.phi_op = phi, CSW, phi_op
        self.max_iter, self.device, self.reg_lam = max_iter, device, lam
        # graph=True: the inner iterations are replayed from a hipGraph (GraphedAscent); needs a capturable optimizer
        self._graphed = GraphedAscent(phi, phi_op, self._ascent_objective) if graph else None

    @staticmethod
    def regularization_of_normalizing_flow(x):
        if x.dim() == 2:
            x = x.unsqueeze(0)
        return torch.sum(torch.abs(torch.linalg.vector_norm(x, dim=-1) - 1))           # :222-230

    def _ascent_objective(self, a, b):
        reg = self.reg_lam * (self.regularization_of_normalizing_flow(a) / (a.shape[0] * a.shape[1])
                              + self.regularization_of_normalizing_flow(b) / (b.shape[0] * b.shape[1]))
        return self.CSW(a, b) - reg                                                  # maximised (:249: loss = reg - cswd)

    def forward(self, first_samples, second_samples, train_or_test="train"):
        first_detach, second_detach = first_samples.detach(), second_samples.detach()
        if train_or_test == "train":
            self.phi.train()
            if self._graphed is not None:
                self._graphed.run(first_detach, second_detach, self.max_iter)
            else:
                for _ in range(self.max_iter):
                    self.phi_op.zero_grad()
                    a, b = self.phi(first_detach), self.phi(second_detach)
                    loss = -self._ascent_objective(a, b)                              # gradient ascent (:249)
                    loss.backward(retain_graph=True)
                    self.phi_op.step()
        elif train_or_test == "test":
            self.phi.eval()
        else:
            raise ValueError("train_or_test must be 'train' or 'test'")
        a, b = self.phi(first_samples), self.phi(second_samples)
        return self.CSW(a, b), a, b


class SSWCriterion(nn.Module):
    """Trainer-level slot of train_W_COS.py:133,171: `criteria(template, transformed_source,
    train_or_test=...) -> (loss, phi(template), phi(source))` with the sliced loss in the CSW position and an
    optional sphere map phi (identity when None)."""

    def __init__(self, device, p=2, num_projections=512, phi=None):
        super().__init__()
        self.csw = SlicedSphereW(device, p, num_projections)
        self.phi = phi

    def forward(self, template, source, train_or_test="train"):
        a = template if self.phi is None else self.phi(template)
        b = source if self.phi is None else self.phi(source)
        return self.csw(a, b), a, b


class ChamferCriterion(nn.Module):
    """train_CD.py:123,161: `criteria(template, transformed_source)[0]`."""

    def forward(self, template, source, batch_reduction="mean"):
        return chamfer_distance(template, source, batch_reduction=batch_reduction)


__all__ = ["GraphedAscent", "SlicedSphereW", "max_spherical_wassersten_distance", "max_spherical_wassersten_distance_fast",
           "max_cos_disimilarity_wassersten_distance",
           "SSWCriterion", "ChamferCriterion", "sliced_wasserstein_sphere", "sliced_wasserstein_sphere_fast"]
