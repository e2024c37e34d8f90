#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X spherical sliced-Wasserstein loss.

Metric (BASELINE.json): point-pairs/s = B*N*L / wall time of one loss evaluation, at N=2048, L=512,
p=2, fp32 (BASELINE config 3: batch=64).  A "step" is one full loss evaluation of the batch
(projection + per-slice sorts + circular OT solve + reduction to the scalar) with the clouds and
directions already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU (weak scaling): the (pair x slice) axis is sharded by pairs -- every rank owns 64 pairs
(global batch 64*N) -- and the only collective is ONE RCCL all-reduce of the scalar loss.

Rank 0 prints ONE JSON line.  At N=1 it also carries
  "roofline":     algorithmic bytes of the dominant kernel / its measured duration vs the 8 TB/s HBM peak
  "cpu_baseline": the CPU oracle (oracle/ref_mirror.py, kind "port") timed on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def next_pow2(v):
    r = 1
    while r < v:
        r <<= 1
    return max(r, 64)


def stages(n):
    k = next_pow2(n).bit_length() - 1
    return k * (k + 1) // 2


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when its communicator is created; the contract is ONE JSON line on
    stdout, so file descriptor 1 points at stderr while the process group and its first collectives come up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def make_inputs(B, N, L, rank, device):
    """host-generated, seeded, unit-normalised Gaussian clouds, explicit directions."""
    def cloud(seed):
        g = torch.Generator().manual_seed(seed)
        return torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1)
    x = cloud(1234 + 10 * rank)
    y = cloud(1235 + 10 * rank)
    g = torch.Generator().manual_seed(4321 + 10 * rank)
    U = torch.linalg.qr(torch.randn(B, L, 3, 2, generator=g))[0]
    return x.to(device), y.to(device), U.contiguous().to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--slices", type=int, default=512)
    ap.add_argument("--p", type=float, default=2.0)
    ap.add_argument("--mode", default="forward", choices=["forward", "train", "chamfer"],
                    help="forward: loss evaluation (the headline metric); train: loss + input gradients through "
                         "the Python mirror's autograd Function; chamfer: Chamfer baseline forward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as a hipGraph instead of launching eagerly (measured on MI355X: eager "
                         "0.352 ms/step vs replay 0.358 -- the step is one ~0.35 ms kernel plus one small one, so "
                         "replay's fixed cost outweighs the launch it saves; kept as an option)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=8)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # under torchrun: always exercise the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            dist.init_process_group("nccl", device_id=device)   # "nccl" is RCCL on ROCm
            warm = torch.zeros(1, device=device)
            dist.all_reduce(warm)                                # creates the communicator (and its banner) now
            dist.barrier()
            torch.cuda.synchronize(device)

    import shw_amd
    lib = shw_amd._lib.load()
    from shw_amd import _lib

    B, N, L, p = args.batch, args.points, args.slices, args.p
    x, y, U = make_inputs(B, N, L, rank, device)
    stream = torch.cuda.current_stream(device).cuda_stream

    slice_cost = torch.empty(B * L, dtype=torch.float32, device=device)
    slice_shift = torch.empty(B * L, dtype=torch.int32, device=device)
    pair_loss = torch.empty(B, dtype=torch.float32, device=device)
    totals = [torch.empty(2, dtype=torch.float32, device=device) for _ in range(2)]   # ring: see step()
    total = totals[0]

    if args.mode != "forward":
        return side_modes(args, shw_amd, x, y, U, device, rank)

    def enqueue_loss(out=None):
        """The hot path: every kernel of one loss evaluation, enqueued on torch's current HIP stream."""
        out = total if out is None else out
        st = torch.cuda.current_stream(device).cuda_stream
        _lib.check(lib.shw_ssw_forward(x.data_ptr(), y.data_ptr(), U.data_ptr(), B, N, N, L, L * 6, p,
                                       slice_cost.data_ptr(), slice_shift.data_ptr(), st), "shw_ssw_forward")
        _lib.check(lib.shw_ssw_reduce(slice_cost.data_ptr(), B, L, 1.0 / L, pair_loss.data_ptr(),
                                      out.data_ptr(), st), "shw_ssw_reduce")

    graph = None
    if args.graph:
        # launch-bound tail (two small kernels behind a ~0.3 ms one): replay the step as a hipGraph
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            enqueue_loss()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            enqueue_loss()

    pending = [None, None]
    counter = [0]

    def step():
        """One loss evaluation.  Multi-GPU: the scalar all-reduce (the path's only collective, RCCL over xGMI)
        is issued asynchronously on RCCL's stream and awaited one step later, so its ~40 us of launch latency
        overlaps the next evaluation's kernels; results land in a two-deep ring of scalars."""
        if dist is None:
            if graph is not None:
                graph.replay()
            else:
                enqueue_loss()
            return total[0:1]
        slot = counter[0] & 1
        counter[0] += 1
        if pending[slot] is not None:
            pending[slot].wait()
        enqueue_loss(totals[slot])
        loss = totals[slot][0:1]
        pending[slot] = dist.all_reduce(loss, async_op=True)
        return loss

    def drain():
        for w in pending:
            if w is not None:
                w.wait()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    print("[bench] inputs ready, warming up", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enq = time.perf_counter() - t0
    if dist is not None:
        drain()
    torch.cuda.synchronize(device)
    t_sync = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    print("[bench] enqueue %.2f ms, device done %.2f ms, after barrier %.2f ms" % (1e3 * t_enq, 1e3 * t_sync, 1e3 * elapsed),
          file=sys.stderr, flush=True)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_value = float(loss.item())

    ms_per_step = 1e3 * elapsed / args.steps
    print("[bench] timed region done: %.3f ms/step" % ms_per_step, file=sys.stderr, flush=True)
    units_per_step = world * B * N * L
    out = {
        "metric": "point-pairs/sec (B*N*L projected+sorted+solved) at N=%d L=%d" % (N, L),
        "value": units_per_step / (elapsed / args.steps),
        "unit": "point-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (unit-normalised Gaussian clouds, seeds 1234/1235; QR directions, seed 4321)",
        "config": {"workload": "BASELINE config 3: sliced-W loss forward, batch=%d pairs/GPU, N=M=%d, L=%d, p=%g, "
                               "independent clouds" % (B, N, L, p),
                   "global_batch": world * B, "points": N, "slices": L, "p": p,
                   "sharding": "pairs across ranks, one all-reduce of the scalar loss" if world > 1 else "single GPU",
                   "launch": "hipGraph replay" if args.graph else "eager"},
        "loss": loss_value,
    }

    if True:
        # ---- roofline of the dominant kernel (ssw_forward_kernel), HIP events on the launch stream;
        #      per GPU (every rank measures its own launches, rank 0's figure is reported)
        reps = max(10, min(args.steps, 200))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            _lib.check(lib.shw_ssw_forward(x.data_ptr(), y.data_ptr(), U.data_ptr(), B, N, N, L, L * 6, p,
                                           slice_cost.data_ptr(), slice_shift.data_ptr(), stream), "shw_ssw_forward")
        ev1.record()
        torch.cuda.synchronize(device)
        kernel_ms = ev0.elapsed_time(ev1) / reps
        # clouds read once + directions + per-pair loss
        algo_bytes = 12 * B * (N + N) + 24 * B * L + 4 * B
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r01_traffic.json: FETCH_SIZE
        # doubled per the gfx950 correction + WRITE_SIZE); only valid for the workload it was measured on
        traffic = os.environ.get("SHW_BENCH_TRAFFIC_BYTES")
        if traffic is None and (B, N, L, p) == (64, 2048, 512, 2.0):
            try:
                with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
                    traffic = json.load(fh)["ssw_forward_kernel<32,4,2,true>"]["traffic_bytes"]
            except Exception:
                traffic = None
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": float(traffic) if traffic else None,
            "kernel": "ssw_forward_kernel<32,4,2,true>" if (N, p) == (2048, 2.0) else "ssw_forward_kernel",
            "kernel_ms": kernel_ms, "algorithmic_bytes": algo_bytes,
            "secondary_model": {"unit": "compare-exchanges/s", "achieved": B * L * 2 * (next_pow2(N) // 2) * stages(N) / (kernel_ms * 1e-3),
                                "note": "two bitonic sorts of next_pow2(N) keys per slice; the VALU/LDS-crossbar price "
                                        "list in DESIGN.md section 4 puts the bound at ~0.245 ms per launch at config 3"},
            "note": "compulsory HBM traffic is 0.06 B/point-pair: the kernel is VALU/LDS-crossbar bound (in-register "
                    "bitonic sort), not HBM bound; see DESIGN.md for the compare-exchange model",
            "point_pairs_per_s_kernel_only": B * N * L / (kernel_ms * 1e-3),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x, y, U, p, args.cpu_sample_pairs)
    if dist is not None:
        dist.barrier()

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def side_modes(args, shw, x, y, U, device, rank):
    """Secondary measurements (not the headline line): training step and Chamfer baseline, single GPU."""
    B, N, L, p = args.batch, args.points, args.slices, args.p
    if args.mode == "train":
        xs = x.clone().requires_grad_(True)
        ys = y.clone().requires_grad_(True)

        def step():
            xs.grad = None
            ys.grad = None
            shw.sliced_cost(xs, ys, U, p=p).backward()
        unit, per_step = "point-pairs/s (forward + input gradients)", B * N * L
    else:
        def step():
            shw.chamfer_distance(x, y)
        unit, per_step = "pair-distances/s (both directions)", 2 * B * N * N
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    el = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": args.mode, "value": per_step / el, "unit": unit, "ms_per_step": 1e3 * el,
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "dtype": "f32",
                      "config": {"workload": "B=%d N=%d L=%d p=%g" % (B, N, L, p)}}), flush=True)


def cpu_baseline(x, y, U, p, sample_pairs):
    """The CPU oracle (torch-CPU restatement of the reference's algorithm, oracle/ref_mirror.py) on the
    first `sample_pairs` pairs of the same workload; used here ONLY as the timed baseline."""
    from oracle import ref_mirror
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, int(os.environ.get("SHW_BENCH_CPU_THREADS", "16"))))   # GPU box: 16-core share per GPU
    torch.set_num_threads(cores)
    print("[bench] cpu baseline on %d threads ..." % cores, file=sys.stderr, flush=True)
    xs, ys, Us = x[:sample_pairs].cpu(), y[:sample_pairs].cpu(), U[:sample_pairs].cpu()
    N, L = xs.shape[1], Us.shape[1]
    ref_mirror.sliced_cost_batched(xs[:1], ys[:1], Us[:1], p=p)          # warm-up
    best = float("inf")
    val = None
    for _ in range(2):
        t0 = time.perf_counter()
        val = ref_mirror.sliced_cost_batched(xs, ys, Us, p=p)
        best = min(best, time.perf_counter() - t0)
    return {"value": sample_pairs * N * L / best, "unit": "point-pairs/s", "cores": cores, "kind": "port",
            "sample": "first %d pairs of the same batch (N=%d, L=%d, p=%g), best of 2 after 1 warm-up, "
                      "torch %d threads; %.2f s per run" % (sample_pairs, N, L, p, cores, best),
            "loss_of_sample": float(val.item())}


if __name__ == "__main__":
    main()
