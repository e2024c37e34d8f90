#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X spherical sliced-Wasserstein loss.

Metric (BASELINE.json): point-pairs/s = B*N*L / wall time of one loss evaluation, p=2, fp32.  A "step" is one
full loss evaluation of the batch (projection + per-slice sorts + circular OT solve + reduction to per-pair losses
and the scalar) with the clouds and directions already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts its N ranks itself, as child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

  N = 1 : BASELINE config 3 -- batch=64, N=M=2048, L=512 on one MI355X (the configuration the metric is quoted on).
  N > 1 : BASELINE config 4 -- GLOBAL batch=512, N=M=2048, L=1024, the (pair x slice) work sharded over the ranks:
          --shard slices (default, what config 4 names): rank r evaluates L/N directions of every pair;
          --shard pairs : rank r evaluates all 1024 directions of 512/N pairs.
          Either way each rank reduces its slices to per-pair partial sums (scaled by 1/L_global) and the ONLY
          collective is one RCCL all-reduce (sum) of 514 floats [512 per-pair losses | total, mean] per step.
          Total work is fixed as N grows ("scaling": "strong").

Rank 0 prints ONE JSON line.  At N=1 it also carries
  "roofline":     algorithmic bytes of the dominant kernel / its measured duration vs the 8 TB/s HBM peak
  "cpu_baseline": the CPU oracle (oracle/ref_mirror.py, kind "port") timed on a bounded sample (BASELINE.md section 3:
                  3 warm-ups, median of 5; forward and forward+backward; 1 thread and the box's CPU share), with the CPU
                  model and the host's physical core count
  "parity_rel_err": max relative difference between the GPU's per-pair losses and the CPU oracle's on the sample
                  pairs; the process exits non-zero when it exceeds 1e-5 (a wrong-but-fast kernel prints no headline)
Other modes (--mode train | chamfer | config5 | mirror) print their own single line; they are secondary figures.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
PARITY_TOL = 1e-5               # north_star: loss values within 1e-5 relative of the reference


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


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def metric_name(cfg_name, N, L):
    """BASELINE.json's metric string, verbatim, for the two configurations it is quoted on (config 3 on one GPU, config 4
    = its 8-GPU scaling leg); anything else names its own sizes."""
    if cfg_name != "custom":
        try:
            with open(os.path.join(ROOT, "BASELINE.json")) as fh:
                return json.load(fh)["metric"]
        except Exception:
            return "point-pairs/sec (B\u00b7N\u00b7L projected+sorted) at N=2048 L=512; 1/2/4/8-GPU scaling"
    return "point-pairs/sec (B*N*L projected+sorted) at N=%d L=%d" % (N, L)


def make_clouds(B, N, device, pair_lo=0, pair_hi=None):
    """host-generated, seeded, unit-normalised Gaussian clouds (every rank generates the same global
    batch and keeps the pairs it needs, so any GPU count sees identical data)."""
    pair_hi = B if pair_hi is None else pair_hi

    def cloud(seed):
        g = torch.Generator().manual_seed(seed)
        return torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1)[pair_lo:pair_hi].contiguous()
    return cloud(1234).to(device), cloud(1235).to(device)


def make_directions(shw, B, L, device, pair_lo=0, pair_hi=None, slice_lo=0, slice_hi=None):
    """Gaussian (B,L,3,2) matrices from a seeded CPU generator; the frames of the needed block come out of the HIP
    Householder kernel (shw_stiefel_frames: LAPACK's sign convention)."""
    pair_hi = B if pair_hi is None else pair_hi
    slice_hi = L if slice_hi is None else slice_hi
    g = torch.Generator().manual_seed(4321)
    Z = torch.randn(B, L, 3, 2, generator=g)[pair_lo:pair_hi, slice_lo:slice_hi].contiguous()
    return shw.stiefel_frames(Z.to(device))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=None, help="GLOBAL number of pairs (default 64 at N=1, 512 at N>1)")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--slices", type=int, default=None, help="GLOBAL number of slices (default 512 at N=1, 1024 at N>1)")
    ap.add_argument("--p", type=float, default=2.0)
    ap.add_argument("--shard", default="slices", choices=["slices", "pairs"],
                    help="N>1: which axis of the (pair x slice) work is split over the ranks")
    ap.add_argument("--mode", default="forward", choices=["forward", "train", "chamfer", "config5", "mirror"],
                    help="forward: loss evaluation through the C ABI (the headline metric); mirror: the same through "
                         "the drop-in Python call shw.sliced_cost; train: loss + input gradients through the mirror's "
                         "autograd Function; chamfer: Chamfer baseline forward; config5: PCRNet-shaped training step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as a hipGraph instead of launching eagerly (measured: no gain, the step is "
                         "one ~0.3 ms kernel plus one small one; kept as an option)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=4)
    args = ap.parse_args()

    # `python bench.py --gpus N` outside torchrun: start the N ranks ourselves.  Nothing has touched the GPU yet (torch
    # is imported, no HIP call made), and the ranks are fresh CHILD processes -- never an exec of this one.
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("SHW_BENCH_FORCE_SPAWN") == "1"):
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one process per GPU (python bench.py --gpus N starts them "
                         "itself; under torch.distributed.run pass --nproc-per-node N --gpus N)" % (args.gpus, world))
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
    from shw_amd.dist import shard_bounds

    multi = world > 1
    B = args.batch if args.batch is not None else (512 if multi else 64)
    L = args.slices if args.slices is not None else (1024 if multi else 512)
    N, p = args.points, args.p

    if args.mode != "forward":
        return side_modes(args, shw_amd, B, N, L, device)

    # ---- this rank's block of the (pair x slice) work
    pair_lo, pair_hi, slice_lo, slice_hi = 0, B, 0, L
    if multi:
        if args.shard == "slices":
            slice_lo, slice_hi = shard_bounds(L, world, rank)
        else:
            pair_lo, pair_hi = shard_bounds(B, world, rank)
    Bl, Ll = pair_hi - pair_lo, slice_hi - slice_lo
    x, y = make_clouds(B, N, device, pair_lo, pair_hi)
    U = make_directions(shw_amd, B, L, device, pair_lo, pair_hi, slice_lo, slice_hi)
    stream = torch.cuda.current_stream(device).cuda_stream

    slice_cost = torch.empty(max(Bl * Ll, 1), dtype=torch.float32, device=device)
    slice_shift = torch.empty(max(Bl * Ll, 1), dtype=torch.int32, device=device)
    # [per-pair losses of the GLOBAL batch (B) | total, mean (2)]: the block the all-reduce sums; two-deep ring
    outs = [torch.zeros(B + 2, dtype=torch.float32, device=device) for _ in range(2)]

    def enqueue_loss(out):
        """The hot path: every kernel of one loss evaluation of this rank's block, on torch's current HIP stream.
        Per-pair partials are scaled by 1/L_global, so the sum over ranks is the loss itself."""
        st = torch.cuda.current_stream(device).cuda_stream
        if Bl * Ll == 0:
            return
        _lib.check(lib.shw_ssw_forward(x.data_ptr(), y.data_ptr(), U.data_ptr(), Bl, N, N, Ll, Ll * 6, p,
                                       slice_cost.data_ptr(), slice_shift.data_ptr(), st), "shw_ssw_forward")
        _lib.check(lib.shw_ssw_reduce(slice_cost.data_ptr(), Bl, Ll, 1.0 / L, out.data_ptr() + 4 * pair_lo,
                                      out.data_ptr() + 4 * B, st), "shw_ssw_reduce")

    # ---- roofline of the dominant kernel (ssw_forward_kernel), HIP events on the launch stream, measured BEFORE
    #      the timed region: it also brings the GPU to its steady-state clocks, which 5 warm-up steps of 0.3 ms do not
    reps = max(20, min(args.steps, 200))
    for _ in range(100):                     # ~25 ms of launches: the clocks are up before anything is timed
        enqueue_loss(outs[0])
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        _lib.check(lib.shw_ssw_forward(x.data_ptr(), y.data_ptr(), U.data_ptr(), Bl, N, N, Ll, Ll * 6, p,
                                       slice_cost.data_ptr(), slice_shift.data_ptr(), stream), "shw_ssw_forward")
    ev1.record()
    torch.cuda.synchronize(device)
    kernel_ms = ev0.elapsed_time(ev1) / reps

    graph = None
    if args.graph and not multi:
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            enqueue_loss(outs[0])
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            enqueue_loss(outs[0])

    pending = [None, None]
    counter = [0]

    def step():
        """One loss evaluation.  Multi-GPU: the all-reduce of the 514-float block (the path's only collective, RCCL
        over xGMI) is issued asynchronously on RCCL's stream and awaited one step later, so its launch latency
        overlaps the next evaluation's kernels; results land in a two-deep ring."""
        if dist is None:
            if graph is not None:
                graph.replay()
            else:
                enqueue_loss(outs[0])
            return outs[0]
        slot = counter[0] & 1
        counter[0] += 1
        if pending[slot] is not None:
            pending[slot].wait()
        out = outs[slot]
        if multi and args.shard == "pairs":
            out.zero_()                      # foreign pairs must contribute 0 to the sum
        enqueue_loss(out)
        pending[slot] = dist.all_reduce(out, async_op=True)
        return out

    def drain():
        for w in pending:
            if w is not None:
                w.wait()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    log("inputs ready (rank %d: pairs [%d,%d) slices [%d,%d)), warming up" % (rank, pair_lo, pair_hi, slice_lo, slice_hi))
    for _ in range(args.warmup):
        step()
    if dist is not None:
        drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    t_enq = time.perf_counter() - t0
    if dist is not None:
        drain()
    torch.cuda.synchronize(device)
    t_sync = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    log("enqueue %.2f ms, device done %.2f ms, after barrier %.2f ms" % (1e3 * t_enq, 1e3 * t_sync, 1e3 * elapsed))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    pair_losses = out[:B].clone()
    loss_value = float(out[B].item())                     # N > 1: sum over ranks of the partial totals

    ms_per_step = 1e3 * elapsed / args.steps
    log("timed region done: %.3f ms/step" % ms_per_step)
    units_per_step = B * N * L
    cfg_name = "BASELINE config 4" if (multi and (B, N, L) == (512, 2048, 1024)) else \
        ("BASELINE config 3" if (B, N, L) == (64, 2048, 512) else "custom")
    sharding = "single GPU"
    if multi:
        sharding = ("slices: each rank takes %d of the %d directions of every pair" % (Ll, L) if args.shard == "slices"
                    else "pairs: each rank takes %d of the %d pairs" % (Bl, B)) + \
            "; one RCCL all-reduce (sum) of %d floats per step" % (B + 2)
    result = {
        "metric": metric_name(cfg_name, N, L),
        "value": units_per_step / (elapsed / args.steps),
        "unit": "point-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if multi else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (unit-normalised Gaussian clouds, seeds 1234/1235; Householder-QR directions of seeded "
                "Gaussians, seed 4321)",
        "config": {"workload": "%s: sliced-W loss forward, global batch=%d pairs, N=M=%d, L=%d, p=%g, independent "
                               "clouds" % (cfg_name, B, N, L, p),
                   "global_batch": B, "points": N, "slices": L, "p": p, "sharding": sharding,
                   "launch": "hipGraph replay" if graph is not None else "eager"},
        "loss": loss_value,
    }

    # clouds read once + directions + per-pair loss -- for this rank's block
    algo_bytes = 12 * Bl * (N + N) + 24 * Bl * Ll + 4 * Bl
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE/WRITE_SIZE corrected per the
    # guide); only valid for the workload it was measured on
    traffic = os.environ.get("SHW_BENCH_TRAFFIC_BYTES")
    if traffic is None and (Bl, N, Ll, p) == (64, 2048, 512, 2.0):
        for name in ("r03_traffic.json", "r02_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    traffic = json.load(fh)["ssw_forward_kernel<32,1,2,true>"]["traffic_bytes"]
                break
            except Exception:
                traffic = None
    result["roofline"] = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS, "traffic": float(traffic) if traffic else None,
        "kernel": "ssw_forward_kernel<32,1,2,true>" if (N, p) == (2048, 2.0) else "ssw_forward_kernel",
        "kernel_ms": kernel_ms, "algorithmic_bytes": algo_bytes,
        "secondary_model": {"unit": "compare-exchanges/s",
                            "achieved": Bl * Ll * 2 * (next_pow2(N) // 2) * stages(N) / (kernel_ms * 1e-3),
                            "note": "compare-exchanges of two bitonic sorts of next_pow2(N) keys per slice -- the unit "
                                    "of round 1's network kernel, kept for comparison; the distribution sort of round 2 "
                                    "does the same job with ~1/4 of them (DESIGN.md section 4)"},
        "note": "compulsory HBM traffic is 0.06 B/point-pair: the kernel is VALU-issue / LDS bound (projection, "
                "distribution sort through LDS, shift solve), not HBM bound; DESIGN.md section 4 has the VALU model",
        "point_pairs_per_s_kernel_only": Bl * N * Ll / (kernel_ms * 1e-3),
    }

    rc = 0
    if rank == 0:
        if not multi and not args.no_cpu_baseline:
            base = cpu_baseline(x, y, U, p, args.cpu_sample_pairs)
            ref_pairs = base.pop("_pair_losses")
            got = pair_losses[:len(ref_pairs)].double().cpu()
            err = float(((got - ref_pairs).abs() / ref_pairs.abs()).max())
            result["cpu_baseline"] = base
            result["parity_rel_err"] = err
            result["parity_checked"] = "per-pair losses of the first %d pairs vs oracle/ref_mirror (CPU)" % len(ref_pairs)
        else:
            # no full CPU baseline (N > 1, or switched off): still guard the headline with a small oracle sample --
            # the first 8 of this rank's slices of its first pair
            err, what = mini_parity(x, y, U, slice_cost, Ll, p)
            result["parity_rel_err"] = err
            result["parity_checked"] = what
        if os.environ.get("SHW_BENCH_SKIP_PARITY") == "1":       # developer ablation builds (tools/ab.sh) only
            result["parity_checked"] = "SKIPPED (SHW_BENCH_SKIP_PARITY=1): not a valid headline"
        elif not (result["parity_rel_err"] <= (PARITY_TOL if "cpu_baseline" in result else 2e-5)):
            log("PARITY FAILURE: relative error %.3e against the CPU oracle" % result["parity_rel_err"])
            rc = 1
        # the same evaluation through the drop-in Python call (what a reference caller gets), for the record
        if not multi:
            result["mirror_ms_per_step"] = time_mirror(shw_amd, x, y, U, p, device)
    if dist is not None:
        dist.barrier()

    if rank == 0:
        if rc == 0:
            print(json.dumps(result), flush=True)
        else:
            print(json.dumps({"error": "parity guard failed", "parity_rel_err": result["parity_rel_err"]}), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return rc


def self_launch(gpus):
    """Run this script as `gpus` ranks under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a
    free port), pass rank 0's JSON line through on stdout, everything else on stderr, and return the launcher's status."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.pop("SHW_BENCH_FORCE_SPAWN", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: what this pool's driver supports
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting %d ranks: %s" % (gpus, " ".join(cmd)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.lstrip().startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def time_mirror(shw, x, y, U, p, device, reps=50):
    """ms per `shw.sliced_cost(x, y, U, p)` -- the reference's call shape (_fast.py:258), loss-only mode."""
    for _ in range(10):
        shw.sliced_cost(x, y, U, p=p)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        v = shw.sliced_cost(x, y, U, p=p)
    torch.cuda.synchronize(device)
    del v
    return 1e3 * (time.perf_counter() - t0) / reps


def mini_parity(x, y, U, slice_cost, Ll, p):
    from oracle import ref_mirror
    k = min(8, Ll)
    ref = ref_mirror.per_slice_costs(x[0].cpu(), y[0].cpu(), U[0, :k].cpu(), p=p).double()
    got = slice_cost[:k].double().cpu()
    err = float(((got - ref).abs() / ref.abs().clamp_min(1e-30)).max())
    return err, "per-slice costs of the first %d slices of this rank's first pair vs oracle/ref_mirror (CPU), bound 2e-5" % k


def side_modes(args, shw, B, N, L, device):
    """Secondary measurements (not the headline line), single GPU."""
    p = args.p
    if args.mode == "config5":
        return config5_mode(args, shw, device)
    x, y = make_clouds(B, N, device)
    U = make_directions(shw, B, L, device)
    if args.mode == "train":
        xs = x.clone().requires_grad_(True)
        ys = y.clone().requires_grad_(True)

        def step():
            xs.grad = None
            ys.grad = None
            shw.sliced_cost(xs, ys, U, p=p).backward()
        unit, per_step = "point-pairs/s (forward + input gradients)", B * N * L
    elif args.mode == "mirror":
        def step():
            shw.sliced_cost(x, y, U, p=p)
        unit, per_step = "point-pairs/s (forward through the drop-in Python call)", B * N * L
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
    return 0


def config5_mode(args, shw, device):
    """BASELINE config 5: PCRNet-shaped registration network, 8 refinement iterations, B=32, N=2048, phi-max
    criterion with the sliced loss (L=512) in the CSW slot; forward + backward + Adam (examples/config5_train_step.py).
    Prints the step time and how much of it the sliced-loss kernels take."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("config5_train_step", os.path.join(ROOT, "examples", "config5_train_step.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    B = args.batch or 32
    L = args.slices or 512
    steps = min(args.steps, 50)
    losses, times = mod.run(batch=B, points=args.points, slices=L, steps=max(steps, 6), verbose=False, criterion="csw",
                            phi_max_iter=1, iteration_num=8)
    med = statistics.median(times[3:])
    ssw = mod.ssw_share(B, args.points, L, evaluations=2)
    print(json.dumps({"metric": "config5 training step", "value": 1.0 / med, "unit": "steps/s",
                      "ms_per_step": 1e3 * med, "ssw_ms_per_step": 1e3 * ssw, "ssw_share": ssw / med,
                      "torch_ms_per_step": 1e3 * (med - ssw), "n_gpus": 1, "steps": len(times), "dtype": "f32",
                      "loss_first": losses[0], "loss_last": losses[-1],
                      "config": {"workload": "BASELINE config 5: PCRNet-shaped regressor (emb 1024, 5 FC, 8 iterations) + "
                                             "phi-max criterion (planar flow, 1 inner step) with SlicedSphereW in the CSW "
                                             "slot; B=%d, N=%d, L=%d, forward+backward+Adam" % (B, args.points, L)}}),
          flush=True)
    return 0


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def physical_cores():
    """(physical cores, sockets) of the host from /proc/cpuinfo: distinct (physical id, core id) pairs."""
    cores, sockets = set(), set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                        sockets.add(phys)
                    phys = core = None
    except Exception:
        pass
    return (len(cores) or None), (len(sockets) or None)


def cpu_baseline(x, y, U, p, sample_pairs):
    """The CPU oracle (torch-CPU restatement of the reference's algorithm, oracle/ref_mirror.py) on a bounded sample of
    the same workload; used here ONLY as the timed baseline and as the checker of the parity guard.
    BASELINE.md section 3 protocol inside a ~25 s time box: 3 warm-ups and the median of 5 runs of the forward on the
    first `sample_pairs` pairs with this process's CPU share as torch threads (a one-GPU box gives a job 16 of the
    host's cores; `physical_cores` says what the host has); forward + backward (autograd through the restatement, as
    the reference's users get it) on half the sample, median of 3; one thread on one pair and a quarter of its slices."""
    from oracle import ref_mirror
    try:
        visible = len(os.sched_getaffinity(0))
    except Exception:
        visible = os.cpu_count() or 1
    cores = max(1, min(visible, int(os.environ.get("SHW_BENCH_CPU_THREADS", "16"))))   # GPU box: 16-core share per GPU
    phys, sockets = physical_cores()
    xs, ys, Us = x[:sample_pairs].cpu(), y[:sample_pairs].cpu(), U[:sample_pairs].cpu()
    sample_pairs = xs.shape[0]
    N, L = xs.shape[1], Us.shape[1]

    def timed(fn, warm, runs):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            val = fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), val

    def forward(k):
        return torch.stack([ref_mirror.per_slice_costs(xs[b], ys[b], Us[b], p=p).mean() for b in range(k)])

    def forward_backward(k):
        a, b = xs[:k].clone().requires_grad_(True), ys[:k].clone().requires_grad_(True)
        total = sum(ref_mirror.per_slice_costs(a[i], b[i], Us[i], p=p).mean() for i in range(k))
        total.backward()
        return total.detach()

    torch.set_num_threads(cores)
    log("cpu baseline on %d threads (%s, %s physical cores) ..." % (cores, cpu_model(), phys))
    t_all, pair_vals = timed(lambda: forward(sample_pairs), warm=3, runs=5)
    kb = max(1, sample_pairs // 2)
    t_fb, _ = timed(lambda: forward_backward(kb), warm=1, runs=3)
    L1 = max(1, L // 4)
    torch.set_num_threads(1)
    log("cpu baseline on 1 thread ...")
    t_one, _ = timed(lambda: ref_mirror.per_slice_costs(xs[0], ys[0], Us[0, :L1], p=p).mean(), warm=3, runs=5)
    torch.set_num_threads(cores)
    return {"value": sample_pairs * N * L / t_all, "unit": "point-pairs/s", "cores": cores, "kind": "port",
            "physical_cores": phys, "sockets": sockets, "host_threads_visible": visible,
            "value_allcores": sample_pairs * N * L / t_all, "value_fwd_bwd": kb * N * L / t_fb,
            "value_1thread": N * L1 / t_one, "cpu_model": cpu_model(),
            "sample": "threads=%d: forward on the first %d pairs of the same batch (N=%d, L=%d, p=%g), %.2f s per run, median "
                      "of 5 after 3 warm-ups; forward+backward on the first %d pairs, %.2f s per run, median of 3 after 1 "
                      "warm-up; threads=1: first pair, first %d slices, %.2f s per run, median of 5 after 3 warm-ups"
                      % (cores, sample_pairs, N, L, p, t_all, kb, t_fb, L1, t_one),
            "calibration": "the port runs 1.2x faster than the real reference in the build container (N=2048, L=128, 8 "
                           "threads: 0.221 s vs 0.266 s, values 8e-8 apart; BASELINE.md section 3)",
            "loss_of_sample": float(pair_vals.sum().item()),
            "_pair_losses": pair_vals.double()}


if __name__ == "__main__":
    sys.exit(main())
