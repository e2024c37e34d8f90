"""CPU checks of the boundary: the C-ABI library loads (through torch's HIP runtime, no GPU needed) and
exports every symbol include/shw.h declares; the Python mirror refuses CPU tensors loudly."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shw():
    import shw_amd
    if not os.path.exists(shw_amd._lib.LIB_PATH):
        shw_amd._lib.build()
    return shw_amd


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "shw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(shw_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_loader_binds(shw):
    assert declared_symbols() == sorted(shw._lib.EXPORTED_SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol(shw):
    lib = shw._lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.shw_abi_version() == shw._lib.ABI_VERSION == 3
    assert lib.shw_max_points() == 8192
    assert lib.shw_ssw_coef_bytes(2, 10, 20, 3) == 2 * 3 * 30 * 4


def test_one_hip_runtime_in_process(shw):
    shw._lib.load()
    assert len(shw._lib.hip_runtimes_mapped()) == 1


def test_argument_validation_needs_no_gpu(shw):
    lib = shw._lib.load()
    # null pointers / bad sizes are rejected before any HIP call
    assert lib.shw_ssw_forward(None, None, None, 1, 8, 8, 1, 0, 2.0, None, None, None) == 1
    assert lib.shw_ssw_reduce(None, 1, 1, 1.0, None, None, None) == 1
    assert lib.shw_chamfer_forward(None, None, 1, 1, 1, None, None, None, None, None, None) == 1


def test_cpu_tensors_are_refused_not_silently_computed(shw):
    x = torch.zeros(8, 3)
    U = torch.zeros(2, 3, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        shw.sliced_cost(x, x, U)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        shw.chamfer_distance(x[None], x[None])


def test_direction_sampling_on_cpu_matches_golden_stream(shw, golden):
    import numpy as np
    g = golden("g5_rng.npz")
    torch.manual_seed(int(g["seed"]))
    assert np.array_equal(shw.draw_directions(24, "cpu").numpy(), g["U_pair"])
    torch.manual_seed(int(g["seed"]))
    assert np.array_equal(shw.draw_directions(12, "cpu", batch=3).numpy(), g["U_batched"])


def test_bench_gpus_2_starts_its_ranks_and_fails_only_at_device_selection():
    """`python bench.py --gpus 2` on a box without GPUs must get through argument handling, start two
    ranks under torch.distributed.run and fail where they select their device -- not exit with a usage error."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert res.returncode != 0
    assert "starting 2 ranks" in res.stderr
    assert "No HIP GPUs are available" in res.stderr
    assert res.stdout.strip() == ""                      # no JSON line from a failed run
