"""ctypes binding of the C-ABI library (include/shw.h).  The HIP library is the product: if it is
missing or cannot be loaded this module raises -- there is no CPU or PyTorch fallback."""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch  # noqa: F401  (imported first on purpose: the process must hold ONE HIP runtime -- torch's)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHW_LIB_PATH") or os.path.join(_HERE, "libshw_hip.so")   # override: kernel A/B builds
CSRC = os.path.join(_HERE, "csrc")

_c_f32p = ctypes.c_void_p
ABI_VERSION = 3           # include/shw.h SHW_ABI_VERSION
CIRCLE_AS_SLICED, CIRCLE_BISECTION, CIRCLE_LEVEL_MEDIAN = 0, 1, 2      # include/shw.h SHW_CIRCLE_*
_SIGNATURES = {
    # name: (restype, argtypes)
    "shw_abi_version": (ctypes.c_int, []),
    "shw_max_points": (ctypes.c_int, []),
    "shw_stiefel_frames": (ctypes.c_int, [_c_f32p, ctypes.c_long, _c_f32p, ctypes.c_void_p]),
    "shw_ssw_forward": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_long, ctypes.c_float, _c_f32p, ctypes.c_void_p,
                                       ctypes.c_void_p]),
    "shw_ssw_reduce": (ctypes.c_int, [_c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_float, _c_f32p, _c_f32p,
                                      ctypes.c_void_p]),
    "shw_ssw_coef_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "shw_ssw_forward_grad": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_long, ctypes.c_float, _c_f32p, ctypes.c_void_p,
                                            _c_f32p, _c_f32p, ctypes.c_void_p]),
    "shw_ssw_forward_general": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_long, ctypes.c_long,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                               ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p]),
    "shw_ssw_backward_points": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                               ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p,
                                               ctypes.c_void_p]),
    "shw_circle_ot": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, _c_f32p, _c_f32p, _c_f32p,
                                     _c_f32p, ctypes.c_void_p]),
    "shw_esw_forward": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                       ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p]),
    "shw_esw_backward_points": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_long, _c_f32p, _c_f32p, ctypes.c_void_p]),
    "shw_esw_backward_dirs": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, _c_f32p, ctypes.c_void_p]),
    "shw_sinkhorn_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "shw_sinkhorn_forward": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                            _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p]),
    "shw_sinkhorn_train_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "shw_sinkhorn_forward_train": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                                  _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p]),
    "shw_sinkhorn_backward": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, _c_f32p, _c_f32p,
                                             _c_f32p, ctypes.c_void_p]),
    "shw_chamfer_forward": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_f32p,
                                           ctypes.c_void_p, _c_f32p, ctypes.c_void_p, _c_f32p, ctypes.c_void_p]),
    "shw_chamfer_backward": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_void_p, ctypes.c_void_p, _c_f32p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_f32p, _c_f32p,
                                            ctypes.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libshw_hip.so (in-tree) with hipcc; returns the path."""
    cmd = ["make", "-j8", "-C", CSRC] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libshw_hip.so failed:\n" + res.stdout)
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load the HIP library.  Raises if it is absent: the HIP path is the only path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "There is no fallback implementation.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.shw_abi_version() != ABI_VERSION:
        raise RuntimeError("libshw_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError_t {rc}"
                           + (" (invalid value / unsupported size)" if rc == 1 else ""))


def hip_runtimes_mapped() -> list[str]:
    """Paths of every libamdhip64 mapped into this process (should be exactly one)."""
    seen = []
    with open("/proc/self/maps") as fh:
        for line in fh:
            if "libamdhip64" in line:
                path = line.split()[-1]
                if path not in seen:
                    seen.append(path)
    return seen
