"""ctypes binding of libgptq_hip.so (the C ABI in include/gptq_hip.h).

PyTorch is plumbing here: device memory, streams.  Tensors cross the boundary as
raw device pointers + sizes.  There is no fallback: if the library is missing, or
a tensor is not on a GPU, the call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPTQ_HIP_LIB: load another build of the same ABI (tools use it for the diagnostic library, gptq_amd.build --diag)
LIB_PATH = os.environ.get("GPTQ_HIP_LIB") or os.path.join(_HERE, "libgptq_hip.so")

F32, F16, BF16 = 0, 1, 2
_DTYPES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

_p, _i, _f, _z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_SIGNATURES = {
    "gptq_hip_abi_version": (C.c_int, []),
    "gptq_last_error": (C.c_char_p, []),
    "gptq_hessian_accum": (C.c_int, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "gptq_hessian_accum_multi": (C.c_int, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "gptq_hessian_accum_group": (C.c_int, [_i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p]),
    "gptq_hessian_accum_mixed": (C.c_int, [_i, _p, _p, _p, _i, _i, _p, _p, _i, _p, _i, _i, _p]),
    "gptq_symmetrize": (C.c_int, [_p, _i, _i, _p]),
    "gptq_find_params": (C.c_int, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p]),
    "gptq_quantize_rows": (C.c_int, [_p, _i, _i, _i, _p, _p, _i, _p]),
    "gptq_hinv_workspace_bytes": (_z, [_i]),
    "gptq_hinv_upper": (C.c_int, [_p, _i, _i, _f, _p, _p, _p, _z, _p]),
    "gptq_rfactor_upper": (C.c_int, [_p, _i, _i, _f, _p, _p, _p, _z, _p]),
    "gptq_fasterquant_factor_form": (C.c_int, [_i, _i, _i, _i]),
    "gptq_quant_block": (C.c_int, [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _i, _p, _p, _i, _p, _p, _p]),
    "gptq_fasterquant_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "gptq_fasterquant": (C.c_int, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p,
                                   _p, _p, _p, _z, _p]),
    "gptq_fasterquant_rows": (C.c_int, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p,
                                        _p, _p, _p, _p, _z, _p]),
    "gptq_chol_begin": (C.c_int, [_p, _i, _i, _f, _p, _p, _p, _z, _p]),
    "gptq_chol_panel": (C.c_int, [_p, _i, _i, _p, _p]),
    "gptq_chol_update": (C.c_int, [_p, _i, _i, _i, _i, _p]),
    "gptq_chol_end": (C.c_int, [_p, _i, _i, _p, _p]),
    "gptq_solve_prepare": (C.c_int, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "gptq_fasterquant_rows_factored": (C.c_int, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p, _p,
                                                _p, _p, _p, _p, _p, _p, _z, _p]),
    "gptq_pack_weights": (C.c_int, [_p, _i, _i, _i, _i, _p, _p, _i, _p, _p]),
    "gptq_pack_codes": (C.c_int, [_p, _i, _i, _i, _i, _p, _p]),
    "gptq_dequant_packed": (C.c_int, [_p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _p]),
    "gptq_vecquant3matmul": (C.c_int, [_p, _i, _p, _p, _p, _p, _i, _i, _p]),
    "gptq_vecquant4matmul": (C.c_int, [_p, _i, _p, _p, _p, _p, _i, _i, _p]),
    "gptq_vecquant_matmul_grouped": (C.c_int, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class GptqHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library (no GPU needed to load it).  A clean checkout carries sources only (*.so is git-ignored) and
    NOTHING is built here: N ranks importing at once would race on the object files and the link target, and a compiler
    launched from a process that a profiler has preloaded (rocprofv3 --pmc) or that has initialised the GPU is an exec this
    pool forbids.  Build explicitly, once, before any GPU process starts: `python -m gptq_amd.build` (what
    `__graft_entry__.build()` runs; it links to a temporary name and publishes with an atomic rename).  There is no
    non-HIP fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GptqHipError(f"{LIB_PATH} not found: build it first with `python -m gptq_amd.build` (hipcc, gfx950). "
                               "gptq_amd never builds at import time and has no non-HIP fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.gptq_last_error().decode(errors="replace")
        kind = {1: "invalid argument", 2: "HIP error", 3: "unsupported"}.get(rc, f"error {rc}")
        if rc == 3:
            raise NotImplementedError(f"{name}: {msg}")
        raise GptqHipError(f"{name}: {kind}: {msg}")


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise GptqHipError(f"{what} must live on an MI355X device (got {t.device}); gptq_amd has no CPU path")


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}") from None


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream
