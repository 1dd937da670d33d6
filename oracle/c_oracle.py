"""ctypes access to oracle/_build/liboracle_c.so (TEST INFRASTRUCTURE; see oracle_c.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def load(build=True):
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH) and build:
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pack(iw: np.ndarray, bits: int) -> np.ndarray:
    iw = np.ascontiguousarray(iw, dtype=np.uint32)
    n_in, n_out = iw.shape
    out = np.zeros((n_in // 32 * 3 if bits == 3 else n_in // 8, n_out), dtype=np.int32)
    getattr(load(), "oracle_pack3" if bits == 3 else "oracle_pack4")(_p(iw), n_in, n_out, _p(out))
    return out


def vecquant3matmul(vec, mat, mul, scales, zeros):
    """In place on `mul` (fp32), like the reference kernel."""
    vec = np.ascontiguousarray(vec, dtype=np.float32)
    mat = np.ascontiguousarray(mat, dtype=np.int32)
    s = np.ascontiguousarray(scales, dtype=np.float32).reshape(-1)
    z = np.ascontiguousarray(zeros, dtype=np.float32).reshape(-1)
    assert mul.dtype == np.float32 and mul.flags.c_contiguous
    load().oracle_vecquant3matmul_f32(_p(vec), _p(mat), _p(mul), _p(s), _p(z), C.c_int(mat.shape[0]), C.c_int(mat.shape[1]))
    return mul
