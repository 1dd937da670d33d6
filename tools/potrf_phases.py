#!/usr/bin/env python3
"""Phase times of the diagonal-block kernel from its s_memtime stamps (diagnostic library only):
    python -m gptq_amd.build --diag && GPTQ_HIP_LIB=gptq_amd/libgptq_hip_diag.so python tools/potrf_phases.py [C]
runs one chain (gptq_rfactor_upper) and prints the phases of the LAST diagonal block (s_memtime counts at 100 MHz)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gptq_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = _lib.load()
dev = torch.device("cuda:0")
X = torch.randn(2 * n, n, device=dev)
H = (X.t() @ X) * (2.0 / X.shape[0])
nb = lib.gptq_hinv_workspace_bytes(n)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
if os.environ.get("POTRF_ABLATE"):
    lib.gptq_diag_potrf_ablate.argtypes = [C.c_int]
    lib.gptq_diag_potrf_ablate(int(os.environ["POTRF_ABLATE"]))
for _ in range(3):
    Hc = H.clone()
    _lib.call("gptq_rfactor_upper", _lib.ptr(Hc), Hc.stride(0), n, 0.01, None, _lib.ptr(info), _lib.ptr(ws), nb,
              _lib.stream(dev))
torch.cuda.synchronize()
out = (C.c_ulonglong * 17)()
fn = lib.gptq_diag_potrf_stamps
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong)]
assert fn(out) == 0
t = [out[i] for i in range(17)]
names = ["load", "mirror"] + [f"s{s} {p}" for s in range(4) for p in ("A1 factor+inverse 32x32", "A2 sub-panel", "A3 update")] + ["inverse assembly", "store"]
for i, nm in enumerate(names):
    print(f"{nm:28s} {(t[i + 1] - t[i]) / 100.0:7.2f} us")
print(f"{'total':28s} {(t[16] - t[0]) / 100.0:7.2f} us")
