#!/usr/bin/env python3
"""Phase times of the column-loop kernel from its s_memtime stamps (diagnostic library only; workgroup 0 of the LAST
block of one solve; cycles of the shader clock):
    python -m gptq_amd.build --diag && GPTQ_HIP_LIB=gptq_amd/libgptq_hip_diag.so python tools/qb_phases.py [RxC] [--groupsize G]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd, gptq_amd.gptq as gmod
from gptq_amd import _lib
gmod.VERBOSE = False
shape = next((a for a in sys.argv[1:] if "x" in a), "4096x1024")
R, n = (int(v) for v in shape.split("x"))
gs = int(sys.argv[sys.argv.index("--groupsize") + 1]) if "--groupsize" in sys.argv else -1
dev = torch.device("cuda:0")
X = torch.randn(2 * n, n, device=dev)
H = (X.t() @ X) * (2.0 / X.shape[0])
for _ in range(2):
    lin = torch.nn.Linear(n, R, bias=False, device=dev, dtype=torch.float16)
    g = gptq_amd.GPTQ(lin)
    g.quantizer = gptq_amd.Quantizer(); g.quantizer.configure(4, perchannel=True, sym=False, mse=False)
    g.H = H.clone(); g.nsamples = 2
    g.fasterquant(blocksize=128, percdamp=0.01, groupsize=gs, static_groups=gs > 0)
torch.cuda.synchronize()
lib = _lib.load()
out = (C.c_ulonglong * 32)()
fn = lib.gptq_diag_qb_stamps
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong)]
assert fn(out) == 0
t = [out[i] for i in range(14)]
names = ["prologue (loads, first U rows)"] + [f"phase {p} {q}" for p in range(4) for q in ("chain of 32 columns", "retire + next U rows", "barrier")][:11] + ["epilogue"]
print(f"{shape}: workgroup 0, cycles")
for i in range(13):
    print(f"  {names[i] if i < len(names) else i:34s} {t[i + 1] - t[i]:8d}")
print(f"  {'total':34s} {t[13] - t[0]:8d}")
out4 = (C.c_ulonglong * 4)()
for fname, what in (("gptq_diag_trailing_stamps", "last trailing_kernel launch (64 x 64 tile, workgroup 0)"),
                    ("gptq_diag_chain64_stamps", "last 64-tile launch of the chain (rtilde / syrk / panel, workgroup 0)")):
    f = getattr(lib, fname)
    f.restype = C.c_int
    f.argtypes = [C.POINTER(C.c_ulonglong)]
    assert f(out4) == 0
    u = [out4[i] for i in range(4)]
    print(f"{what}: first stage {u[1] - u[0]}, k loop {u[2] - u[1]}, epilogue {u[3] - u[2]}, total {u[3] - u[0]} cycles")
