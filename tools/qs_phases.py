#!/usr/bin/env python3
"""Phase times of quant_super_kernel from its s_memtime stamps (diagnostic library only; workgroup 0, first block of the
LAST super-block launch of one solve; cycles of the shader clock):
    python -m gptq_amd.build --diag && GPTQ_HIP_LIB=gptq_amd/libgptq_hip_diag.so python tools/qs_phases.py [RxC] [--actorder]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd, gptq_amd.gptq as gmod
from gptq_amd import _lib
gmod.VERBOSE = False
shape = next((a for a in sys.argv[1:] if "x" in a), "4096x1024")
R, n = (int(v) for v in shape.split("x"))
dev = torch.device("cuda:0")
X = torch.randn(2 * n, n, device=dev)
H = (X.t() @ X) * (2.0 / X.shape[0])
for _ in range(2):
    lin = torch.nn.Linear(n, R, bias=False, device=dev, dtype=torch.float16)
    g = gptq_amd.GPTQ(lin)
    g.quantizer = gptq_amd.Quantizer(); g.quantizer.configure(4, perchannel=True, sym=False, mse=False)
    g.H = H.clone(); g.nsamples = 2
    g.fasterquant(blocksize=128, percdamp=0.01, actorder="--actorder" in sys.argv)
torch.cuda.synchronize()
lib = _lib.load()
out = (C.c_ulonglong * 8)()
fn = lib.gptq_diag_qs_stamps
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong)]
assert fn(out) == 0
t = [out[i] for i in range(8)]
names = ["block prologue (grp, barrier, loads of w / grids / w0)", "phase 0", "phase 1", "phase 2", "phase 3", "retire",
         "near update (+ barrier)"]
print(f"{shape}, GPTQ_QS_LANES={os.environ.get('GPTQ_QS_LANES', 'default')}: workgroup 0, block 0, cycles")
for i in range(7):
    print(f"  {names[i]:56s} {t[i + 1] - t[i]:8d}")
print(f"  {'total':56s} {t[7] - t[0]:8d}")
