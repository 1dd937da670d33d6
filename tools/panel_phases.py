#!/usr/bin/env python3
"""Timeline of one chol_panel_kernel launch from its s_memtime stamps (diagnostic library only, 100 MHz):
    python -m gptq_amd.build --diag && GPTQ_HIP_LIB=gptq_amd/libgptq_hip_diag.so python tools/panel_phases.py [C] [p0]
runs gptq_rfactor_upper and prints, for the launch of outer panel p0 (a multiple of 4), the chain's and the first two
slabs' stamps relative to the chain's first one."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gptq_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load()
dev = torch.device("cuda:0")
X = torch.randn(2 * n, n, device=dev)
H = (X.t() @ X) * (2.0 / X.shape[0])
nb = lib.gptq_hinv_workspace_bytes(n)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
out = (C.c_ulonglong * 96)()
fn = lib.gptq_diag_panel_stamps
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert fn(out, p0) == 0
for _ in range(3):
    Hc = H.clone()
    _lib.call("gptq_rfactor_upper", _lib.ptr(Hc), Hc.stride(0), n, 0.01, None, _lib.ptr(info), _lib.ptr(ws), nb,
              _lib.stream(dev))
torch.cuda.synchronize()
assert fn(out, p0) == 0
t = [out[i] for i in range(96)]
t0 = t[0]
us = lambda v: (v - t0) / 100.0
print(f"C = {n}, outer panel p0 = {p0} (us after the chain's start)")
for k in range(4):
    c = t[4 * k: 4 * k + 4]
    print(f"chain step {k}: wait from {us(c[0]):7.2f} to {us(c[1]):7.2f}, block done {us(c[2]):7.2f}, published {us(c[3]):7.2f}")
for s in range(2):
    for k in range(4):
        b = t[16 + 32 * s + 8 * k: 16 + 32 * s + 8 * k + 7]
        if not b[0] or b[0] < t0: continue
        print(f"slab {s} step {k}: at {us(b[0]):7.2f}, D seen {us(b[1]):7.2f}, panel done {us(b[2]):7.2f}, published {us(b[3]):7.2f}, "
              f"operands seen {us(b[4]):7.2f}, updates done {us(b[5]):7.2f}, published {us(b[6]):7.2f}")
