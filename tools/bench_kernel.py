#!/usr/bin/env python3
"""Kernel benchmark for the packed mat-vec and the packer (BASELINE configs[3]).

The reference's README (lines 11, 92) points to `test_kernel.py`, which is absent from the fork;
this is the equivalent: FC2 of OPT-66B (in = 36864, out = 9216), single token, comparing a dense
fp16 GEMV with vecquant3matmul (fp32 and fp16 activations) and vecquant4matmul, and checking the
quantized output against the dense one.  Reports microseconds per call and achieved HBM GB/s on the
ALGORITHMIC bytes (in/32*bits*out*4 + in*sizeof(x) + out*16), MI355X peak 8000 GB/s.

    python tools/bench_kernel.py [--in 36864 --out 9216 --iters 200]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd
from gptq_amd import quant_cuda

PEAK = 8000.0


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in", dest="n_in", type=int, default=36864)
    ap.add_argument("--out", dest="n_out", type=int, default=9216)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    res = {"in": a.n_in, "out": a.n_out}
    x32 = torch.randn(a.n_in, device=dev, generator=gen)
    x16 = x32.half()
    W16 = (torch.randn(a.n_out, a.n_in, device=dev, generator=gen) * 0.02).half()
    t = timeit(lambda: torch.mv(W16, x16), a.iters)
    res["fp16_gemv"] = {"us": round(t, 2), "GBps": round(W16.numel() * 2 / t / 1e3, 1)}
    for bits in (3, 4):
        codes = torch.randint(0, 2 ** bits, (a.n_out, a.n_in), device=dev, generator=gen, dtype=torch.uint8)
        scales = torch.rand(a.n_out, 1, device=dev, generator=gen) * 0.01 + 1e-3
        zeros = torch.randint(0, 2 ** bits, (a.n_out, 1), device=dev, generator=gen).float() * scales
        tp = timeit(lambda: gptq_amd.pack_codes(codes, bits), max(10, a.iters // 10))
        qw = gptq_amd.pack_codes(codes, bits)
        res[f"pack{bits}_codes"] = {"us": round(tp, 2), "GBps": round((codes.numel() + qw.numel() * 4) / tp / 1e3, 1)}
        dense = scales * codes.float() - zeros
        ref = dense.double() @ x32.double()
        fn = quant_cuda.vecquant3matmul if bits == 3 else quant_cuda.vecquant4matmul
        fn16 = quant_cuda.vecquant3matmul_faster if bits == 3 else quant_cuda.vecquant4matmul
        for tag, x, f in (("fp32", x32, fn), ("fp16", x16, fn16)):
            y = torch.zeros(a.n_out, device=dev)
            f(x, qw, y, scales, zeros)
            err = float((y.double() - ref).abs().max() / ref.abs().max())
            t = timeit(lambda: f(x, qw, y, scales, zeros), a.iters)
            nbytes = qw.numel() * 4 + x.numel() * x.element_size() + a.n_out * 16
            res[f"vecquant{bits}matmul_{tag}"] = {"us": round(t, 2), "GBps": round(nbytes / t / 1e3, 1),
                                                   "frac_hbm_peak": round(nbytes / t / 1e3 / PEAK, 3),
                                                   "bytes": nbytes, "max_rel_err": err}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
