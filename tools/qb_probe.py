#!/usr/bin/env python3
"""Kernel-level timing of the solve of one Linear by kernel name (rocprofv3-free: HIP events around fasterquant)
and, with GPTQ_QB_ABLATE=1|2|3, of diagnostic column-loop builds (timing only, results wrong).
    python3 tools/qb_probe.py [RxC ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd, gptq_amd.gptq as gmod
gmod.VERBOSE = False
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2048, 8192)]
for R, C in shapes:
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(2 * C, C, device=dev, generator=gen) * (1 + torch.arange(C, device=dev) % 7)
    H0 = (X.t() @ X) * (2.0 / X.shape[0])
    W = (torch.randn(R, C, device=dev, generator=gen) * 0.02).half()
    times = []
    for rep in range(4):
        lin = torch.nn.Linear(C, R, bias=False, device=dev, dtype=torch.float16)
        lin.weight.data = W.clone()
        g = gptq_amd.GPTQ(lin)
        g.quantizer = gptq_amd.Quantizer(); g.quantizer.configure(4, perchannel=True, sym=False, mse=False)
        g.H = H0.clone(); g.nsamples = 2
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            g.fasterquant(blocksize=128, percdamp=0.01, groupsize=128, static_groups=True)
        except Exception as e:      # ablation builds may produce garbage
            print("  (", type(e).__name__, ")")
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    print(f"QB_ABLATE={os.environ.get('GPTQ_QB_ABLATE', '0')} {R}x{C}: fasterquant {min(times) * 1e3:.2f} ms", flush=True)
