#!/usr/bin/env python3
"""Per-lane duration and host enqueue time of fasterquant_many on the bench's block (no profiler attached)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd, gptq_amd.gptq as gm
gm.VERBOSE = False
dev = torch.device("cuda:0")
SHAPES = [("q", 2048, 2048), ("k", 2048, 2048), ("v", 2048, 2048), ("o", 2048, 2048), ("fc1", 8192, 2048), ("fc2", 2048, 8192)]
gen = torch.Generator(device=dev).manual_seed(0)
Hs = {}
for C in (2048, 8192):
    X = torch.randn(2 * C, C, device=dev, generator=gen) * (1 + torch.arange(C, device=dev) % 7)
    Hs[C] = (X.t() @ X) * (2.0 / X.shape[0])
for rep in range(4):
    sol = []
    for n, R, C in SHAPES:
        lin = torch.nn.Linear(C, R, bias=False, device=dev, dtype=torch.float16)
        lin.weight.data = (torch.randn(R, C, device=dev, generator=gen) * 0.02).half()
        g = gptq_amd.GPTQ(lin)
        g.quantizer = gptq_amd.Quantizer(); g.quantizer.configure(4, perchannel=True, sym=False, mse=False)
        g.H = Hs[C].clone(); g.nsamples = 2
        sol.append(g)
    torch.cuda.synchronize()
    gm.LANE_EVENTS = []
    t0 = time.perf_counter()
    gptq_amd.fasterquant_many(sol, blocksize=128, percdamp=0.01, groupsize=128, static_groups=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"rep {rep}: wall {wall * 1e3:.2f} ms; " + "; ".join(
        f"lane {j}: gpu {a.elapsed_time(b):.2f} ms, host enqueue {h * 1e3:.2f} ms" for j, a, b, h in gm.LANE_EVENTS), flush=True)
