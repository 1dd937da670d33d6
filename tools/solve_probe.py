#!/usr/bin/env python3
"""Small driver for profiling the solve alone: `python3 tools/solve_probe.py [RxC ...]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd, gptq_amd.gptq as gmod
gmod.VERBOSE = False
dev = torch.device("cuda:0")
ACT = "--actorder" in sys.argv
GS = int(sys.argv[sys.argv.index("--groupsize") + 1]) if "--groupsize" in sys.argv else -1
REPS = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(2048, 2048), (2048, 8192)]
for R, C in shapes:
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(2 * C, C, device=dev, generator=gen) * (1 + torch.arange(C, device=dev) % 7)
    H0 = (X.t() @ X) * (2.0 / X.shape[0])
    W = (torch.randn(R, C, device=dev, generator=gen) * 0.02).half()
    times = []
    for rep in range(REPS):
        lin = torch.nn.Linear(C, R, bias=False, device=dev, dtype=torch.float16)
        lin.weight.data = W.clone()
        g = gptq_amd.GPTQ(lin)
        g.quantizer = gptq_amd.Quantizer(); g.quantizer.configure(4, perchannel=True, sym=False, mse=False)
        g.H = H0.clone(); g.nsamples = 2
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.fasterquant(blocksize=128, percdamp=0.01, groupsize=GS, static_groups=GS > 0, actorder=ACT)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    print(f"{R}x{C}: fasterquant {min(times) * 1e3:.2f} ms = {R * C / min(times) / 1e6:.0f} Mparams/s", flush=True)
