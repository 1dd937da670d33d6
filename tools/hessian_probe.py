#!/usr/bin/env python3
"""Small driver for profiling the Hessian kernel alone: `python3 tools/hessian_probe.py [C ...]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd

dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [2048, 8192]
S, reps = 2048, 6
for C in sizes:
    lin = torch.nn.Linear(C, 8, bias=False, device=dev, dtype=torch.float16)
    g = gptq_amd.GPTQ(lin)
    xs = [torch.randn(1, S, C, device=dev, dtype=torch.float16) for _ in range(reps)]
    g.add_batch(xs[0], None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in xs[1:]:
        g.add_batch(x, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps - 1)
    print(f"C={C}: {dt * 1e6:.1f} us/launch, {S * C * C / dt / 1e12:.1f} TFLOP/s algorithmic", flush=True)
