#!/usr/bin/env python3
"""Small driver for profiling the Hessian kernel alone:
    python3 tools/hessian_probe.py [--defer N] [C ...]
Each timed launch folds N samples of S = 2048 tokens (fp16) into H, like bench.py does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd, gptq_amd.gptq as gmod

argv = sys.argv[1:]
defer = 1
if argv and argv[0] == "--defer":
    defer = int(argv[1]); argv = argv[2:]
gmod.HESSIAN_DEFER = defer
dev = torch.device("cuda:0")
sizes = [int(a) for a in argv] or [2048, 8192]
S, reps = 2048, 6
for C in sizes:
    lin = torch.nn.Linear(C, 8, bias=False, device=dev, dtype=torch.float16)
    g = gptq_amd.GPTQ(lin)
    xs = [torch.randn(1, S, C, device=dev, dtype=torch.float16) for _ in range(reps * defer)]
    for x in xs[:defer]:
        g.add_batch(x, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in xs[defer:]:
        g.add_batch(x, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps - 1)
    print(f"C={C} defer={defer}: {dt * 1e6:.1f} us/launch, {defer * S * C * C / dt / 1e12:.1f} TFLOP/s algorithmic", flush=True)
