#!/usr/bin/env python3
"""Per-lane GPU time and host enqueue time of fasterquant_many on the bench's block (no profiler attached).
    python3 tools/lane_probe.py [--hooks N]     N > 0: feed N calibration samples through add_batch first (q/k/v share
                                                their input), so that LAZY_HESSIANS leaves work for the side lanes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd, gptq_amd.gptq as gm
gm.VERBOSE = False
gm.HESSIAN_DEFER = 8
dev = torch.device("cuda:0")
hooks = int(sys.argv[sys.argv.index("--hooks") + 1]) if "--hooks" in sys.argv else 0
SHAPES = [("q", 2048, 2048), ("k", 2048, 2048), ("v", 2048, 2048), ("o", 2048, 2048), ("fc1", 8192, 2048), ("fc2", 2048, 8192)]
gen = torch.Generator(device=dev).manual_seed(0)
Hs = {}
for C in (2048, 8192):
    X = torch.randn(2 * C, C, device=dev, generator=gen) * (1 + torch.arange(C, device=dev) % 7)
    Hs[C] = (X.t() @ X) * (2.0 / X.shape[0])
acts = {}
if hooks:
    mk = lambda C: torch.randn(hooks, 2048, C, device=dev, generator=gen, dtype=torch.float16)
    qkv = mk(2048)
    acts = {"q": qkv, "k": qkv, "v": qkv, "o": mk(2048), "fc1": mk(2048), "fc2": mk(8192)}
for rep in range(4):
    sol = []
    for n, R, C in SHAPES:
        lin = torch.nn.Linear(C, R, bias=False, device=dev, dtype=torch.float16)
        lin.weight.data = (torch.randn(R, C, device=dev, generator=gen) * 0.02).half()
        g = gptq_amd.GPTQ(lin)
        g.quantizer = gptq_amd.Quantizer(); g.quantizer.configure(4, perchannel=True, sym=False, mse=False)
        if not hooks:
            g.H = Hs[C].clone(); g.nsamples = 2
        sol.append(g)
    if hooks:
        for j in range(hooks):
            for g, (n, R, C) in zip(sol, SHAPES):
                g.add_batch(acts[n][j:j + 1], None)
    torch.cuda.synchronize()
    gm.LANE_EVENTS = []
    t0 = time.perf_counter()
    gptq_amd.fasterquant_many(sol, blocksize=128, percdamp=0.01, groupsize=128, static_groups=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"rep {rep}: wall {wall * 1e3:.2f} ms; " + "; ".join(
        f"lane {j}: gpu {a.elapsed_time(b):.2f} ms, host enqueue {h * 1e3:.2f} ms" for j, a, b, h in gm.LANE_EVENTS), flush=True)
