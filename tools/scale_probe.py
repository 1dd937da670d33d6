#!/usr/bin/env python3
"""Scale sanity: fasterquant at LLaMA-7B / 65B Linear shapes (BASELINE configs[2], [4]) with the flags
those configs use.  python3 tools/scale_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gptq_amd, gptq_amd.gptq as gmod
gmod.VERBOSE = False
gmod.HESSIAN_DEFER = 8
dev = torch.device("cuda:0")
cases = [("llama7b q_proj act-order", 4096, 4096, dict(actorder=True)),
         ("llama7b down_proj act-order", 4096, 11008, dict(actorder=True)),
         ("llama7b gate_proj g128 act-order static", 11008, 4096, dict(groupsize=128, actorder=True, static_groups=True)),
         ("llama65b q_proj", 8192, 8192, dict()),
         ("llama65b down_proj", 8192, 22016, dict())]
for name, R, C, kw in cases:
    gen = torch.Generator(device=dev).manual_seed(0)
    lin = torch.nn.Linear(C, R, bias=False, device=dev, dtype=torch.float16)
    lin.weight.data = (torch.randn(R, C, device=dev, generator=gen) * 0.02).half()
    g = gptq_amd.GPTQ(lin)
    g.quantizer = gptq_amd.Quantizer(); g.quantizer.configure(4, perchannel=True, sym=False, mse=False)
    chan = (1 + torch.arange(C, device=dev) % 7).half()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for j in range(16):
        g.add_batch(torch.randn(1, 2048, C, device=dev, generator=gen, dtype=torch.float16) * chan, None)
    gmod.flush_pending(); torch.cuda.synchronize(); t1 = time.perf_counter()
    g.fasterquant(blocksize=128, percdamp=0.01, **kw)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    w = lin.weight.data
    ok = bool(torch.isfinite(w).all()) and int(g.codes.max()) <= 15
    print(f"{name}: {R}x{C}  hessian(16 samples) {1e3*(t1-t0):.1f} ms  fasterquant {1e3*(t2-t1):.1f} ms = {R*C/(t2-t1)/1e6:.0f} Mparams/s"
          f"  error {g.error:.4g}  finite/on-grid {ok}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    g.free(); del g, lin
    torch.cuda.empty_cache()
