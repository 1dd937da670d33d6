#!/usr/bin/env python3
"""Host time of the bench's hook loop (768 add_batch calls) against the time until the GPU has folded what was not
deferred: shows whether the Hessian phase is host- or GPU-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gptq_amd, gptq_amd.gptq as gm
gm.VERBOSE=False; gm.HESSIAN_DEFER=16
dev=torch.device("cuda:0")
SH=[("q",2048,2048),("k",2048,2048),("v",2048,2048),("o",2048,2048),("fc1",8192,2048),("fc2",2048,8192)]
mk=lambda C: torch.randn(128,2048,C,device=dev,dtype=torch.float16)
qkv=mk(2048); acts={"q":qkv,"k":qkv,"v":qkv,"o":mk(2048),"fc1":mk(2048),"fc2":mk(8192)}
for rep in range(3):
    sol=[]
    for n,R,C in SH:
        lin=torch.nn.Linear(C,R,bias=False,device=dev,dtype=torch.float16)
        g=gptq_amd.GPTQ(lin); g.quantizer=gptq_amd.Quantizer(); g.quantizer.configure(4,perchannel=True,sym=False,mse=False); sol.append(g)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for j in range(128):
        for g,(n,R,C) in zip(sol,SH):
            g.add_batch(acts[n][j:j+1],None)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"hooks host {1e3*(t1-t0):.2f} ms, until GPU idle {1e3*(t2-t0):.2f} ms")
    for g in sol: g.free()
