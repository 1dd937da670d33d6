#!/usr/bin/env python3
"""One LLaMA-7B-sized decoder block (random init) through gptq_amd.sequential at full width: act-order + static groups,
all Linears hooked at once (q/k/v and gate/up share inputs -> shared Hessians, joint solves, lazy Hessians).
    python3 tools/llama_block_probe.py [nsamples]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers import LlamaConfig, LlamaForCausalLM
import gptq_amd.gptq as gm
from gptq_amd.sequential import QuantArgs, llama_sequential

gm.VERBOSE = False
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
cfg = LlamaConfig(vocab_size=1024, hidden_size=4096, intermediate_size=11008, num_hidden_layers=1,
                  num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=2048)
torch.manual_seed(0)
model = LlamaForCausalLM(cfg).half().eval()
model.seqlen = 2048
gen = torch.Generator().manual_seed(1)
calib = [(torch.randint(0, 1024, (1, 2048), generator=gen), None) for _ in range(n)]
for ts in (False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    q = llama_sequential(model, calib, dev, QuantArgs(wbits=4, nsamples=n, groupsize=128, act_order=True,
                                                      static_groups=True, true_sequential=ts))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    params = sum(p.numel() for name, p in model.model.layers[0].named_parameters() if p.dim() == 2)
    from gptq_amd.sequential import quantize_sequential
    errs = {r["name"].split(".")[-1]: round(r["error"], 1) for r in quantize_sequential.last_records}
    print(f"true_sequential={ts}: {len(q)} Linears, {params / 1e6:.0f} M params in {dt:.2f} s "
          f"(incl. {2 * n} block forwards); errors {errs}; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
