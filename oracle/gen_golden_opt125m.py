#!/usr/bin/env python3
"""PPL-proxy golden at a BASELINE size (configs[0]): the REFERENCE's opt_sequential + opt_eval (opt.py:29-359) on a
random-init OPT-125m ARCHITECTURE (hidden 768, ffn 3072, 12 layers, 12 heads, seqlen 2048; vocabulary cut to 2048 to
keep the embedding light), nsamples = 32 synthetic samples, CPU fp32, in the build container:

    MPLBACKEND=Agg python oracle/gen_golden_opt125m.py

Numbers only (no weights: the model is `torch.manual_seed(0); OPTForCausalLM(cfg)`, CPU RNG, reproducible from the
seed): calibration / eval tokens, the per-Linear `error` sequence (gptq.py:294), and the fp / RTN-4 / GPTQ-4
perplexities -> tests/golden/g6_opt125m.npz.  It is a PROXY: Wiki2 and real checkpoints are not available offline.
"""
import contextlib
import io
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden_driver as GD  # imports the reference's opt.py, stubs the CUDA calls  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
from transformers import OPTConfig, OPTForCausalLM  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g6_opt125m.npz")
VOCAB, SEQ, NS = 2048, 2048, 32


def model125m():
    cfg = OPTConfig(vocab_size=VOCAB, hidden_size=768, ffn_dim=3072, num_hidden_layers=12, num_attention_heads=12,
                    max_position_embeddings=2048, word_embed_proj_dim=768, do_layer_norm_before=True,
                    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, layerdrop=0.0)
    torch.manual_seed(0)
    m = OPTForCausalLM(cfg).float().eval()
    m.seqlen = SEQ
    return m


def main():
    torch.set_num_threads(8)
    gen = torch.Generator().manual_seed(1)
    calib = torch.randint(0, VOCAB, (NS, 1, SEQ), generator=gen)
    test = torch.randint(0, VOCAB, (1, SEQ * 4), generator=gen)
    out = {"calib": calib.numpy().astype(np.int16), "test": test.numpy().astype(np.int16)}
    t0 = time.time()
    out["ppl_fp"] = np.float64(GD.ppl_of(model125m(), test, GD.run_args(wbits=16, nsamples=NS, model="opt-125m-arch")))
    print("fp", out["ppl_fp"], time.time() - t0, flush=True)
    out["ppl_rtn4"] = np.float64(GD.ppl_of(model125m(), test, GD.run_args(nearest=True, nsamples=NS, model="opt-125m-arch")))
    print("rtn4", out["ppl_rtn4"], time.time() - t0, flush=True)
    model = model125m()
    args = GD.run_args(nsamples=NS, model="opt-125m-arch")
    GD.ref_opt.args = args
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        quantizers = GD.ref_opt.opt_sequential(model, [(calib[i], None) for i in range(NS)], torch.device("cpu"))
    lines = buf.getvalue().splitlines()
    out["errors"] = np.array([float(l.split()[1]) for l in lines if l.startswith("error")])
    out["names_in_order"] = np.array([l.split()[1] for l in lines if len(l.split()) == 2 and l.split()[0].isdigit()])
    print("quantized", len(quantizers), time.time() - t0, flush=True)
    out["ppl_gptq4"] = np.float64(GD.ppl_of(model, test, args))
    np.savez_compressed(OUT, **out)
    print("ppl fp / rtn4 / gptq4:", out["ppl_fp"], out["ppl_rtn4"], out["ppl_gptq4"])
    print(len(out["errors"]), "errors;", os.path.getsize(OUT) / 1024, "KiB")


if __name__ == "__main__":
    main()
