#!/usr/bin/env python3
"""Driver-level golden: the REFERENCE's opt_sequential + opt_eval on a tiny, locally
constructed OPT (random init, synthetic tokens), CPU, in the build container.

    MPLBACKEND=Agg python oracle/gen_golden_driver.py

Stores the model's initial state_dict, the calibration / eval tokens, the per-Linear `error`
sequence the reference prints (gptq.py:294), every quantized Linear weight and the perplexities
(fp32 baseline, GPTQ 4-bit, RTN 4-bit) into tests/golden/g6_opt_tiny.npz.
"""
import contextlib
import io
import os
import sys
from types import SimpleNamespace

os.environ.setdefault("MPLBACKEND", "Agg")
REF = os.environ.get("GPTQ_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch

torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda *a, **k: None
with contextlib.redirect_stdout(io.StringIO()):
    import opt as ref_opt
from transformers import OPTConfig, OPTForCausalLM

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g6_opt_tiny.npz")


def tiny_model():
    cfg = OPTConfig(vocab_size=128, hidden_size=64, ffn_dim=256, num_hidden_layers=2, num_attention_heads=4,
                    max_position_embeddings=128, word_embed_proj_dim=64, do_layer_norm_before=True,
                    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, layerdrop=0.0)
    torch.manual_seed(0)
    m = OPTForCausalLM(cfg).float().eval()
    m.seqlen = 128
    return m


def run_args(**kw):
    base = dict(nsamples=4, wbits=4, sym=False, percdamp=0.01, groupsize=-1, act_order=False, static_groups=True,
                trits=False, nearest=False, layermix=False, linearmix=False, quant_config=None, lut_eval=False,
                columnwise=False, non_linear=False, bcq=False, model="tiny-opt", new_eval=False)
    base.update(kw)
    return SimpleNamespace(**base)


def ppl_of(model, tokens, args):
    ref_opt.args = args
    buf = io.StringIO()
    cwd = os.getcwd()
    os.makedirs("/tmp/g6/quant_bit", exist_ok=True)     # opt_eval appends to quant_bit/ppl.txt (opt.py:335-357)
    os.chdir("/tmp/g6")
    try:
        with contextlib.redirect_stdout(buf):
            ref_opt.opt_eval(model, SimpleNamespace(input_ids=tokens), torch.device("cpu"))
    finally:
        os.chdir(cwd)
    vals = [float(l) for l in buf.getvalue().splitlines() if l.replace(".", "", 1).replace("e-", "", 1).replace("e+", "", 1).isdigit()]
    return vals[-1]


def main():
    gen = torch.Generator().manual_seed(1)
    calib = torch.randint(0, 128, (4, 1, 128), generator=gen)
    test = torch.randint(0, 128, (1, 128 * 6), generator=gen)
    model = tiny_model()
    init = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    out = {"calib": calib.numpy(), "test": test.numpy()}
    for k, v in init.items():
        out["init/" + k] = v

    out["ppl_fp"] = np.float64(ppl_of(tiny_model(), test, run_args(wbits=16)))
    out["ppl_rtn4"] = np.float64(ppl_of(tiny_model(), test, run_args(nearest=True)))

    args = run_args()
    ref_opt.args = args
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        quantizers = ref_opt.opt_sequential(model, [(calib[i], None) for i in range(4)], torch.device("cpu"))
    errors = [float(l.split()[1]) for l in buf.getvalue().splitlines() if l.startswith("error")]
    names = sorted(quantizers)
    out["errors"] = np.array(errors)
    out["names_in_order"] = np.array([l.split()[1] for l in buf.getvalue().splitlines()
                                      if len(l.split()) == 2 and l.split()[0].isdigit()])
    sd = model.state_dict()
    for n in names:
        out["q/" + n] = sd[n + ".weight"].numpy()
        out["scale/" + n] = quantizers[n].scale.numpy()
        out["zero/" + n] = quantizers[n].zero.numpy()
    out["ppl_gptq4"] = np.float64(ppl_of(model, test, run_args()))
    np.savez_compressed(OUT, **out)
    print("errors", errors)
    print("ppl fp / rtn4 / gptq4:", out["ppl_fp"], out["ppl_rtn4"], out["ppl_gptq4"])
    print(os.path.getsize(OUT) / 1024, "KiB")


if __name__ == "__main__":
    main()
