#!/usr/bin/env python3
"""Factor form vs inverse form of the solve on the same inputs: `python tools/rform_compare.py RxC [...] [--actorder]
[--groupsize G] [--dynamic]` runs each shape once per form (child processes: the form is read from GPTQ_RFORM when the
library loads) and prints how many integer codes differ."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(out):
    import torch
    import gptq_amd, gptq_amd.gptq as gmod
    gmod.VERBOSE = False
    dev = torch.device("cuda:0")
    act = "--actorder" in sys.argv
    gs = int(sys.argv[sys.argv.index("--groupsize") + 1]) if "--groupsize" in sys.argv else -1
    dyn = "--dynamic" in sys.argv
    res = {}
    for a in sys.argv[1:]:
        if "x" not in a:
            continue
        R, C = (int(v) for v in a.split("x"))
        gen = torch.Generator(device=dev).manual_seed(C)
        X = torch.randn(2 * C, C, device=dev, generator=gen) * (1 + torch.arange(C, device=dev) % 7)
        H = (X.t() @ X) * (2.0 / X.shape[0])
        lin = torch.nn.Linear(C, R, bias=False, device=dev, dtype=torch.float16)
        lin.weight.data = (torch.randn(R, C, device=dev, generator=gen) * 0.02).half()
        g = gptq_amd.GPTQ(lin)
        g.quantizer = gptq_amd.Quantizer(); g.quantizer.configure(4, perchannel=True, sym=False, mse=False)
        g.H = H; g.nsamples = 2
        g.fasterquant(blocksize=128, percdamp=0.01, groupsize=gs, static_groups=gs > 0 and not dyn, actorder=act)
        res[a] = dict(codes=g.codes.cpu(), error=g.error, form=g.Hinv_form)
    torch.save(res, out)


if __name__ == "__main__":
    if os.environ.get("RFORM_CHILD"):
        child(os.environ["RFORM_CHILD"])
        sys.exit(0)
    import torch
    outs = {}
    with tempfile.TemporaryDirectory() as d:
        for form in ("1", "0"):
            path = os.path.join(d, f"f{form}.pt")
            subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], check=True,
                           env=dict(os.environ, GPTQ_RFORM=form, RFORM_CHILD=path))
            outs[form] = torch.load(path, weights_only=True)
    for k in outs["1"]:
        a, b = outs["1"][k], outs["0"][k]
        diff = int((a["codes"] != b["codes"]).sum())
        rows = int((a["codes"] != b["codes"]).any(1).sum())
        print(f"{k}: {a['form']} vs {b['form']}: {diff} of {a['codes'].numel()} codes differ ({rows} rows); "
              f"error {a['error']:.6g} vs {b['error']:.6g}", flush=True)
