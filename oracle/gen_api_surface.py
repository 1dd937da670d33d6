#!/usr/bin/env python3
"""Record the names and signatures the reference's drivers rely on (`from gptq import *; from modelutils import *;
from quant import *`, opt.py:7-9, llama.py:8-10) into tests/golden/api_surface.json, by importing the REFERENCE in the
build container.  Data only (names, parameter lists, defaults): tests/test_dropin_api.py compares `dropin/` to it.

    MPLBACKEND=Agg python oracle/gen_api_surface.py
"""
import contextlib
import inspect
import io
import json
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
REF = os.environ.get("GPTQ_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import torch  # noqa: E402

torch.cuda.synchronize = lambda *a, **k: None
with contextlib.redirect_stdout(io.StringIO()):
    import gptq as ref_gptq
    import modelutils as ref_modelutils
    import quant as ref_quant

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "api_surface.json")


def sig(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        d = None if p.default is inspect.Parameter.empty else repr(p.default)
        out.append({"name": p.name, "default": d, "kind": p.kind.name})
    return out


def main():
    api = {"modules": {}}
    # what `from X import *` hands the drivers: public callables / classes defined by the hot path
    api["modules"]["gptq"] = {
        "GPTQ": {m: sig(getattr(ref_gptq.GPTQ, m)) for m in ("__init__", "add_batch", "fasterquant", "free")},
        "reexports": sorted(n for n in ("quantize", "Quantizer", "Quant3Linear", "make_quant3") if hasattr(ref_gptq, n)),
    }
    api["modules"]["quant"] = {
        "quantize": sig(ref_quant.quantize),
        "Quantizer": {m: sig(getattr(ref_quant.Quantizer, m))
                      for m in ("__init__", "configure", "find_params", "quantize", "enabled", "ready")},
        "Quant3Linear": {m: sig(getattr(ref_quant.Quant3Linear, m)) for m in ("__init__", "pack", "forward")},
        "make_quant3": sig(ref_quant.make_quant3),
    }
    api["modules"]["modelutils"] = {"find_layers": sig(ref_modelutils.find_layers), "DEV": str(ref_modelutils.DEV)}
    # quant_cuda is a compiled extension in the reference (quant_cuda.cpp:51-54): names + positional arity from the source
    api["modules"]["quant_cuda"] = {"vecquant3matmul": 5, "vecquant3matmul_faster": 5}
    json.dump(api, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps(api, indent=1)[:1500])


if __name__ == "__main__":
    main()
