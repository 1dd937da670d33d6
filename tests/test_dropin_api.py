"""The drop-in import boundary: `dropin/` imported exactly as the reference drivers do
(`from gptq import *; from modelutils import *; from quant import *`, opt.py:7-9, llama.py:8-10) must export the
names the drivers use, with the reference's signatures.  tests/golden/api_surface.json was recorded from the
reference itself by oracle/gen_api_surface.py (names, parameter lists, defaults -- data, no source)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r'''
import inspect, json, sys
from gptq import *
from modelutils import *
from quant import *
import quant_cuda
def sig(fn):
    return [{"name": p.name, "default": None if p.default is inspect.Parameter.empty else repr(p.default),
             "kind": p.kind.name} for p in inspect.signature(fn).parameters.values()]
out = {"gptq": {"GPTQ": {m: sig(getattr(GPTQ, m)) for m in ("__init__", "add_batch", "fasterquant", "free")}},
       "quant": {"quantize": sig(quantize),
                 "Quantizer": {m: sig(getattr(Quantizer, m)) for m in ("__init__", "configure", "find_params", "quantize", "enabled", "ready")},
                 "Quant3Linear": {m: sig(getattr(Quant3Linear, m)) for m in ("__init__", "pack", "forward")},
                 "make_quant3": sig(make_quant3)},
       "modelutils": {"find_layers": sig(find_layers), "DEV": str(DEV)},
       "quant_cuda": {n: len(inspect.signature(getattr(quant_cuda, n)).parameters) for n in ("vecquant3matmul", "vecquant3matmul_faster")},
       "star_names": sorted(n for n in dir() if not n.startswith("_"))}
print("APIJSON" + json.dumps(out))
'''


def compatible(ours, ref, where):
    """Same parameters in the same order with the same defaults; extra trailing parameters need defaults."""
    assert len(ours) >= len(ref), f"{where}: fewer parameters than the reference"
    for a, b in zip(ours, ref):
        assert a["name"] == b["name"], f"{where}: parameter {a['name']!r} where the reference has {b['name']!r}"
        assert a["default"] == b["default"], f"{where}: default of {a['name']} is {a['default']}, reference {b['default']}"
    for extra in ours[len(ref):]:
        assert extra["default"] is not None or extra["kind"].startswith("VAR"), f"{where}: extra parameter {extra['name']} without default"


def test_dropin_modules_export_the_reference_surface():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "api_surface.json")))["modules"]
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "dropin") + os.pathsep + ROOT)
    r = subprocess.run([sys.executable, "-c", PROBE], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ours = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("APIJSON"))[7:])
    for m in ("__init__", "add_batch", "fasterquant", "free"):
        compatible(ours["gptq"]["GPTQ"][m], ref["gptq"]["GPTQ"][m], f"GPTQ.{m}")
    for name in ref["gptq"]["reexports"]:                       # gptq.py re-exports quant.* (`from quant import *`)
        assert name in ours["star_names"], f"`from gptq import *` misses {name}"
    compatible(ours["quant"]["quantize"], ref["quant"]["quantize"], "quantize")
    compatible(ours["quant"]["make_quant3"], ref["quant"]["make_quant3"], "make_quant3")
    for cls in ("Quantizer", "Quant3Linear"):
        for m, s in ref["quant"][cls].items():
            compatible(ours["quant"][cls][m], s, f"{cls}.{m}")
    compatible(ours["modelutils"]["find_layers"], ref["modelutils"]["find_layers"], "find_layers")
    assert ours["modelutils"]["DEV"] == ref["modelutils"]["DEV"]
    for n, arity in ref["quant_cuda"].items():
        assert ours["quant_cuda"][n] == arity, f"quant_cuda.{n} takes {ours['quant_cuda'][n]} arguments, reference {arity}"
    for n in ("GPTQ", "Quantizer", "quantize", "Quant3Linear", "make_quant3", "find_layers", "DEV"):
        assert n in ours["star_names"]
