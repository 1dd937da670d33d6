#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Run in the build container only (needs the reference checkout, default
/root/reference):   MPLBACKEND=Agg python oracle/gen_golden.py

The reference modules are imported by absolute path and never copied; only the
small input/output vectors written here travel with the repo.  Every fixture
records the inputs next to the outputs so tests need nothing but the .npz.

Harness-side adjustments (do not change the reference's arithmetic):
  * torch.cuda.synchronize is stubbed (gptq.py:292 calls it; there is no GPU here);
  * torch.linalg.cholesky is wrapped to record the upper factor Hinv (a local
    variable in gptq.py:179-180);
  * stdout is captured to read the printed `error` (gptq.py:294).
"""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
REF = os.environ.get("GPTQ_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

torch.cuda.synchronize = lambda *a, **k: None
with contextlib.redirect_stdout(io.StringIO()):
    import gptq as ref_gptq
    import quant as ref_quant

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def f16_weights(gen, R, C, std=0.02):
    """fp16-representable weights held in an fp32 Linear, so the reference's final
    cast (gptq.py:305) is the identity and Q comes back in full fp32."""
    w = (torch.randn(R, C, generator=gen) * std).half().float()
    lin = nn.Linear(C, R, bias=True)
    lin.weight.data = w.clone()
    lin.bias.data = torch.randn(R, generator=gen).half().float()
    return lin, w


def calib(gen, S, C):
    chan = 1.0 + (torch.arange(C) % 7).float()          # non-flat diag(H): act-order is non-trivial
    return (torch.randn(1, S, C, generator=gen) * chan).half()


# ---------------------------------------------------------------- G1 add_batch
def g1():
    gen = torch.Generator().manual_seed(101)
    C, S = 192, 80
    lin, _ = f16_weights(gen, 8, C)
    g = ref_gptq.GPTQ(lin)
    xs, hs, means = [], [], []
    for _ in range(3):
        x = calib(gen, S, C)
        g.add_batch(x, None)
        xs.append(x.numpy())
        hs.append(g.H.clone().numpy())
        means.append(g.input.clone().numpy())
    save("g1_add_batch", X=np.stack(xs), H_after=np.stack(hs), input_mean=np.stack(means),
         nsamples=np.int64(g.nsamples))


# ------------------------------------------------ G2 find_params / quantize
def g2():
    gen = torch.Generator().manual_seed(202)
    W = torch.randn(12, 40, generator=gen) * 0.05
    W[0] = 0                           # all-zero row -> [-1, 1]   (quant.py:65-67)
    W[1] = W[1].abs() + 0.01           # all-positive row: xmin clamps to 0 (quant.py:56-57)
    W[2] = -W[2].abs() - 0.01          # all-negative row
    W[3, :] = 0.03125                  # constant row
    W[4] *= 40                         # wide row
    out = {"W": W.numpy()}
    for bits in (2, 3, 4, 8):
        for sym in (False, True):
            q = ref_quant.Quantizer()
            q.configure(bits, perchannel=True, sym=sym, mse=False)
            q.find_params(W, weight=True)
            tag = f"b{bits}_{'sym' if sym else 'asym'}"
            out[tag + "_scale"] = q.scale.numpy()
            out[tag + "_zero"] = q.zero.numpy()
            out[tag + "_q"] = ref_quant.quantize(W, q.scale, q.zero, q.maxq).numpy()
    save("g2_find_params", **out)


# --------------------------------------------------------------- G3 fasterquant
def run_fasterquant(lin, H, n, bits, sym, **kw):
    g = ref_gptq.GPTQ(lin)
    g.H = H.clone()
    g.nsamples = n
    g.quantizer = ref_quant.Quantizer()
    g.quantizer.configure(bits, perchannel=True, sym=sym, mse=False)
    grabbed = {"grids": []}
    real_chol = torch.linalg.cholesky
    real_quantize = ref_gptq.quantize

    def quantize_spy(x, scale, zero, maxq):        # gptq.py:262-264: one call per column, in loop order
        grabbed["grids"].append((scale.clone().flatten(), zero.clone().flatten()))
        return real_quantize(x, scale, zero, maxq)

    def spy(a, *args, upper=False, **kwargs):
        r = real_chol(a, *args, upper=upper, **kwargs)
        if upper:
            grabbed["Hinv"] = r.clone()
        return r

    torch.linalg.cholesky = spy
    ref_gptq.quantize = quantize_spy
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            g.fasterquant(**kw)
    finally:
        torch.linalg.cholesky = real_chol
        ref_gptq.quantize = real_quantize
    err = None
    for line in buf.getvalue().splitlines():
        if line.startswith("error"):
            err = float(line.split()[1])
    Q = lin.weight.data.clone()
    # per-column grids in PROCESSING order -> original column order, and the integer codes
    cs = torch.stack([a for a, _ in grabbed["grids"]], 1)
    cz = torch.stack([b for _, b in grabbed["grids"]], 1)
    if kw.get("actorder"):
        dead = torch.diag(H) == 0
        Hd = H.clone(); Hd[dead, dead] = 1
        perm = torch.argsort(torch.diag(Hd), descending=True)      # gptq.py:166 on the same input
        inv = torch.argsort(perm)
        cs, cz = cs[:, inv], cz[:, inv]
    codes = torch.clamp(torch.round(Q / cs) + cz, 0, 2 ** bits - 1).to(torch.uint8)
    assert torch.equal(cs * (codes.float() - cz), Q)
    return dict(Q=Q.numpy(), scale=g.quantizer.scale.clone().numpy(),
                zero=g.quantizer.zero.clone().numpy(), error=np.float64(err),
                Hinv=grabbed["Hinv"].numpy(), col_scale=cs.numpy(), col_zero=cz.numpy(), codes=codes.numpy())


def hessian_for(gen, C, S, n, dead=()):
    lin0 = nn.Linear(C, 4)
    g = ref_gptq.GPTQ(lin0)
    for _ in range(n):
        x = calib(gen, S, C)
        if dead:
            x[..., list(dead)] = 0
        g.add_batch(x, None)
    return g.H.clone(), g.nsamples


def g3():
    cases = [
        # name,            R,   C,  bits, sym,  kwargs, dead
        ("plain_c256",     32, 256, 4, False, dict(groupsize=-1), ()),
        ("plain_c320",     32, 320, 4, False, dict(groupsize=-1), ()),            # tail block of 64
        ("plain_3bit",     96, 256, 3, False, dict(groupsize=-1), ()),
        ("plain_sym",      32, 256, 4, True,  dict(groupsize=-1), ()),
        ("plain_2bit",     32, 256, 2, False, dict(groupsize=-1), ()),
        ("dead_col",       32, 256, 4, False, dict(groupsize=-1), (5, 130)),
        ("dead_actorder",  32, 256, 4, False, dict(groupsize=-1, actorder=True), (5, 130, 131)),
        ("actorder",       96, 320, 4, False, dict(groupsize=-1, actorder=True), ()),
        ("g128_dyn",       32, 256, 4, False, dict(groupsize=128), ()),
        ("g64_dyn",        32, 256, 4, False, dict(groupsize=64), ()),            # stale-W quirk (gptq.py:255)
        ("g64_dyn_c320",   32, 320, 4, False, dict(groupsize=64), ()),
        ("g32_dyn_sym",    32, 256, 3, True,  dict(groupsize=32), ()),
        ("g256_dyn",       32, 512, 4, False, dict(groupsize=256), ()),           # group spans two blocks
        ("g128_static",    32, 256, 4, False, dict(groupsize=128, static_groups=True), ()),
        ("g64_static_act", 96, 320, 4, False, dict(groupsize=64, static_groups=True, actorder=True), ()),
        ("g128_dyn_act",   32, 256, 4, False, dict(groupsize=128, actorder=True), ()),
        ("static_nogroup", 32, 256, 4, False, dict(groupsize=-1, static_groups=True), ()),  # opt.py:584-587 default
        ("blocksize64",    32, 256, 4, False, dict(groupsize=-1, blocksize=64), ()),
        ("percdamp10",     32, 256, 4, False, dict(groupsize=-1, percdamp=0.1), ()),
    ]
    for k, (name, R, C, bits, sym, kw, dead) in enumerate(cases):
        gen = torch.Generator().manual_seed(3000 + k)
        lin, w = f16_weights(gen, R, C)
        H, n = hessian_for(gen, C, 2 * C, 3, dead)
        full = dict(blocksize=128, percdamp=0.01, groupsize=-1, actorder=False, static_groups=False)
        full.update(kw)
        res = run_fasterquant(lin, H, n, bits, sym, **full)
        save("g3_" + name, W=w.numpy(), H=H.numpy(), bits=np.int64(bits), sym=np.bool_(sym),
             blocksize=np.int64(full["blocksize"]), percdamp=np.float64(full["percdamp"]),
             groupsize=np.int64(full["groupsize"]), actorder=np.bool_(full["actorder"]),
             static_groups=np.bool_(full["static_groups"]), **res)

    # one mid-size case for code-mismatch statistics; inputs stored compactly (fp16 W, fp16 X)
    gen = torch.Generator().manual_seed(3999)
    R = C = 512
    lin, w = f16_weights(gen, R, C)
    lin0 = nn.Linear(C, 4)
    g = ref_gptq.GPTQ(lin0)
    xs = []
    for _ in range(2):
        x = calib(gen, 1024, C)
        g.add_batch(x, None)
        xs.append(x.numpy())
    res = run_fasterquant(lin, g.H.clone(), g.nsamples, 4, False, blocksize=128, percdamp=0.01,
                          groupsize=-1, actorder=False, static_groups=False)
    save("g3_mid512", W=w.half().numpy(), X=np.stack(xs), H=g.H.numpy(), codes=res["codes"],
         scale=res["scale"], zero=res["zero"], error=res["error"], Q=res["Q"].astype(np.float32),
         bits=np.int64(4), sym=np.bool_(False))


# ----------------------------------------------------------------- G4 packing
def g4():
    gen = torch.Generator().manual_seed(404)
    out = {}
    for bits, R, C in ((3, 64, 96), (4, 64, 96)):
        lin, w = f16_weights(gen, R, C, std=0.05)
        q = ref_quant.Quantizer()
        q.configure(bits, perchannel=True, sym=False, mse=False)
        q.find_params(lin.weight.data, weight=True)
        lin.weight.data = ref_quant.quantize(lin.weight.data, q.scale, q.zero, q.maxq).half().float()
        tag = f"b{bits}_"
        out[tag + "W"] = lin.weight.data.numpy()
        out[tag + "bias"] = lin.bias.data.numpy()
        out[tag + "scale"] = q.scale.numpy()
        out[tag + "zero"] = q.zero.numpy()
        if bits == 3:
            m = ref_quant.Quant3Linear(C, R)
            m.pack(lin, q.scale, q.zero)                      # quant.py:152-187
            out[tag + "qweight"] = m.qweight.numpy()
            out[tag + "zeros_buf"] = m.zeros.numpy()
            out[tag + "scales_buf"] = m.scales.numpy()
        else:
            import importlib.util
            spec = importlib.util.spec_from_file_location(
                "ref_zs_quant", os.path.join(REF, "zeroShot", "models", "quant.py"))
            zs = importlib.util.module_from_spec(spec)
            with contextlib.redirect_stdout(io.StringIO()):
                spec.loader.exec_module(zs)
            m = zs.Quant4Linear(lin, q.scale, q.zero)         # zeroShot/models/quant.py:172-185
            out[tag + "qweight"] = m.qweight.numpy()
            out[tag + "zeros_buf"] = m.zeros.numpy()
            out[tag + "scales_buf"] = m.scales.numpy()
    save("g4_pack", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    g1(); g2(); g3(); g4()
