#!/usr/bin/env python3
"""Correctness + speed of the Hessian kernels through the C ABI (gptq_hessian_accum_group).
    GPTQ_HESS_BIG=0|1|2 python3 tools/hessian_big_check.py [--dump path] [--no-time]
Writes H (C = 1024, 2 problems) to `--dump` so two builds / settings can be compared bit for bit."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gptq_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()


CU_LIMIT = int(os.environ.get("GPTQ_CHECK_CU_LIMIT", "0"))   # size every launch for that many compute units


def accum(Hs, Xs, nb, batch):
    """Hs: list of [C,C] fp32; Xs: list (per problem) of lists of [tokens, C] slabs."""
    if CU_LIMIT:                       # the CU budget is an argument of the mixed-width entry point
        return accum_mixed(Hs, Xs, nb, batch)
    n_x = len(Xs[0])
    C = Hs[0].shape[0]
    Hp = (ctypes.c_void_p * len(Hs))(*[h.data_ptr() for h in Hs])
    Xp = (ctypes.c_void_p * (len(Hs) * n_x))(*[x.data_ptr() for xs in Xs for x in xs])
    nbp = (ctypes.c_int * len(Hs))(*nb)
    x0 = Xs[0][0]
    _lib.call("gptq_hessian_accum_group", len(Hs), Hp, Hs[0].stride(0), Xp, n_x, _lib._DTYPES[x0.dtype],
              x0.stride(0), C, x0.shape[0], nbp, batch, _lib.stream(dev))


def check(C, tokens, n_x, n_prob, dtype):
    g = torch.Generator(device=dev).manual_seed(C + tokens)
    Xs = [[(torch.randn(tokens, C, device=dev, generator=g) * (1 + torch.arange(C, device=dev) % 7)).to(dtype)
           for _ in range(n_x)] for _ in range(n_prob)]
    Hs = [torch.randn(C, C, device=dev, generator=g) for _ in range(n_prob)]
    H0 = [h.clone() for h in Hs]
    nb = [3 + p for p in range(n_prob)]
    accum(Hs, Xs, nb, n_x)
    torch.cuda.synchronize()
    worst = 0.0
    for p in range(n_prob):
        X = torch.cat(Xs[p], 0).double()
        n_after = nb[p] + n_x
        ref = H0[p].double() * (nb[p] / n_after) + (2.0 / n_after) * (X.t() @ X)
        up = torch.triu(torch.ones(C, C, device=dev, dtype=torch.bool))
        err = ((Hs[p].double() - ref)[up].norm() / ref[up].norm()).item()
        low_untouched = torch.equal(Hs[p][~up], H0[p][~up])
        worst = max(worst, err)
        assert low_untouched, "strict lower triangle was written"
    print(f"check C={C} tokens={tokens} slabs={n_x} problems={n_prob} {dtype}: rel err {worst:.2e}", flush=True)
    assert worst < 2e-6
    return Hs


def accum_mixed(Hs, Xs, nb, batch):
    n_x = len(Xs[0])
    n = len(Hs)
    Hp = (ctypes.c_void_p * n)(*[h.data_ptr() for h in Hs])
    Xp = (ctypes.c_void_p * (n * n_x))(*[x.data_ptr() for xs in Xs for x in xs])
    nbp = (ctypes.c_int * n)(*nb)
    ldh = (ctypes.c_int * n)(*[h.stride(0) for h in Hs])
    ldx = (ctypes.c_int * n)(*[xs[0].stride(0) for xs in Xs])
    Cs = (ctypes.c_int * n)(*[h.shape[0] for h in Hs])
    x0 = Xs[0][0]
    _lib.call("gptq_hessian_accum_mixed", n, Hp, ldh, Xp, n_x, _lib._DTYPES[x0.dtype], ldx, Cs, x0.shape[0], nbp, batch,
              CU_LIMIT, _lib.stream(dev))


def check_mixed(widths, tokens, n_x, dtype):
    g = torch.Generator(device=dev).manual_seed(sum(widths) + tokens)
    Xs = [[(torch.randn(tokens, C, device=dev, generator=g) * (1 + torch.arange(C, device=dev) % 7)).to(dtype)
           for _ in range(n_x)] for C in widths]
    Hs = [torch.randn(C, C, device=dev, generator=g) for C in widths]
    H0 = [h.clone() for h in Hs]
    nb = [2 + p for p in range(len(widths))]
    accum_mixed(Hs, Xs, nb, n_x)
    torch.cuda.synchronize()
    worst = 0.0
    for p, C in enumerate(widths):
        X = torch.cat(Xs[p], 0).double()
        n_after = nb[p] + n_x
        ref = H0[p].double() * (nb[p] / n_after) + (2.0 / n_after) * (X.t() @ X)
        up = torch.triu(torch.ones(C, C, device=dev, dtype=torch.bool))
        worst = max(worst, ((Hs[p].double() - ref)[up].norm() / ref[up].norm()).item())
        assert torch.equal(Hs[p][~up], H0[p][~up]), "strict lower triangle was written"
    print(f"check mixed widths={widths} tokens={tokens} slabs={n_x} {dtype}: rel err {worst:.2e}", flush=True)
    assert worst < 2e-6


def timeit_mixed(widths, n_x=8, tokens=2048, reps=5):
    Xs = [[torch.randn(tokens, C, device=dev, dtype=torch.float16) for _ in range(n_x)] for C in widths]
    Hs = [torch.zeros(C, C, device=dev) for C in widths]
    accum_mixed(Hs, Xs, [0] * len(widths), n_x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        accum_mixed(Hs, Xs, [n_x * (r + 1)] * len(widths), n_x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"time mixed widths={widths} slabs={n_x}x{tokens}: {ms * 1e3:.1f} us/launch, "
          f"{n_x * tokens * sum(c * c for c in widths) / ms / 1e9:.1f} TFLOP/s algorithmic", flush=True)


def timeit(C, n_prob, n_x=8, tokens=2048, reps=5):
    Xs = [[torch.randn(tokens, C, device=dev, dtype=torch.float16) for _ in range(n_x)] for _ in range(n_prob)]
    Hs = [torch.zeros(C, C, device=dev) for _ in range(n_prob)]
    accum(Hs, Xs, [0] * n_prob, n_x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        accum(Hs, Xs, [n_x * (r + 1)] * n_prob, n_x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"time C={C} problems={n_prob} slabs={n_x}x{tokens}: {ms * 1e3:.1f} us/launch, "
          f"{n_prob * n_x * tokens * C * C / ms / 1e9:.1f} TFLOP/s algorithmic", flush=True)


args = sys.argv[1:]
print("GPTQ_HESS_BIG =", os.environ.get("GPTQ_HESS_BIG", "(default)"), "RING =", os.environ.get("GPTQ_HESS_RING", "-"),
      "ABLATE =", os.environ.get("GPTQ_HESS_ABLATE", "-"), flush=True)
if "--time-only" in args:          # --time-only C:problems [C:problems ...]
    for spec in args[args.index("--time-only") + 1:]:
        c, n = spec.split(":")
        timeit(int(c), int(n))
    sys.exit(0)
Hs = check(1024, 96, 3, 2, torch.float16)
if "--dump" in args:
    torch.save([h.cpu() for h in Hs], args[args.index("--dump") + 1])
check(4352, 64, 2, 1, torch.float16)       # 153 tiles of 256: the default heuristic takes the big kernel
check(3072, 160, 1, 2, torch.bfloat16)
check(512, 32, 5, 3, torch.float16)
check_mixed([2304, 1024, 512, 320, 768], 96, 2, torch.float16)      # 320 takes the 128x128 / ragged kernels
check_mixed([4352, 256, 256], 64, 3, torch.bfloat16)
if "--no-time" not in args:
    timeit_mixed([8192, 2048, 2048, 2048])
    timeit(8192, 1)
    timeit(2048, 5)
    timeit(4096, 1)
    timeit(5120, 2)
