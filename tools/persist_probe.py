#!/usr/bin/env python3
"""chol_panel_kernel (one launch per outer panel) against the launch-per-step chain: the factor form bit for bit, and
the time of `gptq_rfactor_upper` alone.  `python3 tools/persist_probe.py [C ...]`; each mode runs in a child process
(the mode is read once per process), under a timeout."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(C, out):
    import torch
    from gptq_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(C)
    X = torch.randn(2 * C, C, device="cuda", generator=gen) * (1 + torch.arange(C, device="cuda") % 7)
    H0 = (X.t() @ X) * (1.0 / C)
    nb = lib.gptq_hinv_workspace_bytes(C)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    best = 1e9
    stress = int(os.environ.get("PERSIST_STRESS", "0"))        # that many more runs under uneven load, every word compared
    first, diffs = None, 0
    if stress:                                                 # a second stream keeps part of the chip busy with GEMMs of
        side = torch.cuda.Stream()                             # changing size while the factorizations run
        A = torch.randn(4096, 4096, device="cuda")
    for rep in range(4 + stress):
        H = H0.clone()
        if stress and rep >= 4:
            with torch.cuda.stream(side):
                for _ in range(1 + rep % 3):
                    n = 512 * (1 + (rep * 7) % 8)
                    torch.mm(A[:n], A[:, :n])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.call("gptq_rfactor_upper", _lib.ptr(H), H.stride(0), C, 0.01, None, _lib.ptr(info), _lib.ptr(ws), nb,
                  _lib.stream(H.device))
        if not stress or rep < 4:
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        if first is None:
            first = H.clone()
        elif not torch.equal(first, H):
            diffs += 1
    torch.cuda.synchronize()
    if stress:
        print(f"C={C} persist={os.environ.get('GPTQ_CHOL_PERSIST', '1')}: {stress} runs under load, {diffs} differ from the first", flush=True)
        if diffs: sys.exit(3)
    print(f"C={C} persist={os.environ.get('GPTQ_CHOL_PERSIST', '1')} wgs={os.environ.get('GPTQ_CHOL_WGS', '-')}: "
          f"rfactor {best * 1e3:.3f} ms, info {int(info.item())}", flush=True)
    torch.save(H.cpu(), out)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    import torch
    Cs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [640, 1408, 4096]
    wgs = [a.split("=")[1] for a in sys.argv[1:] if a.startswith("--wgs=")]
    bad = 0
    for C in Cs:
        outs = []
        for mode, w in [("0", None)] + [("1", w) for w in (wgs or [None])]:
            env = dict(os.environ, GPTQ_CHOL_PERSIST=mode)
            if w: env["GPTQ_CHOL_WGS"] = w
            out = f"/tmp/persist_{C}_{mode}_{w}.pt"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(C), out], env=env, timeout=120)
            if r.returncode: sys.exit(f"child failed: C={C} persist={mode}")
            outs.append(torch.load(out))
        for o in outs[1:]:
            same = torch.equal(outs[0], o)
            print(f"C={C}: factor bit-identical to the launch-per-step chain: {same}"
                  + ("" if same else f" (max abs diff {(outs[0] - o).abs().max().item():.3e}, nan {int(torch.isnan(o).sum())})"), flush=True)
            bad += not same
            if not same and C <= 2048:                     # which 128-blocks of the output differ (U-space: block (i, j) is
                nb = C // 128                              #  the image of L-space block (nb-1-j, nb-1-i))
                d = (outs[0] != o) | torch.isnan(o)
                for i in range(nb):
                    print("   ", "".join("x" if d[128 * i:128 * i + 128, 128 * j:128 * j + 128].any() else "." for j in range(nb)), flush=True)
    sys.exit(1 if bad else 0)
