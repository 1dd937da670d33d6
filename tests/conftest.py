import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # the product never builds at import time (gptq_amd/_lib.py); a test session on a clean checkout builds here, before
    # any test has touched the GPU
    try:
        from gptq_amd.build import ensure_built
        ensure_built()
    except Exception as e:   # no hipcc / profiler: the tests that need the library fail loudly themselves
        print(f"conftest: library not built ({e})")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


# ---- tie-aware flip analysis (see the comment above test_fasterquant_mid1024_reference_flag_sets) ----
TIE_EPS_CAP = 1e-4


def hessian_fp64(X):
    """gptq.py:59-65 in fp64 from the stored samples X [k, 1, S, C]."""
    import torch
    C = X.shape[-1]
    H, n = torch.zeros(C, C, dtype=torch.float64), 0
    for k in range(X.shape[0]):
        x = X[k].reshape(-1, C).double()
        H *= n / (n + 1)
        n += 1
        H += (2.0 / n) * (x.t() @ x)
    return H


def tie_analysis(O, W, H32, H64, ours, ref, bits, **kw):
    """ours / ref: integer codes [R, C] (original column order).  Returns a list of dicts, one per differing row:
    first differing column (in processing order), margin of the exact value from k + 0.5, eps of that column."""
    import torch
    r32 = O.fasterquant(W, H32, bits=bits, xtrace=True, **kw)
    r64 = O.fasterquant(W, H64, bits=bits, dtype=torch.float64, xtrace=True, **kw)
    C = W.shape[1]
    pos = torch.arange(C) if r64.perm is None else torch.argsort(r64.perm)
    x32, x64 = r32.xtrace.double(), r64.xtrace
    clean = (r32.codes == r64.codes).all(1)          # rows on which fp32 and fp64 arithmetic agree throughout: noise only
    out = []
    for r in torch.nonzero((ours != ref).any(1)).flatten().tolist():
        cols = torch.nonzero(ours[r] != ref[r]).flatten()
        c = int(cols[torch.argmin(pos[cols])])
        x = float(x64[r, c])
        margin = abs(x - np.floor(x) - 0.5)
        eps = min(TIE_EPS_CAP, 2.0 * float((x32[clean, c] - x64[clean, c]).abs().max()))
        out.append(dict(row=r, col=c, flips=int(len(cols)), x_fp64=x, margin=margin, eps=eps,
                        step=abs(int(ours[r, c]) - int(ref[r, c]))))
    return out


def assert_flips_are_ties(ties, max_rows):
    for t in ties:
        assert t["step"] == 1, t                     # a tie moves the first code by exactly one step
        assert t["margin"] <= t["eps"], f"flipped code is not a rounding tie: {t}"
    assert len(ties) <= max_rows, ties


# ---- parity record: what the GPU tests OBSERVED (per fixture: rel-Fro, flipped codes, rows of the packed buffer that are
# bit-identical to the reference's), written to gpurun_out/parity.json at the end of every `-m gpu` session; the copy
# under profiles/rNN_parity.json is the tracked one.
PARITY = {}


def record_parity(name, **fields):
    PARITY[name] = fields


def parity_summary():
    """One line with the observed parity numbers, for the terminal summary (so that a driver that keeps only the tail
    of the pytest output still records them)."""
    if not PARITY:
        return None
    small = {k: v for k, v in PARITY.items() if k.startswith("g3_") and "flipped_codes" in v}
    parts = []
    if small:
        zero = sum(1 for v in small.values() if v["flipped_codes"] == 0)
        parts.append(f"{zero}/{len(small)} small reference fixtures 0 flipped codes"
                     + ("" if zero == len(small) else f" (max {max(v['flipped_codes'] for v in small.values())})"))
    mid = [(k, PARITY[k]) for k in sorted(PARITY) if k.startswith("g5_mid1024")]
    if mid:
        parts.append("mid1024 flips " + "/".join(str(v["flipped_codes"]) for _, v in mid) + " tie rows "
                     + "/".join(str(len(v.get("tie_rows") or [])) for _, v in mid)
                     + " non-tie relFro " + "/".join(f"{v.get('relfro_Q_non_tie_rows', float('nan')):.1e}" for _, v in mid))
    if "rows70016_actorder" in PARITY:
        v = PARITY["rows70016_actorder"]
        parts.append(f"70016-row flips {v['flipped_codes']} ({len(v.get('tie_rows') or [])} tie rows)")
    for k in sorted(PARITY):
        if k.startswith("full_matvec"):
            v = PARITY[k]
            parts.append(f"{k[5:]} {v['max_rel_fp32_x']:.1e}/{v['max_rel_fp16_x']:.1e}")
    if "ppl_proxy_opt125m_arch" in PARITY:
        v = PARITY["ppl_proxy_opt125m_arch"]
        parts.append(f"ppl fp/rtn4/gptq4 {v['ppl_fp']:.4f}/{v['ppl_rtn4']:.4f}/{v['ppl_gptq4']:.4f}")
    return "parity: " + "; ".join(parts) if parts else None


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    line = parity_summary()
    if line:
        terminalreporter.write_line(line)


def pytest_sessionfinish(session, exitstatus):
    if not PARITY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity.json"), "w") as f:
            json.dump({"exitstatus": int(exitstatus), "fixtures": PARITY}, f, indent=1, sort_keys=True)
    except OSError:
        pass
