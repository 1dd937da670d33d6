"""Build libgptq_hip.so (gfx950 only) in-tree with hipcc.

`python -m gptq_amd.build` or `gptq_amd.build.build_library()`.  The library is a
plain C-ABI shared object (include/gptq_hip.h); nothing links against torch.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgptq_hip.so")
DIAG_LIB = os.path.join(HERE, "libgptq_hip_diag.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
SOURCES = ["core.cpp", "hessian.hip", "cholesky.hip", "fasterquant.hip", "quant_super.hip", "pack.hip", "matvec.hip"]
# -ffp-contract=off: the quantize / error-feedback chain must round exactly like the reference's
# separate torch ops (no implicit FMA); MFMA and explicit fmaf() are unaffected.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
# The factorization chain is tolerance-level by construction (its reductions already differ from
# LAPACK's order), so FMA contraction is allowed there.
PER_FILE_FLAGS = {"cholesky.hip": ["-ffp-contract=fast"],
                  # the SLP vectorizer pairs scalar fp32 operations of different columns into v_pk_* (slower beside a
                  # dependent chain, MI355X_MICROARCH.md) and ties their live ranges together: hundreds of spills
                  "quant_super.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True, diag: bool = False) -> str:
    """diag=True builds libgptq_hip_diag.so instead: the same sources with -DGPTQ_DIAG, which compiles in the
    timing-only ablation variants (GPTQ_*_ABLATE environment knobs; their results are WRONG by design).  The product
    library never contains them; tools load the diagnostic one explicitly through GPTQ_HIP_LIB."""
    hipcc = _hipcc()
    import fcntl
    os.makedirs(globals()["OBJ"], exist_ok=True)
    with open(os.path.join(globals()["OBJ"], ".lock"), "w") as lock:   # one builder at a time per checkout
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(hipcc, force, verbose, diag)


def _build_locked(hipcc: str, force: bool, verbose: bool, diag: bool) -> str:
    LIB = DIAG_LIB if diag else globals()["LIB"]
    OBJ = globals()["OBJ"] + ("_diag" if diag else "")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "gptq_hip.h"))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            lang = ["-x", "hip"] if src.endswith(".cpp") else []
            jobs.append([hipcc] + FLAGS + (["-DGPTQ_DIAG"] if diag else []) + PER_FILE_FLAGS.get(src, []) + lang
                        + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        tmp = f"{LIB}.tmp.{os.getpid()}"                     # never a half-written library under the final name
        try:
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB


def ensure_built() -> str:
    """For entry points that run BEFORE anything touches the GPU (test session start, bench.py's first lines,
    __graft_entry__): build the library if it is missing.  Refuses inside a profiler-preloaded process tree (a compiler
    launch there is an exec from a GPU-initialised process); concurrent callers (torchrun ranks) serialise on the build
    lock and the late ones find the library present."""
    lib = globals()["LIB"]
    if os.path.exists(lib):
        return lib
    if os.environ.get("LD_PRELOAD") or any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ):
        raise RuntimeError(f"{lib} is missing and this process runs under a profiler: build first with "
                           "`python -m gptq_amd.build`")
    return build_library(verbose=False)


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, diag="--diag" in sys.argv))
