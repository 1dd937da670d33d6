"""The Hessian kernels order their inline-asm LDS reads by hand; check on the generated gfx950 ISA that the
compiler placed nothing that touches a read's destination before the wait covering it (tools/check_async_lds.py)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_use_of_async_lds_reads_before_their_wait():
    import check_async_lds as chk
    path = chk.compile_hessian()
    kernels = chk.parse_kernels(path)
    seen = 0
    for name, ins in kernels.items():
        n, bad = chk.check_kernel(name, ins)
        seen += n
        assert not bad, f"{name}: " + "; ".join(f"`{ins[i][1]}` <- `{ins[pc][1]}` ({why})" for i, pc, why in bad[:5])
    assert seen >= 60, "the inline-asm reads were not found: did the kernels or the parser change?"
