"""The C-ABI library loads without a GPU and exports every symbol include/gptq_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gptq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gptq_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from gptq_amd import _lib
    from gptq_amd.build import build_library
    build_library(verbose=False)
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gptq_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "ctypes binding table and header disagree"
    assert lib.gptq_hip_abi_version() == 2


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    import gptq_amd
    from gptq_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lin = torch.nn.Linear(8, 4)
    with pytest.raises(_lib.GptqHipError):
        gptq_amd.GPTQ(lin)
    q = gptq_amd.Quantizer(); q.configure(4, perchannel=True, sym=False)
    with pytest.raises(_lib.GptqHipError):
        q.find_params(torch.zeros(4, 8))
    with pytest.raises(_lib.GptqHipError):
        gptq_amd.quant_cuda.vecquant3matmul(torch.zeros(32), torch.zeros(3, 4, dtype=torch.int32), torch.zeros(4),
                                            torch.ones(4, 1), torch.zeros(4, 1))


def test_workspace_queries_need_no_gpu():
    from gptq_amd import _lib
    lib = _lib.load()
    assert lib.gptq_hinv_workspace_bytes(0) == 0
    assert lib.gptq_hinv_workspace_bytes(300) >= 2 * 384 * 384 * 4
    a = lib.gptq_fasterquant_workspace_bytes(64, 300, 128, -1, 0, 0)
    b = lib.gptq_fasterquant_workspace_bytes(64, 300, 128, -1, 1, 0)
    assert b - a >= 64 * 300 * 4


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gptq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle's", ""), f"{f} mentions the oracle"
