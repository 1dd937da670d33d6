import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gptq_amd import _lib
lib = _lib.load()
C = 8192
gen = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(2 * C, C, device="cuda", generator=gen)
H0 = (X.t() @ X) * (1.0 / C)
nb = lib.gptq_hinv_workspace_bytes(C)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
info = torch.zeros(1, dtype=torch.int32, device="cuda")
for rep in range(3):
    H = H0.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.call("gptq_hinv_upper", _lib.ptr(H), H.stride(0), C, 0.01, None, _lib.ptr(info), _lib.ptr(ws), nb, _lib.stream(H.device))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"hinv_upper C={C}: {dt*1e3:.2f} ms")
