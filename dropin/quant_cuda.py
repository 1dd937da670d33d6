"""Drop-in shim: put this directory on PYTHONPATH and the reference drivers'
`from quant_cuda import *` / `import quant_cuda` resolve to the MI355X implementation."""
from gptq_amd.quant_cuda import *  # noqa: F401,F403
from gptq_amd.quant_cuda import vecquant3matmul, vecquant3matmul_faster, vecquant4matmul  # noqa: F401,E402
