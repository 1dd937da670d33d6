"""Drop-in shim: put this directory on PYTHONPATH and the reference drivers'
`from quant import *` / `import quant` resolve to the MI355X implementation."""
from gptq_amd.quant import *  # noqa: F401,F403
