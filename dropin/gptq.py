"""Drop-in shim: put this directory on PYTHONPATH and the reference drivers'
`from gptq import *` / `import gptq` resolve to the MI355X implementation."""
from gptq_amd.gptq import *  # noqa: F401,F403
