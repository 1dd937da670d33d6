"""Drop-in shim: put this directory on PYTHONPATH and the reference drivers'
`from modelutils import *` / `import modelutils` resolve to the MI355X implementation."""
from gptq_amd.modelutils import *  # noqa: F401,F403
