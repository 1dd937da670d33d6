"""gptq_amd -- MI355X-native GPTQ hot path (Hessian, damped inverse factor, column loop, pack, dequant mat-vec).

Importing the package does not touch the GPU or load libgptq_hip.so; the first kernel
call does, and fails loudly if the library was not built (`python -m gptq_amd.build`).
"""
from .gptq import GPTQ, fasterquant_many, flush_pending  # noqa: F401
from .modelutils import DEV, find_layers  # noqa: F401
from .quant import (Quant3Linear, Quant4Linear, QuantGroupLinear, Quantizer, make_quant3,  # noqa: F401
                    make_quant4, pack_codes, quantize)

__version__ = "0.1.0"
