#!/usr/bin/env python3
"""More reference-made fixtures for the solve, at widths that are NOT a whole number of 512-column super-blocks and
with several 128-blocks per row of the factor: what the factor form of gptq_fasterquant (include/gptq_hip.h:
gptq_rfactor_upper) adds to the code paths -- a partial last super-block, far updates over more than one super-block,
dynamic groups inside a block next to act-order.  Same recipe as oracle/gen_golden.py (the REFERENCE's own
GPTQ.add_batch / fasterquant run on CPU in this container, outputs stored next to the inputs):

    python oracle/gen_golden_wide.py        # writes tests/golden/g3_w*.npz
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import gen_golden as gg

CASES = [
    # name,             R,   C,  bits, sym,  kwargs          (640 columns = one whole super-block of 512 + one block)
    ("w640_actorder",   48,  640, 4, False, dict(groupsize=-1, actorder=True)),
    ("w640_g64_dyn",    48,  640, 4, False, dict(groupsize=64)),
]

if __name__ == "__main__":
    torch.set_num_threads(8)
    for k, (name, R, C, bits, sym, kw) in enumerate(CASES):
        gen = torch.Generator().manual_seed(7000 + k)
        lin, w = gg.f16_weights(gen, R, C)
        H, n = gg.hessian_for(gen, C, 2 * C, 3, ())
        full = dict(blocksize=128, percdamp=0.01, groupsize=-1, actorder=False, static_groups=False)
        full.update(kw)
        res = gg.run_fasterquant(lin, H, n, bits, sym, **full)
        gg.save("g3_" + name, W=w.numpy(), H=H.numpy(), bits=np.int64(bits), sym=np.bool_(sym),
                blocksize=np.int64(full["blocksize"]), percdamp=np.float64(full["percdamp"]),
                groupsize=np.int64(full["groupsize"]), actorder=np.bool_(full["actorder"]),
                static_groups=np.bool_(full["static_groups"]), **res)
