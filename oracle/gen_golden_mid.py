#!/usr/bin/env python3
"""Mid-size goldens (SURVEY G3 "one mid-size 1024x1024 case for mismatch statistics", with the flag sets the BASELINE
configs actually use) made by running the REFERENCE itself on CPU in the build container:

    MPLBACKEND=Agg python oracle/gen_golden_mid.py

One shared input file (fp16 W [1024, 1024], fp16 calibration X [2, 1024, 1024]) and three compact result files --
integer codes, grids and the `error` scalar instead of the fp32 Q:
  g5_mid1024_g128_static  4-bit, groupsize 128, static groups   (BASELINE configs[1] flags; opt.py:585)
  g5_mid1024_actorder     4-bit, per-row grid, --act-order      (BASELINE configs[2] flags)
  g5_mid1024_3bit         3-bit, per-row grid                   (BASELINE configs[3] flags)
Q is exactly col_scale * (codes - col_zero) (asserted here), so nothing is lost.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as GG   # imports the reference, stubs torch.cuda.synchronize  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def main():
    torch.set_num_threads(8)
    gen = torch.Generator().manual_seed(5000)
    R = C = 1024
    lin0, w = GG.f16_weights(gen, R, C)
    g = GG.ref_gptq.GPTQ(nn.Linear(C, 4))
    xs = []
    for _ in range(2):
        x = GG.calib(gen, 1024, C)
        g.add_batch(x, None)
        xs.append(x.numpy())
    H, n = g.H.clone(), g.nsamples
    GG.save("g5_mid1024_inputs", W=w.half().numpy(), X=np.stack(xs))
    cases = [("g128_static", 4, dict(groupsize=128, static_groups=True)),
             ("actorder", 4, dict(groupsize=-1, actorder=True)),
             ("3bit", 3, dict(groupsize=-1))]
    for name, bits, kw in cases:
        lin = nn.Linear(C, R, bias=False)
        lin.weight.data = w.clone()
        full = dict(blocksize=128, percdamp=0.01, groupsize=-1, actorder=False, static_groups=False)
        full.update(kw)
        res = GG.run_fasterquant(lin, H, n, bits, False, **full)
        out = dict(codes=res["codes"], scale=res["scale"], zero=res["zero"], error=res["error"], bits=np.int64(bits),
                   groupsize=np.int64(full["groupsize"]), actorder=np.bool_(full["actorder"]),
                   static_groups=np.bool_(full["static_groups"]))
        gs = full["groupsize"]
        if gs > 0:      # one grid per group of ORIGINAL columns (static groups): keep the table, not [R, C]
            cs, cz = res["col_scale"], res["col_zero"]
            assert all(np.array_equal(cs[:, j * gs:(j + 1) * gs], np.repeat(cs[:, j * gs:j * gs + 1], gs, 1)) for j in range(C // gs))
            out["group_scale"], out["group_zero"] = cs[:, ::gs].copy(), cz[:, ::gs].copy()
        else:
            assert np.array_equal(res["col_scale"], np.repeat(res["scale"], C, 1))
        GG.save("g5_mid1024_" + name, **out)


if __name__ == "__main__":
    main()
