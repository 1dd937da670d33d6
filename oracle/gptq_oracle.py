"""CPU oracle for the GPTQ hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``gptq_amd``) never does; it fails
loudly when the HIP library is missing.

This is a restatement (torch CPU fp32 for the floating-point steps, numpy for
the integer/bit steps) of the reference algorithm.  Every function cites the
reference lines it follows (paths relative to the reference checkout).

Parity pin: ``tests/golden/*.npz`` were produced by ``oracle/gen_golden.py``
running the *reference itself* (imported from its checkout, CPU) in the build
container; ``tests/test_oracle_golden.py`` checks this module against them
(bit-exact for H, Q, scale, zero, packed buffers).  The reference has no tests
or golden vectors of its own for this path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

__all__ = [
    "quantize", "find_params", "hessian_add_batch", "hinv_upper", "fasterquant",
    "pack3", "unpack3", "pack4", "unpack4", "intweight", "dequant_matvec",
    "FasterquantResult",
]


# --------------------------------------------------------------------------
# affine grid  (quant.py:6-10, 37-77)
# --------------------------------------------------------------------------
def quantize(x: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor, maxq: int) -> torch.Tensor:
    """quant.py:6-10 -- round-to-nearest-even onto the affine grid, dequantized."""
    if maxq < 0:  # ternary ("trits") branch, quant.py:7-8
        return (x > scale / 2).float() * scale + (x < zero / 2).float() * zero
    codes = torch.clamp(torch.round(x / scale) + zero, 0, maxq)
    return scale * (codes - zero)


def find_params(x: torch.Tensor, maxq: int, sym: bool):
    """quant.py:37-77,105-109 with perchannel=True, weight=True, mse=False.

    x: [R, n] fp32.  Returns (scale[R,1], zero[R,1]) fp32.
    """
    x = x.flatten(1)
    lo = torch.minimum(x.min(1)[0], torch.zeros(x.shape[0]))      # quant.py:56-57
    hi = torch.maximum(x.max(1)[0], torch.zeros(x.shape[0]))      # quant.py:58
    if sym:                                                       # quant.py:60-64
        hi = torch.maximum(lo.abs(), hi)
        neg = lo < 0
        lo = torch.where(neg, -hi, lo)
    flat = (lo == 0) & (hi == 0)                                  # quant.py:65-67
    lo = torch.where(flat, torch.full_like(lo, -1.0), lo)
    hi = torch.where(flat, torch.full_like(hi, +1.0), hi)
    if maxq < 0:                                                  # quant.py:69-71
        scale, zero = hi, lo
    else:
        scale = (hi - lo) / maxq                                  # quant.py:73
        if sym:
            zero = torch.full_like(scale, (maxq + 1) / 2)         # quant.py:75
        else:
            zero = torch.round(-lo / scale)                       # quant.py:77
    return scale.reshape(-1, 1), zero.reshape(-1, 1)


# --------------------------------------------------------------------------
# Hessian running mean  (gptq.py:38-65)
# --------------------------------------------------------------------------
def hessian_add_batch(H: torch.Tensor, nsamples: int, inp: torch.Tensor):
    """gptq.py:42-65 for nn.Linear.  H [C,C] fp32 is updated in place.

    inp: [S, C] or [B, S, C] (any float dtype).  Returns the new sample count.
    """
    if inp.dim() == 2:
        inp = inp.unsqueeze(0)
    batch = inp.shape[0]                                          # gptq.py:44
    xt = inp.reshape(-1, inp.shape[-1]).t()                       # gptq.py:46-48  [C, B*S]
    H *= nsamples / (nsamples + batch)                            # gptq.py:59
    nsamples += batch
    xt = math.sqrt(2 / nsamples) * xt.float()                     # gptq.py:62 (cast, then scale)
    H += xt.matmul(xt.t())                                        # gptq.py:65
    return nsamples


# --------------------------------------------------------------------------
# damped inverse factor  (gptq.py:141-145, 165-180)
# --------------------------------------------------------------------------
def hinv_upper(H: torch.Tensor, percdamp: float, perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Upper factor U with U^T U = (H + damp I)^-1 -- gptq.py:174-180.

    H must already have the dead-column fix applied (gptq.py:143-144).
    """
    H = H.clone()
    if perm is not None:
        H = H[perm][:, perm]                                      # gptq.py:168
    n = H.shape[0]
    damp = percdamp * torch.mean(torch.diag(H))                   # gptq.py:174
    idx = torch.arange(n)
    H[idx, idx] += damp                                           # gptq.py:175-176
    L = torch.linalg.cholesky(H)                                  # gptq.py:177
    Hi = torch.cholesky_inverse(L)                                # gptq.py:178
    return torch.linalg.cholesky(Hi, upper=True)                  # gptq.py:179


@dataclass
class FasterquantResult:
    Q: torch.Tensor                 # [R,C] fp32 dequantized weights, original column order (gptq.py:300-301)
    scale: torch.Tensor             # [R,1] grid left in the quantizer after the call
    zero: torch.Tensor              # [R,1]
    error: float                    # sum(Losses)  (gptq.py:294)
    perm: Optional[torch.Tensor]    # act-order permutation (gptq.py:166) or None
    Hinv: torch.Tensor              # upper factor actually used (permuted order)
    col_scale: torch.Tensor         # [R,C] grid used for every column (original order)
    col_zero: torch.Tensor          # [R,C]
    codes: torch.Tensor             # [R,C] int32 integer codes, original order
    group_scale: Optional[torch.Tensor] = None  # [R,G] static-group table (gptq.py:157-163)
    group_zero: Optional[torch.Tensor] = None
    W_after: torch.Tensor = field(default=None)  # compensated working weights at exit (permuted order)
    xtrace: Optional[torch.Tensor] = None       # [R,C] w / scale of every column right before it is rounded (original order)


def fasterquant(
    W: torch.Tensor, H: torch.Tensor, *, bits: int, sym: bool = False,
    blocksize: int = 128, percdamp: float = 0.01, groupsize: int = -1,
    actorder: bool = False, static_groups: bool = False,
    scale: Optional[torch.Tensor] = None, zero: Optional[torch.Tensor] = None,
    Hinv_override: Optional[torch.Tensor] = None, trace: Optional[list] = None,
    dtype: torch.dtype = torch.float32, xtrace: bool = False,
) -> FasterquantResult:
    """gptq.py:126-305, default (plain affine quantizer) branch.

    W [R,C] (any float dtype; cast to fp32, gptq.py:135), H [C,C] fp32 (consumed).
    ``scale``/``zero`` pre-set a "ready" quantizer (gptq.py:181).  ``Hinv_override``
    replaces the factorization chain's output (test hook for bit-exact loop parity);
    ``trace`` (a list) receives the working weights at every block start.

    Test hooks for the tie analysis of the parity tests (the default arguments are the reference's arithmetic,
    untouched): ``dtype=torch.float64`` runs the factorization chain and the column loop in fp64 on the SAME fp32
    grids (grids are always found in fp32, gptq.py:157-163 / quant.py:37-77, then widened); ``xtrace=True`` records
    ``w / scale`` of every column right before it is rounded (quant.py:9) -- its distance from k + 0.5 is the
    rounding margin of that weight.
    """
    maxq = 2 ** bits - 1                                          # quant.py:27
    W = W.clone().float().to(dtype)
    R, C = W.shape
    H = H.clone().to(dtype)
    _find = find_params
    if dtype != torch.float32:
        def _find(x, maxq, sym):                                  # the fp32 grid, widened
            s, z = find_params(x.float(), maxq, sym)
            return s.to(dtype), z.to(dtype)
    XT = torch.zeros_like(W) if xtrace else None

    dead = torch.diag(H) == 0                                     # gptq.py:143-145
    H[dead, dead] = 1
    W[:, dead] = 0

    use_static = static_groups and groupsize > 0
    g_scale = g_zero = None
    if static_groups:                                             # gptq.py:157-163 (range(0,C,-1) empty if groupsize=-1)
        g_scale, g_zero = [], []
        for c0 in range(0, C, groupsize):
            s, z = _find(W[:, c0:c0 + groupsize], maxq, sym)
            g_scale.append(s)
            g_zero.append(z)
        if g_scale:
            g_scale = torch.cat(g_scale, 1)
            g_zero = torch.cat(g_zero, 1)
        else:
            g_scale = g_zero = None

    perm = None
    if actorder:                                                  # gptq.py:165-169
        perm = torch.argsort(torch.diag(H), descending=True)
        W = W[:, perm]
        H = H[perm][:, perm]

    Losses = torch.zeros_like(W)
    Q = torch.zeros_like(W)
    CS = torch.zeros_like(W)
    CZ = torch.zeros_like(W)

    if Hinv_override is not None:
        Hinv = Hinv_override
    else:
        Hinv = hinv_upper(H, percdamp)                            # gptq.py:174-180

    ready = scale is not None and bool(torch.all(scale != 0))     # quant.py:130-131
    if not ready:                                                 # gptq.py:181-185
        scale, zero = _find(W, maxq, sym)

    for i1 in range(0, C, blocksize):                             # gptq.py:191
        i2 = min(i1 + blocksize, C)
        n = i2 - i1
        if trace is not None:       # test hook: the global working weights entering this block
            trace.append(W.clone())
        W1 = W[:, i1:i2].clone()                                  # gptq.py:195-199
        Q1 = torch.zeros_like(W1)
        E1 = torch.zeros_like(W1)
        L1 = torch.zeros_like(W1)
        U1 = Hinv[i1:i2, i1:i2]
        for i in range(n):                                        # gptq.py:201
            w = W1[:, i]
            d = U1[i, i]
            if groupsize != -1:                                   # gptq.py:252-260
                if not static_groups:
                    if (i1 + i) % groupsize == 0:                 # reads the GLOBAL W (stale inside a block)
                        scale, zero = _find(W[:, (i1 + i):(i1 + i + groupsize)], maxq, sym)
                else:
                    col = int(perm[i1 + i]) if actorder else i1 + i
                    scale = g_scale[:, col // groupsize].reshape(-1, 1)
                    zero = g_zero[:, col // groupsize].reshape(-1, 1)
            if XT is not None:
                XT[:, i1 + i] = w / scale.flatten()
            q = quantize(w.unsqueeze(1), scale, zero, maxq).flatten()   # gptq.py:262-264
            Q1[:, i] = q
            L1[:, i] = (w - q) ** 2 / d ** 2                      # gptq.py:267
            e = (w - q) / d                                       # gptq.py:269
            W1[:, i:] -= e.unsqueeze(1).matmul(U1[i, i:].unsqueeze(0))  # gptq.py:270
            E1[:, i] = e
            CS[:, i1 + i] = scale.flatten()
            CZ[:, i1 + i] = zero.flatten()
        Q[:, i1:i2] = Q1                                          # gptq.py:273-274
        Losses[:, i1:i2] = L1 / 2
        W[:, i2:] -= E1.matmul(Hinv[i1:i2, i2:])                  # gptq.py:276

    error = torch.sum(Losses).item()                              # gptq.py:294
    codes = torch.clamp(torch.round(Q / CS) + CZ, 0, max(maxq, 0)).to(torch.int32)
    if actorder:                                                  # gptq.py:300-301
        inv = torch.argsort(perm)
        Q, CS, CZ, codes = Q[:, inv], CS[:, inv], CZ[:, inv], codes[:, inv]
        if XT is not None:
            XT = XT[:, inv]
    return FasterquantResult(Q=Q, scale=scale, zero=zero, error=error, perm=perm, Hinv=Hinv,
                             col_scale=CS, col_zero=CZ, codes=codes,
                             group_scale=g_scale if use_static else None,
                             group_zero=g_zero if use_static else None, W_after=W, xtrace=XT)


# --------------------------------------------------------------------------
# packing  (quant.py:152-187; zeroShot/models/quant.py:176-185)
# --------------------------------------------------------------------------
def intweight(weight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor) -> np.ndarray:
    """quant.py:153,158-160: integer codes [in, out] as uint32.

    ``zeros`` is the quantizer's integer zero point; the module stores zero*scale.
    """
    zs = zeros * scales                                           # quant.py:153
    iw = torch.round((weight + zs) / scales).to(torch.int)        # quant.py:158
    return iw.t().contiguous().numpy().astype(np.uint32)          # quant.py:159-160


def pack3(iw: np.ndarray) -> np.ndarray:
    """quant.py:161-186: 32 three-bit codes -> 3 little-endian int32 words along `in`.

    iw: [in, out] uint32.  Returns qweight [in//32*3, out] int32.  Written as the
    96-bit little-endian stream it is, with the reference's
    uint32 wrap-around semantics for out-of-range codes.
    """
    n_in, n_out = iw.shape
    out = np.zeros((n_in // 32 * 3, n_out), dtype=np.uint32)
    for g in range(n_in // 32):
        blk = iw[32 * g:32 * g + 32]
        w0, w1, w2 = out[3 * g], out[3 * g + 1], out[3 * g + 2]
        for j in range(10):
            w0 |= blk[j] << np.uint32(3 * j)                      # quant.py:167-168
        w0 |= blk[10] << np.uint32(30)                            # quant.py:170
        w1 |= (blk[10] >> np.uint32(2)) & np.uint32(1)            # quant.py:172
        for j in range(10):
            w1 |= blk[11 + j] << np.uint32(3 * j + 1)             # quant.py:174-175
        w1 |= blk[21] << np.uint32(31)                            # quant.py:177
        w2 |= (blk[21] >> np.uint32(1)) & np.uint32(3)            # quant.py:179
        for j in range(10):
            w2 |= blk[22 + j] << np.uint32(3 * j + 2)             # quant.py:181-182
    return out.astype(np.int32)                                   # quant.py:186


def unpack3(qweight: np.ndarray) -> np.ndarray:
    """Inverse of pack3 for in-range codes: [in//32*3, out] int32 -> [in, out] uint32.

    Code j of a 32-group sits at bit 3*j of the 96-bit little-endian stream
    w0 | w1<<32 | w2<<64 (quant_cuda_kernel.cu:107-160 decodes the same stream).
    """
    q = qweight.astype(np.uint32)
    groups = q.shape[0] // 3
    codes = np.zeros((groups * 32, q.shape[1]), dtype=np.uint32)
    for j in range(32):
        word, off = divmod(3 * j, 32)
        v = q[word::3][:groups] >> np.uint32(off)
        if off > 29:  # field straddles two words
            v = v | (q[word + 1::3][:groups] << np.uint32(32 - off))
        codes[j::32] = v & np.uint32(7)
    return codes


def pack4(iw: np.ndarray) -> np.ndarray:
    """zeroShot/models/quant.py:181-185: nibble i%8 of word i//8 along `in`."""
    n_in, n_out = iw.shape
    out = np.zeros((n_in // 8, n_out), dtype=np.uint32)
    for i in range(n_in // 8 * 8):
        out[i // 8] |= iw[i] << np.uint32(4 * (i % 8))
    return out.astype(np.int32)


def unpack4(qweight: np.ndarray) -> np.ndarray:
    q = qweight.astype(np.uint32)
    codes = np.zeros((q.shape[0] * 8, q.shape[1]), dtype=np.uint32)
    for j in range(8):
        codes[j::8] = (q >> np.uint32(4 * j)) & np.uint32(0xF)
    return codes


def dequant_matvec(vec: np.ndarray, qweight: np.ndarray, mul: np.ndarray,
                   scales: np.ndarray, zeros: np.ndarray, bits: int) -> np.ndarray:
    """quant_cuda_kernel.cu:88-165 (3-bit) in fp64: mul + sum_k (scale*q - zero) * vec[k].

    ``zeros`` is the stored zero*scale (quant.py:153).  The 4-bit form has no
    reference kernel (quant_cuda.cpp:51-54 exports only the 3-bit entry points);
    it uses the same dequant formula on the zeroShot/models/quant.py:185 layout
    -- parity unpinned by the reference for bits=4.
    """
    codes = (unpack3(qweight) if bits == 3 else unpack4(qweight)).astype(np.float64)   # [in, out]
    s = scales.reshape(-1).astype(np.float64)
    z = zeros.reshape(-1).astype(np.float64)
    wdeq = codes * s[None, :] - z[None, :]
    return mul.astype(np.float64) + vec.reshape(-1).astype(np.float64) @ wdeq
