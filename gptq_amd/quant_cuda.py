"""Packed dequant mat-vec entry points.

The reference loads a torch extension module literally named `quant_cuda`
(quant.py:134-137, setup_cuda.py:5-7) exporting `vecquant3matmul` and
`vecquant3matmul_faster` (quant_cuda.cpp:51-54).  The name is API, so it is kept;
the implementation is the gfx950 kernel in csrc/matvec.hip behind the C ABI.
`vecquant4matmul` is what zeroShot/models/quant.py:193 calls but the reference never
implemented.

All functions accumulate IN PLACE into `mul` (which the caller pre-loads with the
bias, quant.py:192) and return None.
"""
from __future__ import annotations

import torch

from . import _lib


def _run(symbol, bits, vec, mat, mul, scales, zeros, want_vec_dtype):
    for name, t in (("vec", vec), ("mat", mat), ("mul", mul), ("scales", scales), ("zeros", zeros)):
        _lib.require_gpu(t, name)
    if vec.dtype != want_vec_dtype:
        raise TypeError(f"{symbol}: vec must be {want_vec_dtype}, got {vec.dtype}")
    if mat.dtype != torch.int32 or mat.dim() != 2 or not mat.is_contiguous():
        raise TypeError(f"{symbol}: mat must be a contiguous int32 [in/32*{bits}, out] tensor")
    if mul.dtype != torch.float32 or not mul.is_contiguous():
        raise TypeError(f"{symbol}: mul must be contiguous fp32")
    height, width = mat.shape
    if vec.numel() != height // bits * 32:
        raise ValueError(f"{symbol}: vec has {vec.numel()} elements, mat implies {height // bits * 32}")
    if mul.numel() != width or scales.numel() != width or zeros.numel() != width:
        raise ValueError(f"{symbol}: mul/scales/zeros must have {width} elements")
    v = vec.reshape(-1).contiguous()
    s = scales.reshape(-1).to(torch.float32).contiguous()
    z = zeros.reshape(-1).to(torch.float32).contiguous()
    dev = vec.device
    with torch.cuda.device(dev):
        _lib.call(symbol, _lib.ptr(v), _lib.dtype_code(v), _lib.ptr(mat), _lib.ptr(mul), _lib.ptr(s), _lib.ptr(z),
                  height, width, _lib.stream(dev))


def vecquant3matmul(vec, mat, mul, scales, zeros):
    """Single-token 3-bit mat-vec (quant_cuda.cpp:15-21).  fp32 operands run as they are.  The reference also
    dispatches fp64 (AT_DISPATCH_FLOATING_TYPES, quant_cuda_kernel.cu:47): fp64 `vec/mul/scales/zeros` are accepted
    too -- the kernel computes in fp32 (gfx950 has no fp64 matrix path worth a second kernel for a test-only dtype)
    and the sum is added into `mul` in fp64, so the result agrees with the reference's to fp32 rounding (~1e-7
    relative), not to fp64 rounding."""
    if vec.dtype == torch.float64:
        for name, t in (("mul", mul), ("scales", scales), ("zeros", zeros)):
            if t.dtype != torch.float64:
                raise TypeError(f"vecquant3matmul: {name} must be fp64 like vec (the reference dispatches on vec's type)")
        acc = torch.zeros(mul.shape, dtype=torch.float32, device=mul.device)
        _run("gptq_vecquant3matmul", 3, vec.float(), mat, acc, scales.float(), zeros.float(), torch.float32)
        mul += acc.double()
        return
    _run("gptq_vecquant3matmul", 3, vec, mat, mul, scales, zeros, torch.float32)


def vecquant3matmul_faster(vec, mat, mul, scales, zeros):
    """fp16-input form (quant_cuda.cpp:23-29); accumulates in fp32 here."""
    _run("gptq_vecquant3matmul", 3, vec, mat, mul, scales, zeros, torch.float16)


def vecquant4matmul(vec, mat, mul, scales, zeros):
    """4-bit analogue on the zeroShot/models/quant.py:185 layout; vec fp32 or fp16."""
    _run("gptq_vecquant4matmul", 4, vec, mat, mul, scales, zeros, vec.dtype if vec.dtype == torch.float16 else torch.float32)


def vecquant_matmul_grouped(vec, mat, mul, scales, zeros, bits, groupsize):
    """Grouped-grid mat-vec: scales / zeros are [in/groupsize, out], zeros = zero*scale."""
    for name, t in (("vec", vec), ("mat", mat), ("mul", mul), ("scales", scales), ("zeros", zeros)):
        _lib.require_gpu(t, name)
    if vec.dtype not in (torch.float32, torch.float16):
        raise TypeError("vecquant_matmul_grouped: vec must be fp32 or fp16")
    if mat.dtype != torch.int32 or mat.dim() != 2 or not mat.is_contiguous():
        raise TypeError("vecquant_matmul_grouped: mat must be a contiguous int32 matrix")
    height, width = mat.shape
    n_in = height // bits * 32
    groups = n_in // groupsize
    if vec.numel() != n_in or mul.numel() != width or tuple(scales.shape) != (groups, width) \
            or tuple(zeros.shape) != (groups, width):
        raise ValueError("vecquant_matmul_grouped: shape mismatch")
    if mul.dtype != torch.float32 or not mul.is_contiguous():
        raise TypeError("vecquant_matmul_grouped: mul must be contiguous fp32")
    v = vec.reshape(-1).contiguous()
    s = scales.to(torch.float32).contiguous()
    z = zeros.to(torch.float32).contiguous()
    dev = vec.device
    with torch.cuda.device(dev):
        _lib.call("gptq_vecquant_matmul_grouped", _lib.ptr(v), _lib.dtype_code(v), _lib.ptr(mat), _lib.ptr(mul),
                  _lib.ptr(s), _lib.ptr(z), height, width, int(bits), int(groupsize), _lib.stream(dev))
