"""Affine weight quantizer and packed 3/4-bit Linear modules on MI355X.

Mirrors the reference's `quant.py` surface (`quantize`, `Quantizer`, `Quant3Linear`,
`make_quant3`) plus the int4 module whose layout the reference pins in
zeroShot/models/quant.py:172-209.  All arithmetic on the hot path runs in
libgptq_hip.so; torch only owns memory.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from . import quant_cuda  # noqa: F401  (the extension-module name is part of the reference API)


def _dev_of(t: torch.Tensor) -> torch.device:
    _lib.require_gpu(t, "tensor")
    return t.device


def quantize(x, scale, zero, maxq):
    """scale * (clamp(round(x / scale) + zero, 0, maxq) - zero)   (reference quant.py:6-10).

    Fast path: x [R, C] with per-row grids scale/zero [R, 1] -> one HIP pass.  Other
    broadcast shapes are elementwise torch on the tensors' own device.
    """
    maxq_i = int(maxq)
    if maxq_i < 0:
        raise NotImplementedError("trits (maxq < 0) are outside the MI355X hot-path scope")
    def per_row(t):   # [R, 1] -- or [R] against a single column -- broadcasts along rows exactly like quant.py:9-10
        return tuple(t.shape) == (x.shape[0], 1) or (t.dim() == 1 and t.numel() == x.shape[0] and x.shape[1] == 1)
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and per_row(scale) and per_row(zero)
            and ((maxq_i + 1) & maxq_i) == 0 and 0 < maxq_i <= 255):
        out = x.contiguous().clone()
        s = scale.reshape(-1).to(torch.float32).contiguous()
        z = zero.reshape(-1).to(torch.float32).contiguous()
        bits = (maxq_i + 1).bit_length() - 1
        with torch.cuda.device(x.device):
            _lib.call("gptq_quantize_rows", _lib.ptr(out), out.stride(0), out.shape[0], out.shape[1],
                      _lib.ptr(s), _lib.ptr(z), bits, _lib.stream(x.device))
        return out
    q = torch.clamp(torch.round(x / scale) + zero, 0, maxq_i)
    return scale * (q - zero)


class Quantizer(nn.Module):
    """Per-output-channel min/max affine grid (reference quant.py:12-131)."""

    def __init__(self, shape=1):
        super().__init__()
        self.register_buffer('maxq', torch.tensor(0))
        self.register_buffer('scale', torch.zeros(shape))
        self.register_buffer('zero', torch.zeros(shape))

    def configure(self, bits, perchannel=False, sym=True, mse=False, norm=2.4, grid=100, maxshrink=.8,
                  trits=False):
        self.wbits = bits
        self.maxq = torch.tensor(2 ** bits - 1)
        self.perchannel = perchannel
        self.sym = sym
        self.mse = mse
        self.norm = norm
        self.grid = grid
        self.maxshrink = maxshrink
        if trits:
            self.maxq = torch.tensor(-1)

    def find_params(self, x, weight=True):
        """Row-wise grid of a weight matrix (quant.py:37-77,105-109) via gptq_find_params."""
        if int(self.maxq) < 0:
            raise NotImplementedError("trits are outside the MI355X hot-path scope")
        if self.mse:
            raise NotImplementedError("mse grid search is outside the MI355X hot-path scope (never enabled by the drivers)")
        if not (self.perchannel and weight):
            raise NotImplementedError("only perchannel=True, weight=True grids are in the MI355X hot-path scope")
        dev = _dev_of(x)
        self.maxq = self.maxq.to(dev)
        w = x.flatten(1)
        if w.dtype != torch.float32 or w.stride(1) != 1:
            w = w.float().contiguous()
        R, C = w.shape
        scale = torch.empty(R, device=dev, dtype=torch.float32)
        zero = torch.empty(R, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("gptq_find_params", _lib.ptr(w), w.stride(0), R, 0, C, C, int(self.wbits), int(bool(self.sym)),
                      _lib.ptr(scale), _lib.ptr(zero), 1, 0, _lib.stream(dev))
        shape = [-1] + [1] * (x.dim() - 1)
        self.scale = scale.reshape(shape)
        self.zero = zero.reshape(shape)

    def quantize(self, x):
        if self.ready():
            return quantize(x, self.scale, self.zero, self.maxq)
        return x

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        return torch.all(self.scale != 0)


def _pack(weight, scales, zeros_scaled, bits):
    """[out, in] weights on the grid -> int32 [in/32*bits, out] (quant.py:158-186 / zeroShot quant.py:176-185)."""
    dev = _dev_of(weight)
    w = weight if weight.stride(-1) == 1 else weight.contiguous()
    n_out, n_in = w.shape
    s = scales.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
    z = zeros_scaled.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
    qweight = torch.empty((n_in // 32 * bits, n_out), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("gptq_pack_weights", _lib.ptr(w), _lib.dtype_code(w), w.stride(0), n_out, n_in,
                  _lib.ptr(s), _lib.ptr(z), bits, _lib.ptr(qweight), _lib.stream(dev))
    return qweight


def pack_codes(codes, bits):
    """uint8 integer codes [out, in] -> int32 [in/32*bits, out]; same layouts as `_pack`."""
    dev = _dev_of(codes)
    c = codes.contiguous()
    n_out, n_in = c.shape
    qweight = torch.empty((n_in // 32 * bits, n_out), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("gptq_pack_codes", _lib.ptr(c), c.stride(0), n_out, n_in, bits, _lib.ptr(qweight), _lib.stream(dev))
    return qweight


def dequant_packed(qweight, scale, zero, bits, groupsize, dtype=torch.float16):
    """int32 [in/32*bits, out] + grids [G, out] (`zero` = integer zero point) -> dense [out, in] weights,
    computed as scale * (code - zero) like the solver itself (quant.py:10)."""
    dev = _dev_of(qweight)
    n_out = qweight.shape[1]
    n_in = qweight.shape[0] // bits * 32
    W = torch.empty((n_out, n_in), dtype=dtype, device=dev)
    s = scale.to(torch.float32).contiguous()
    z = zero.to(torch.float32).contiguous()
    with torch.cuda.device(dev):
        _lib.call("gptq_dequant_packed", _lib.ptr(qweight.contiguous()), _lib.ptr(s), _lib.ptr(z), n_out, n_in, int(bits),
                  int(groupsize), _lib.ptr(W), _lib.dtype_code(W), W.stride(0), _lib.stream(dev))
    return W


class _QuantLinearBase(nn.Module):
    bits = 0

    def _alloc(self, infeatures, outfeatures):
        self.register_buffer('zeros', torch.zeros((outfeatures, 1)))
        self.register_buffer('scales', torch.zeros((outfeatures, 1)))
        self.register_buffer('bias', torch.zeros(outfeatures))
        self.register_buffer('qweight', torch.zeros((infeatures // 32 * self.bits, outfeatures), dtype=torch.int))

    def pack(self, linear, scales, zeros):
        """Same contract as Quant3Linear.pack (quant.py:152-187): `zeros` is the integer zero point.

        Tensors may live on the host (the reference packs on the CPU, opt.py:370); they are
        staged on the GPU, packed there and the buffers return to the source device.
        """
        home = linear.weight.device
        dev = home if home.type == 'cuda' else torch.device('cuda', torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        if dev is None:
            raise _lib.GptqHipError("packing needs an MI355X device; gptq_amd has no CPU path")
        scales = scales.to(dev)
        self.zeros = (zeros.to(dev) * scales).to(home)
        self.scales = scales.clone().to(home)
        if linear.bias is not None:
            self.bias = linear.bias.detach().clone()
        qweight = _pack(linear.weight.data.to(dev), scales, self.zeros.to(dev), self.bits)
        self.qweight = qweight.to(home)

    def forward(self, x):
        if x.shape[-1] == x.numel():
            outshape = list(x.shape)
            y = self.bias.clone().float()
            outshape[-1] = self.bias.numel()
            dtype = x.dtype
            if getattr(self, 'faster', False):
                x = x.half()
            else:
                x = x.float()
            self._matvec(x, y)
            y = y.to(dtype)
            return y.reshape(outshape)
        raise ValueError('Only supports a single token currently.')


class Quant3Linear(_QuantLinearBase):
    """3-bit packed Linear, single-token forward (reference quant.py:139-203)."""
    bits = 3

    def __init__(self, infeatures, outfeatures, faster=False):
        super().__init__()
        self._alloc(infeatures, outfeatures)
        self.faster = faster

    def _matvec(self, x, y):
        if self.faster:
            quant_cuda.vecquant3matmul_faster(x, self.qweight, y, self.scales, self.zeros)
        else:
            quant_cuda.vecquant3matmul(x, self.qweight, y, self.scales, self.zeros)


class Quant4Linear(_QuantLinearBase):
    """4-bit packed Linear; buffer layout of zeroShot/models/quant.py:172-197 (nibble i%8 of word i//8).

    The reference only defines the layout (its `vecquant4matmul` kernel does not exist); the
    constructor accepts either (infeatures, outfeatures) like Quant3Linear or the reference's
    (linear, scales, zeros) form.
    """
    bits = 4

    def __init__(self, infeatures, outfeatures=None, zeros=None, faster=False):
        super().__init__()
        self.faster = faster
        if isinstance(infeatures, nn.Module):
            linear, scales = infeatures, outfeatures
            self._alloc(linear.in_features, linear.out_features)
            self.pack(linear, scales, zeros)
        else:
            self._alloc(infeatures, outfeatures)

    def _matvec(self, x, y):
        quant_cuda.vecquant4matmul(x, self.qweight, y, self.scales, self.zeros)


class QuantGroupLinear(nn.Module):
    """Packed Linear with one (scale, zero) per `groupsize` input columns.

    The reference cannot pack grouped models (Quant3Linear keeps per-row scalars, quant.py:144-145, and
    fasterquant leaves only the LAST group's grid in the quantizer).  Buffers: `qweight` int32
    [in/32*bits, out] (same bit layouts as the per-row modules, original column order), `scales` /
    `zeros` fp32 [in/groupsize, out] (`zeros` = zero*scale), `bias`.  Filled from a finished `GPTQ`
    object (`codes`, `group_scale`, `group_zero`); valid for static groups or runs without act-order.
    """

    def __init__(self, bits, groupsize, infeatures, outfeatures):
        super().__init__()
        if bits not in (3, 4) or groupsize % 32 or infeatures % groupsize:
            raise ValueError("QuantGroupLinear: bits in {3,4}, groupsize % 32 == 0, in_features % groupsize == 0")
        self.bits, self.groupsize = bits, groupsize
        self.register_buffer('qweight', torch.zeros((infeatures // 32 * bits, outfeatures), dtype=torch.int))
        self.register_buffer('scales', torch.zeros((infeatures // groupsize, outfeatures)))
        self.register_buffer('zeros', torch.zeros((infeatures // groupsize, outfeatures)))
        self.register_buffer('bias', torch.zeros(outfeatures))

    @classmethod
    def from_gptq(cls, solver, bits):
        """Build from a `GPTQ` object after `fasterquant(groupsize=g, ...)`."""
        gs, gz = solver.group_scale, solver.group_zero          # [out, G]
        if gs is None:
            raise ValueError("QuantGroupLinear.from_gptq: the solver ran without groups")
        if solver.perm is not None and not getattr(solver, "static_groups", False):
            raise ValueError("QuantGroupLinear.from_gptq: act-order needs static groups for a g_idx-free format")
        n_out, n_in = solver.codes.shape
        m = cls(bits, n_in // gs.shape[1], n_in, n_out)
        m.qweight = pack_codes(solver.codes, bits)
        m.scales = gs.t().contiguous()
        m.zeros = (gz * gs).t().contiguous()
        bias = getattr(solver.layer, "bias", None)
        m.bias = bias.detach().clone().float() if bias is not None else torch.zeros(n_out, device=gs.device)
        return m

    def forward(self, x):
        if x.shape[-1] == x.numel():
            outshape = list(x.shape)
            y = self.bias.clone().float()
            outshape[-1] = self.bias.numel()
            dtype = x.dtype
            xv = x if x.dtype in (torch.float16, torch.float32) else x.float()
            quant_cuda.vecquant_matmul_grouped(xv.reshape(-1), self.qweight, y, self.scales, self.zeros,
                                               self.bits, self.groupsize)
            return y.to(dtype).reshape(outshape)
        raise ValueError('Only supports a single token currently.')


def make_quant3(module, names, name='', faster=False):
    """Replace the named Linears by empty Quant3Linear modules (reference quant.py:205-216)."""
    _make_quant(module, names, name, Quant3Linear, faster)


def make_quant4(module, names, name='', faster=False):
    _make_quant(module, names, name, Quant4Linear, faster)


def _make_quant(module, names, name, cls, faster):
    if isinstance(module, _QuantLinearBase):
        return
    for attr in dir(module):
        tmp = getattr(module, attr)
        full = name + '.' + attr if name != '' else attr
        if full in names:
            setattr(module, attr, cls(tmp.in_features, tmp.out_features, faster=faster))
    for child_name, child in module.named_children():
        _make_quant(child, names, name + '.' + child_name if name != '' else child_name, cls, faster)
