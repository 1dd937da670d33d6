"""Layer-by-layer quantization driver and perplexity evaluator.

Own counterpart of the reference's `opt_sequential` / `llama_sequential` (opt.py:29-228,
llama.py:31-207) and `opt_eval` / `llama_eval` (opt.py:230-359, llama.py:209-324); the reference
drivers read a module-global `args` and cannot travel to the GPU box.  One generic implementation
serves both families:

  * layer-0 inputs are captured with ALL keyword arguments the decoder layer received
    (attention_mask, position_ids, position_embeddings, cache_position, ...), so modern
    `transformers` LLaMA layers get their rotary embeddings (the reference never forwards them);
  * `true_sequential` quantizes the four LLaMA groups [k,v,q] -> [o] -> [up,gate] -> [down], each after
    its own hooked forward pass (upstream semantics; the reference fork's loop body is dedented out
    of the group loop, llama.py:106-110, so it only ever quantizes `down_proj`);
  * one transformer block is resident on the device at a time, like the reference.

The hot path itself (add_batch / fasterquant) is `gptq_amd.GPTQ`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from .gptq import GPTQ, fasterquant_many
from .modelutils import find_layers
from .quant import Quantizer, quantize

LLAMA_GROUPS = [['self_attn.k_proj', 'self_attn.v_proj', 'self_attn.q_proj'], ['self_attn.o_proj'],
                ['mlp.up_proj', 'mlp.gate_proj'], ['mlp.down_proj']]


@dataclass
class QuantArgs:
    """The reference CLI flags that reach the hot path (opt.py:514-657, llama.py:344-456)."""
    wbits: int = 4
    sym: bool = False
    percdamp: float = 0.01
    groupsize: int = -1
    act_order: bool = False
    static_groups: Optional[bool] = None   # None = the reference driver's default: True for OPT (opt.py:585), False for LLaMA (llama.py:399)
    true_sequential: bool = False
    nsamples: int = 128
    nearest: bool = False
    blocksize: int = 128
    hessian_defer: int = 16     # hook inputs folded into H per launch (gptq_amd.gptq.HESSIAN_DEFER)
    prefetch_blocks: bool = __import__('os').environ.get('GPTQ_SEQ_PREFETCH', '1') == '1'   # upload block i + 1 / download block i - 1 on a copy stream while block i is calibrated and
                                   # solved (the reference moves blocks synchronously, opt.py:104, 219)
    early_exit: bool = __import__('os').environ.get('GPTQ_SEQ_EARLY', '1') == '1'        # leave a hooked calibration pass once every Linear of the group has fired its hook
    cache_outputs: bool = __import__('os').environ.get('GPTQ_SEQ_CACHE', '1') == '1'      # true-sequential runs: once a group is solved, the outputs of its (now final) Linears
                                   # are kept per calibration sample and looked up by the later hooked passes and the final pass of the block instead
                                   # of being recomputed -- every Linear of a block then runs ONCE per sample instead of up to four times (the
                                   # reference recomputes, llama.py:97-139); same values (same GEMM, same inputs), up to `cache_max_bytes` per block
    cache_max_bytes: int = 96 << 30
    forward_batch: int = int(__import__('os').environ.get('GPTQ_SEQ_BATCH', '1'))   # calibration samples per block forward.  1 = the reference (opt.py:187, 216: one
                                   # sample per call, sized for small GPUs); more samples per call turn the block forwards -- 90 % of a block end to
                                   # end -- from 2048-row into 16384-row GEMMs (288 GB of HBM hold them).  The hooks then see [B, S, C] inputs: the same
                                   # running mean (gptq.py:38-65 takes any batch), but the GEMM library may pick other kernels for the larger shapes,
                                   # so activations and H can differ from the one-by-one run in their last bits.


class _Stop(Exception):
    pass


def _family(model):
    inner = model.model
    if hasattr(inner, "decoder"):                       # OPT
        dec = inner.decoder
        pre = [dec.embed_tokens, dec.embed_positions]
        for name in ("project_in", "project_out"):
            if getattr(dec, name, None) is not None:
                pre.append(getattr(dec, name))
        return dict(kind="opt", layers=dec.layers, pre=pre, prefix="model.decoder.layers",
                    norm=getattr(dec, "final_layer_norm", None), project_out=getattr(dec, "project_out", None))
    pre = [inner.embed_tokens, inner.norm]              # LLaMA-like
    if getattr(inner, "rotary_emb", None) is not None:
        pre.append(inner.rotary_emb)
    return dict(kind="llama", layers=inner.layers, pre=pre, prefix="model.layers", norm=inner.norm, project_out=None)


def _capture_layer0(model, fam, batches: Iterable[torch.Tensor], nsamples: int, dev):
    """Run the embedding front-end on `dev` and record what decoder layer 0 receives
    (opt.py:37-66, llama.py:39-66)."""
    layers = fam["layers"]
    for m in fam["pre"]:
        m.to(dev)
    layers[0] = layers[0].to(dev)
    dtype = next(iter(model.parameters())).dtype
    inps = torch.zeros((nsamples, model.seqlen, model.config.hidden_size), dtype=dtype, device=dev)
    cache = {"i": 0, "kwargs": None}

    class Catcher(nn.Module):
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, inp, *args, **kwargs):
            inps[cache["i"]] = inp
            cache["i"] += 1
            cache["kwargs"] = {k: v for k, v in kwargs.items()
                               if k not in ("past_key_value", "past_key_values", "use_cache", "output_attentions")}
            raise _Stop

    layers[0] = Catcher(layers[0])
    try:
        for batch in batches:
            try:
                model(batch.to(dev))
            except _Stop:
                pass
    finally:
        layers[0] = layers[0].module
    layers[0] = layers[0].cpu()
    for m in fam["pre"]:
        m.cpu()
    torch.cuda.empty_cache()
    return inps, cache["kwargs"] or {}


def _run_layer(layer, x, kwargs):
    """x: one sample [S, C] or a batch of samples [B, S, C] (attention mask / position tensors of the capture broadcast)."""
    out = layer(x.unsqueeze(0) if x.dim() == 2 else x, **kwargs)
    out = out[0] if isinstance(out, (tuple, list)) else out
    return out.reshape(x.shape)


@torch.no_grad()
def quantize_sequential(model, dataloader, dev, args: QuantArgs, group=None, timings: Optional[dict] = None) -> Dict[str, Quantizer]:
    """GPTQ every Linear of every decoder block; returns {full_name: quantizer} like the reference.

    With torch.distributed initialised (one process per GPU) the run is DATA-PARALLEL over the calibration samples
    (SURVEY section 8e): rank r captures and forwards only samples r, r + world, ... (so the block forwards are
    sharded too, and no activation ever crosses a link), folds them into its Hessians, and
    `parallel.fasterquant_sharded` does the rest per hooked group -- one all-reduce of H per distinct Hessian, the rows
    of W split over the ranks, one all-gather of the packed rows + grids, every rank rebuilding ALL Linears from the
    packed form, so all ranks hold bit-identical weights for the next forward pass."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if world > 1 and args.wbits not in (3, 4):
        raise NotImplementedError("sharded runs exchange packed weights: wbits must be 3 or 4")
    from . import gptq as _gptq_mod
    old_defer = _gptq_mod.HESSIAN_DEFER
    # deferral is safe here: every hooked forward produces fresh activation tensors and nothing below writes into them
    _gptq_mod.HESSIAN_DEFER = max(1, int(args.hessian_defer))
    try:
        return _quantize_sequential(model, dataloader, dev, args, group, world, rank, timings)
    finally:
        _gptq_mod.HESSIAN_DEFER = old_defer            # later drop-in use in this process gets the safe default back


class _BlockMover:
    """Block i on the device while it is worked on, block i + 1 already on its way and block i - 1 on its way back, both
    on a copy stream through page-locked host buffers.  The reference does `layers[i].to(dev)` ... `layers[i].cpu()` on
    the compute stream (opt.py:104, 219); `prefetch=False` does exactly that."""

    def __init__(self, layers, dev, prefetch):
        self.layers, self.dev, self.prefetch = layers, dev, prefetch
        self.stream = torch.cuda.Stream(device=dev) if prefetch else None
        self.ready = {}          # block index -> event recorded on the copy stream after its upload
        self.host = {}           # block index -> [(tensor holder, page-locked host tensor)] for the way back
        self.pending = []        # (event, block index): downloads in flight

    def _tensors(self, i):
        return list(self.layers[i].parameters()) + list(self.layers[i].buffers())

    def _upload(self, i):
        if i >= len(self.layers) or i in self.ready:
            return
        with torch.cuda.stream(self.stream):
            pairs = []
            for p in self._tensors(i):
                if p.device.type != "cpu":
                    continue
                h = p.data if p.data.is_pinned() else p.data.pin_memory()   # page-locked: the copies really are asynchronous
                pairs.append((p, h))
                p.data = h.to(self.dev, non_blocking=True)
            self.host[i] = pairs
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.ready[i] = ev

    def fetch(self, i):
        """Block i on the device (the compute stream waits for its upload); starts the upload of block i + 1."""
        if not self.prefetch:
            self.layers[i] = self.layers[i].to(self.dev)
            return self.layers[i]
        self._upload(i)
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(self.ready.pop(i))
        for t in self._tensors(i):
            t.data.record_stream(cur)                # allocated on the copy stream, used (and freed) on this one
        return self.layers[i]

    def prefetch_next(self, i):
        """Start the upload of block i + 1 (call it once work for block i is enqueued: page-locking the next block's
        weights is host work that would otherwise leave the GPU idle)."""
        if self.prefetch:
            self._upload(i + 1)

    def release(self, i):
        if not self.prefetch:
            self.layers[i] = self.layers[i].cpu()
            torch.cuda.empty_cache()
            return
        # download behind the block's last use, on the copy stream, into the page-locked buffers it came from: the host
        # does not wait, and there is no empty_cache() (a device-wide synchronisation per block)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(done)
            for p, h in self.host.pop(i, []):
                d = p.data
                h.copy_(d, non_blocking=True)
                d.record_stream(self.stream)
                p.data = h
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pending.append(ev)

    def finish(self):
        for ev in self.pending:
            ev.synchronize()                         # every block is back on the host
        self.pending = []


class _GroupDone(Exception):
    pass


def _quantize_sequential(model, dataloader, dev, args, group, world, rank, timings=None):
    import time
    from . import gptq as gmod
    from . import parallel as par
    t_wall = time.perf_counter()
    use_cache = model.config.use_cache
    model.config.use_cache = False
    fam = _family(model)
    static_groups = args.static_groups if args.static_groups is not None else fam["kind"] == "opt"
    layers = fam["layers"]
    batches = [b[0] for b in dataloader][:args.nsamples]
    mine = list(range(rank, len(batches), world))               # this rank's calibration samples
    inps, kwargs = _capture_layer0(model, fam, (batches[j] for j in mine), len(mine), dev)
    outs = torch.zeros_like(inps)
    quantizers: Dict[str, Quantizer] = {}
    records: List[dict] = []
    kw = dict(blocksize=args.blocksize, percdamp=args.percdamp, groupsize=args.groupsize, actorder=args.act_order,
              static_groups=static_groups)
    mover = _BlockMover(layers, dev, bool(args.prefetch_blocks))
    marks = []                                                  # (phase, start event, end event)
    flush_events = [] if timings is not None else None

    def timed(phase):
        class _T:
            def __enter__(self_):
                if timings is not None:
                    self_.a = torch.cuda.Event(enable_timing=True)
                    self_.a.record()

            def __exit__(self_, *exc):
                if timings is not None:
                    b = torch.cuda.Event(enable_timing=True)
                    b.record()
                    marks.append((phase, self_.a, b))
        return _T()

    for i in range(len(layers)):
        with timed("transfer"):
            layer = mover.fetch(i)
        full = find_layers(layer)
        if args.true_sequential and fam["kind"] == "llama":
            groups = [[n for n in g if n in full] for g in LLAMA_GROUPS]
            named = {n for g in groups for n in g}
            rest = [n for n in full if n not in named]          # Linears the four groups do not name: quantized last,
            groups = [g for g in groups if g] + ([rest] if rest else [])   # never silently skipped
        else:
            groups = [list(full.keys())]
        # outputs of solved Linears per sample batch (cache_outputs): module -> {first sample index: output}
        out_cache, cached_fwd, cache_bytes, cur_j = {}, {}, [0], [0]

        def keep_outputs(lin):
            orig, store = lin.forward, {}

            def fwd(x, _orig=orig, _store=store):
                hit = _store.get(cur_j[0])
                if hit is not None:
                    y, ver = hit
                    if y._version != ver:                         # somebody wrote into the Linear's output in place
                        raise RuntimeError("quantize_sequential: a cached Linear output was modified in place by the model's "
                                           "forward; run with QuantArgs(cache_outputs=False)")
                    return y
                y = _orig(x)
                if cache_bytes[0] + y.numel() * y.element_size() <= args.cache_max_bytes:
                    _store[cur_j[0]] = (y, y._version)
                    cache_bytes[0] += y.numel() * y.element_size()
                return y
            cached_fwd[lin] = orig
            out_cache[lin] = store
            lin.forward = fwd

        for gi, names in enumerate(groups):
            solvers = {}
            for n in names:
                solvers[n] = GPTQ(full[n])
                solvers[n].quantizer = Quantizer()
                solvers[n].quantizer.configure(args.wbits, perchannel=True, sym=args.sym, mse=False)

            # A hooked pass only exists to feed add_batch (its outputs are overwritten by the pass with quantized weights
            # below, opt.py:216-217): once every Linear of the group has seen the sample, the rest of the block forward
            # is skipped.  Same hook inputs as the reference's full passes; [k,v,q] stops in front of the attention.
            seen = set()

            def hook(name):
                def fn(_, inp, out):
                    solvers[name].add_batch(inp[0].data, out.data)
                    seen.add(name)
                    if args.early_exit and len(seen) == len(names):
                        raise _GroupDone
                return fn

            def pre_hook(name):
                # with early exit the hook sits IN FRONT of the Linear (add_batch reads the input only, gptq.py:38-65): the
                # pass ends before the group's last Linear computes an output nobody reads
                def fn(_, inp):
                    solvers[name].add_batch(inp[0].data, None)
                    seen.add(name)
                    if len(seen) == len(names):
                        raise _GroupDone
                return fn

            if args.early_exit and not gmod.DEBUG:
                handles = [full[n].register_forward_pre_hook(pre_hook(n)) for n in names]
            else:
                handles = [full[n].register_forward_hook(hook(n)) for n in names]
            gmod.FLUSH_EVENTS = flush_events
            fb = max(1, int(args.forward_batch))
            with timed("forward_hooked"):
                for j in range(0, len(mine), fb):
                    seen.clear()
                    cur_j[0] = j
                    try:
                        if fb == 1:
                            outs[j] = _run_layer(layer, inps[j], kwargs)
                        else:
                            outs[j:j + fb] = _run_layer(layer, inps[j:j + fb], kwargs)
                    except _GroupDone:
                        pass
            for h in handles:
                h.remove()
            mover.prefetch_next(i)
            with timed("solve"):
                if world > 1:
                    par.fasterquant_sharded([solvers[n] for n in names], bits=args.wbits, group=group, **kw)
                else:
                    fasterquant_many([solvers[n] for n in names], **kw)
            gmod.FLUSH_EVENTS = None
            for n in names:
                key = f"{fam['prefix']}.{i}.{n}"
                quantizers[key] = solvers[n].quantizer
                records.append(dict(name=key, error=solvers[n].error))
                solvers[n].free()
            if args.cache_outputs and len(groups) > 1:          # (a single group: its Linears run once after the solve anyway)
                for n in names:
                    keep_outputs(full[n])
        fb = max(1, int(args.forward_batch))
        with timed("forward_final"):
            for j in range(0, len(mine), fb):                 # opt.py:216-217: next block sees quantized outputs
                cur_j[0] = j
                if fb == 1:
                    outs[j] = _run_layer(layer, inps[j], kwargs)
                else:
                    outs[j:j + fb] = _run_layer(layer, inps[j:j + fb], kwargs)
        for lin, orig in cached_fwd.items():                  # the block's Linears get their own forward back
            del lin.forward                                   # (the instance attribute shadowed the class's method)
        out_cache.clear()
        cached_fwd.clear()
        del layer
        with timed("transfer"):
            mover.release(i)
        inps, outs = outs, inps
    mover.finish()
    model.config.use_cache = use_cache
    quantize_sequential.last_records = records
    if timings is not None:
        torch.cuda.synchronize(dev)
        for phase, a, b in marks:
            timings[phase] = timings.get(phase, 0.0) + a.elapsed_time(b)
        timings["hessian"] = timings.get("hessian", 0.0) + sum(a.elapsed_time(b) for (_, _, a, b) in flush_events)
        timings["blocks"] = len(layers)
        timings["wall_s"] = time.perf_counter() - t_wall
    return quantizers


@torch.no_grad()
def eval_ppl(model, testenc, dev, args: Optional[QuantArgs] = None) -> float:
    """Perplexity of `model` on a token tensor [1, T] (opt.py:230-334); `args.nearest` applies
    round-to-nearest per Linear on the fly (the RTN baseline, opt.py:289-300)."""
    if hasattr(testenc, "input_ids"):
        testenc = testenc.input_ids
    nsamples = testenc.numel() // model.seqlen
    use_cache = model.config.use_cache
    model.config.use_cache = False
    fam = _family(model)
    layers = fam["layers"]
    batches = [testenc[:, i * model.seqlen:(i + 1) * model.seqlen] for i in range(nsamples)]
    inps, kwargs = _capture_layer0(model, fam, batches, nsamples, dev)
    outs = torch.zeros_like(inps)
    for i in range(len(layers)):
        layer = layers[i].to(dev)
        if args is not None and args.nearest:
            for lin in find_layers(layer).values():
                q = Quantizer()
                q.configure(args.wbits, perchannel=True, sym=args.sym, mse=False)
                W = lin.weight.data
                q.find_params(W, weight=True)
                lin.weight.data = quantize(W.float(), q.scale, q.zero, q.maxq).to(W.dtype)
        for j in range(nsamples):
            outs[j] = _run_layer(layer, inps[j], kwargs)
        layers[i] = layer.cpu()
        del layer
        torch.cuda.empty_cache()
        inps, outs = outs, inps
    norm, proj = fam["norm"], fam["project_out"]
    if norm is not None:
        norm.to(dev)
    if proj is not None:
        proj.to(dev)
    model.lm_head.to(dev)
    testenc = testenc.to(dev)
    nlls = []
    for i in range(nsamples):
        h = inps[i].unsqueeze(0)
        if norm is not None:
            h = norm(h)
        if proj is not None:
            h = proj(h)
        logits = model.lm_head(h)
        shift_logits = logits[:, :-1, :].contiguous()
        labels = testenc[:, i * model.seqlen:(i + 1) * model.seqlen][:, 1:]
        loss = nn.functional.cross_entropy(shift_logits.view(-1, shift_logits.size(-1)).float(), labels.reshape(-1))
        nlls.append(loss.float() * model.seqlen)
    ppl = torch.exp(torch.stack(nlls).sum() / (nsamples * model.seqlen))
    model.config.use_cache = use_cache
    return float(ppl.item())


# reference-style names
def opt_sequential(model, dataloader, dev, args: QuantArgs):
    return quantize_sequential(model, dataloader, dev, args)


def llama_sequential(model, dataloader, dev, args: QuantArgs):
    return quantize_sequential(model, dataloader, dev, args)


def opt_eval(model, testenc, dev, args: Optional[QuantArgs] = None):
    return eval_ppl(model, testenc, dev, args)


def llama_eval(model, testenc, dev, args: Optional[QuantArgs] = None):
    return eval_ppl(model, testenc, dev, args)


# ------------------------------------------------------------------------------------------------
# Row f3: packed checkpoints and the token-by-token generation benchmark (opt.py:362-507)
# ------------------------------------------------------------------------------------------------
def _set_module(root, dotted, new):
    parent = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        parent = getattr(parent, p)
    setattr(parent, parts[-1], new)


def pack_model(model, quantizers, bits, faster=False):
    """Replace every quantized Linear by a packed module in place, like `opt_pack3` (opt.py:362-373).

    Per-row grids only (what `Quant3Linear` / `Quant4Linear` can hold): run the quantization with
    groupsize = -1.  The packing itself happens on the GPU (the reference loops over rows in numpy,
    "TODO: perform packing on GPU", opt.py:361)."""
    from .quant import Quant3Linear, Quant4Linear
    cls = {3: Quant3Linear, 4: Quant4Linear}[bits]
    layers = find_layers(model)
    for name, q in quantizers.items():
        lin = layers[name]
        packed = cls(lin.in_features, lin.out_features, faster=faster)
        packed.pack(lin, q.scale.float(), q.zero.float())
        _set_module(model, name, packed)
    return model


def make_packed_skeleton(model, names, bits, faster=False):
    """Empty packed modules for `names` (the load side of opt.py:375-402): load_state_dict then fills
    `qweight / scales / zeros / bias`."""
    from .quant import make_quant3, make_quant4
    (make_quant3 if bits == 3 else make_quant4)(model, names, faster=faster)
    return model


@torch.no_grad()
def benchmark(model, input_ids, dev, check=False, warmup=10):
    """Token-by-token generation timing with the KV cache (opt.py:440-507): every Linear call is a
    single-token packed mat-vec.  Returns (median seconds per token, PPL or None)."""
    import time
    import numpy as np
    input_ids = input_ids.to(dev)
    attention_mask = torch.ones((1, input_ids.numel()), device=dev)
    for _ in range(warmup):
        model(input_ids[:, 0].reshape((1, -1)), past_key_values=None, attention_mask=attention_mask[:, :1])
    torch.cuda.synchronize()
    past = None
    times, tot = [], 0.0
    loss = nn.CrossEntropyLoss()
    for i in range(input_ids.numel()):
        tick = time.time()
        out = model(input_ids[:, i].reshape((1, -1)), past_key_values=past,
                    attention_mask=attention_mask[:, :(i + 1)].reshape((1, -1)), use_cache=True)
        torch.cuda.synchronize()
        times.append(time.time() - tick)
        if check and i != input_ids.numel() - 1:
            tot += loss(out.logits[0].float(), input_ids[:, (i + 1)]).float()
        past = out.past_key_values
        del out
    ppl = float(torch.exp(tot / (input_ids.numel() - 1)).item()) if check else None
    return float(np.median(times)), ppl
