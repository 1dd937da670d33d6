"""GPTQ solver object with the reference's protocol (gptq.py:23-318), backed by libgptq_hip.so.

Callers construct `GPTQ(linear)`, assign `.quantizer`, feed `add_batch(inp, out)` from a
forward hook, call `fasterquant(...)`, read `.quantizer` back and `free()` (opt.py:172-214).
Host orchestration only: every FLOP of the path (Hessian SYRK, damped inverse factor,
column loop, trailing updates) is a gfx950 kernel behind the C ABI in include/gptq_hip.h.
"""
from __future__ import annotations

import functools
import math
import threading
import time
import weakref

import torch
import torch.nn as nn

try:  # Conv1D support mirrors gptq.py:31-32; transformers is optional here
    import transformers
    _Conv1D = transformers.Conv1D
except Exception:  # pragma: no cover
    _Conv1D = ()

from . import _lib
from .quant import *  # noqa: F401,F403  (the reference's gptq.py re-exports quant.*, drivers rely on it)

DEBUG = False
VERBOSE = True   # the reference prints `time` / `error` per layer (gptq.py:293-294)
# The fork stores `self.input = mean(scaled batch)` in add_batch (gptq.py:63); only its out-of-scope
# `analyse` / non_linear branches read it.  It costs an extra pass over X per call, so it is opt-in.
TRACK_INPUT_MEAN = False
# HESSIAN_DEFER = 1 (default): add_batch enqueues the update of H for its input before it returns, exactly like the
# reference (gptq.py:59-65 consumes `inp` inside the call): the launch is ordered on the current stream, so the caller
# may overwrite or recycle the activation buffer right after the hook.
# HESSIAN_DEFER = k > 1: add_batch holds up to k hook inputs (references, no copies) and folds them into H with ONE
# launch -- the reference's own multi-sample batch formula (gptq.py:44, 59-65), so only rounding differs.  It divides
# the fp32 H read-modify-write traffic and the per-tile pipeline fill by the batch.  This REQUIRES that callers do not
# overwrite an activation in place after its hook returned (true for the OPT / LLaMA blocks: every forward produces
# fresh tensors).  What can be checked is checked: an input whose torch version counter moved, or whose storage is
# handed in again while still pending, raises instead of silently corrupting H.
HESSIAN_DEFER = 1
# STAGE_INPUTS (HESSIAN_DEFER = 1 only): add_batch COPIES its input into a buffer the library owns (enqueued on the
# current stream before the hook returns, so the caller may still overwrite or recycle the activation right away) and
# folds the copies of up to STAGE_INPUTS calls into H with ONE launch -- the batched launch's efficiency (a single
# 2048-token sample per launch runs the 256 x 256 kernel with K = 2048: 119 ms of Hessian time per Llama-7B block
# against 37 at 16 samples per launch) without keeping a reference to anything of the caller's.  The price is one pass
# over the input (2 x 45 MB at C = 11008: 25 us per sample) and 16 staged samples per object (0.27-0.72 GB).  What a caller
# can observe is unchanged: `nsamples` moves at once, `.H`, `fasterquant` and `free` fold what is staged first.
# 1 = off: one launch per call, like round 2's default.
STAGE_INPUTS = int(__import__('os').environ.get('GPTQ_STAGE_INPUTS', '16'))


# The reference's free() ends with torch.cuda.empty_cache() (gptq.py:313-318), which hands every cached block
# back to the driver (a device-wide synchronisation + re-allocation on the next Linear).  With 288 GB of HBM3E
# the cache is kept by default; set True for the reference's behaviour when memory is tight.
EMPTY_CACHE_ON_FREE = False

# The registries below (who holds deferred inputs, who shares a Hessian with whom, the solve streams) are process-wide.
# The reference's drivers are single-threaded; callers that drive several models from several Python threads are
# serialised on this lock around every entry point that reads or rewires the registries (add_batch, flush_pending,
# fasterquant / fasterquant_many up to the point where the work is enqueued, free).  GPU work itself is only ENQUEUED
# under the lock.  The knobs (HESSIAN_DEFER, LAZY_HESSIANS, ...) stay module attributes: set them before threads start.
_STATE_LOCK = threading.RLock()


def _locked(fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with _STATE_LOCK:
            return fn(*args, **kwargs)
    return wrapper


_DIRTY = weakref.WeakValueDictionary()   # id -> GPTQ objects holding deferred hook inputs (dropped objects vanish)
_LIVE = weakref.WeakSet()                # GPTQ objects whose Hessian is still being accumulated
FLUSH_EVENTS = None   # set to a list to collect ([C per problem], n_slabs, start_event, end_event) per Hessian call


# Linears that are fed the SAME input tensor from the same forward passes (q/k/v, gate/up) accumulate the same
# Hessian; the reference recomputes it per Linear (SURVEY 8a "identical H recomputed for k/v/q").  With
# SHARE_INPUT_HESSIANS the objects whose deferred inputs are the same device buffers (pointer, shape, strides,
# dtype) from their very first add_batch on keep ONE running H: the first object (the leader) owns it, the others
# copy it when they need their own (solve, `.H`, diverging inputs).  What every object ends up with is bit for bit
# what it would have computed alone -- the kernel, its inputs and the order of the updates are the same.
SHARE_INPUT_HESSIANS = True
# fasterquant_many solves objects that share a Hessian as one problem over the concatenation of their rows: one
# factorization chain and one column loop for q/k/v (rows of W are independent given H, gptq.py:262-276), bit for
# bit the per-object results.
JOINT_SOLVE = True
# When add_batch reaches HESSIAN_DEFER, fold only the widest Linears' inputs (their Hessians are the bulk of the flops
# and sit on the critical path of the block); the narrower ones keep collecting inputs -- references, no copies, up to
# LAZY_MAX_BYTES -- and fasterquant_many folds them on the side lanes, beside the widest Linear's solve, which is
# latency-bound and leaves the chip idle.  They go out as one grouped flush sized for LAZY_CUS compute units, so that the
# solve's small kernels always find free ones (measured on one box: 32 CUs 44.6 ms per block, 64: 37.4, 80: 36.6, 96: 36.6, 128: 38.5, 192: 38.9).
# Only in effect when HESSIAN_DEFER > 1 (the caller already opted into keeping hook inputs for a while).
LAZY_HESSIANS = True
LAZY_MAX_BYTES = 16 << 30
LAZY_CUS = int(__import__('os').environ.get('GPTQ_LAZY_CUS', '80'))   # 0 = no limit


def _input_signature(o):
    """Identity of the deferred inputs of `o` (device buffers, shapes, strides, dtypes), kept incrementally by
    add_batch so that planning a flush costs O(objects), not O(retained inputs)."""
    return (len(o._pending), o._sig)


def _plan_flush():
    """Settle who shares a running Hessian with whom for everything that holds deferred inputs; nothing is launched.
    Returns the objects that own an H to update (leaders and singles)."""
    objs = [o for o in list(_DIRTY.values()) if o._pending and o._H is not None]
    if not objs:
        return []
    sigs = {id(o): _input_signature(o) for o in objs}
    # followers whose inputs stopped matching their leader's take their own copy of H first
    for o in objs:
        L = o._leader
        if L is not None and (id(L) not in sigs or sigs[id(L)] != sigs[id(o)]):
            o._materialize()
    for o in objs:
        for f in list(o._followers):
            if id(f) not in sigs:                       # the leader moves on alone
                f._materialize()
    # fresh objects (no update yet) with identical inputs join the first of them
    if SHARE_INPUT_HESSIANS:
        fresh = {}
        for o in objs:
            if o._fresh and o._leader is None and not o._followers:
                key = (o.dev, o.columns, sigs[id(o)])
                L = fresh.setdefault(key, o)
                if L is not o:
                    o._leader = L
                    L._followers.append(o)
    for o in objs:
        o._fresh = False
    return [o for o in objs if o._leader is None]


def _launch_flush(work, n_cu=0):
    """Fold the deferred inputs of `work` (objects that own their H) into their H on the CURRENT stream, and settle the
    bookkeeping of their followers (same inputs).  One library call per (device, slab shape, batch structure, width)
    (Linears of different widths in one launch measured neutral: the larger K-split last round eats what the shared
    launch saves).  n_cu > 0 sizes the launches for that many compute units (gptq_hessian_accum_mixed)."""
    import ctypes
    work = [o for o in work if o._pending and o._H is not None]
    groups = {}
    for o in work:
        x0 = o._pending[0][0]
        for x, _, ver in o._pending:
            if x._version != ver:
                raise RuntimeError(
                    "add_batch: an input tensor was modified in place after its hook returned while its Hessian update "
                    "was still deferred (gptq_amd.gptq.HESSIAN_DEFER > 1 keeps references, not copies); set "
                    "HESSIAN_DEFER = 1 for callers that recycle activation buffers")
        key = (o.dev, x0.shape[0], x0.dtype, len(o._pending), sum(b for _, b, _ in o._pending), o.columns)
        groups.setdefault(key, []).append(o)
    for (dev, tokens, dtype, n_x, batch, _), members in groups.items():
        members.sort(key=lambda o: -o.columns)         # widest first: its tiles lead the launch
        for i in range(0, len(members), 64):
            chunk = members[i:i + 64]
            n = len(chunk)
            Hs = (ctypes.c_void_p * n)(*[o._H.data_ptr() for o in chunk])
            Xs = (ctypes.c_void_p * (n * n_x))(*[x.data_ptr() for o in chunk for x, _, _ in o._pending])
            nb = (ctypes.c_int * n)(*[int(o._applied) for o in chunk])
            ldh = (ctypes.c_int * n)(*[o._H.stride(0) for o in chunk])
            ldx = (ctypes.c_int * n)(*[o._pending[0][0].stride(0) for o in chunk])
            Cs = (ctypes.c_int * n)(*[o.columns for o in chunk])
            with torch.cuda.device(dev):
                if FLUSH_EVENTS is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                _lib.call("gptq_hessian_accum_mixed", n, Hs, ldh, Xs, n_x, _lib._DTYPES[dtype], ldx, Cs, tokens, nb,
                          int(batch), int(n_cu), _lib.stream(dev))
                if FLUSH_EVENTS is not None:
                    ev1.record()
                    FLUSH_EVENTS.append(([o.columns for o in chunk], n_x, ev0, ev1))
    for L in work:
        for o in [L] + list(L._followers):
            o._applied += sum(b for _, b, _ in o._pending)
            o._pending = []
            o._ptrs = {}                 # (would pin the base tensors of the last hook inputs, e.g. the whole `inps` buffer)
            o._lower_stale = True
            _DIRTY.pop(id(o), None)


@_locked
def flush_pending(heavy_only=False):
    """Fold deferred hook inputs into their H on the current stream.  heavy_only (LAZY_HESSIANS): only the widest
    Linears' (>= half the largest in_features among those waiting); the others keep collecting inputs."""
    work = _plan_flush()
    if heavy_only and work:
        cmax = _widest_live()
        work = [o for o in work if 2 * o.columns >= cmax and len(o._pending) >= max(1, int(HESSIAN_DEFER))]
    _launch_flush(work)


def _widest_live():
    return max((o.columns for o in list(_LIVE) if o._H is not None), default=0)


def _retained_bytes():
    """Bytes of hook inputs kept alive by deferred updates (inputs shared by several objects count once per object)."""
    return sum(len(o._pending) * o._pending[0][0].numel() * o._pending[0][0].element_size()
               for o in list(_DIRTY.values()) if o._pending)


_SOLVE_STREAMS = {}   # device -> pool of streams for fasterquant_many
LANE_EVENTS = None    # set to a list to collect (lane, start_event, end_event, host_enqueue_seconds) per fasterquant_many lane
SOLVE_STREAMS = 3     # concurrent solves per device: the caller's stream (it carries the largest solve and owns the
                      # library's look-ahead helper stream) + 2 more = the 4 hardware queues HIP uses by default;
                      # streams beyond the queues serialize falsely


@_locked
def fasterquant_many(solvers, blocksize=128, percdamp=.01, groupsize=-1, actorder=False, static_groups=False,
                     max_concurrent=None):
    """`fasterquant` for several GPTQ objects at once (the `for name in subset:` loop of opt.py:189-214).

    The Linears hooked in one forward pass are independent problems and a single solve is a chain of
    small latency-bound kernels (one-workgroup diagonal factorizations, 32-128-workgroup column loops)
    that leaves most of the 256 CUs idle, so each solve is enqueued on its own HIP stream, largest first,
    and the results are collected afterwards.  Same kernels, same arithmetic, same results as calling
    `fasterquant` one by one."""
    solvers = list(solvers)
    if not solvers:
        return
    _plan_flush()                        # who shares a Hessian with whom; deferred inputs are folded further down
    # Objects that share one running Hessian (SHARE_INPUT_HESSIANS) and are all solved here become ONE problem over
    # the concatenation of their rows (_JointSolve); the others take private copies of H on the caller's stream,
    # before the solves fan out over the lanes.
    chosen = {id(g) for g in solvers}
    units, grouped = [], set()
    if JOINT_SOLVE:
        for g in solvers:
            if g._leader is None and g._followers and id(g) not in grouped:
                members = [g] + [f for f in g._followers if id(f) in chosen]
                if len(members) > 1 and _JointSolve.compatible(members):
                    units.append(_JointSolve(members))
                    grouped.update(id(m) for m in members)
    for g in solvers:
        if id(g) in grouped:
            continue
        if g._leader is not None or g._followers:      # copies of a shared H need it up to date, here and now
            _launch_flush([g._leader or g])
        g._materialize()
        g._release_followers()
        units.append(g)
    for u in units:
        if isinstance(u, _JointSolve):
            L = u.members[0]
            if any(id(f) not in grouped for f in L._followers):
                _launch_flush([L])
            u.detach()                   # followers outside this call copy H now; the group keeps the leader's
    solvers = units
    by_dev = {}
    for g in solvers:
        by_dev.setdefault(g.dev, []).append(g)
    states = []
    for dev, group in by_dev.items():
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            pool = _SOLVE_STREAMS.setdefault(dev, [])
            want = max(1, min(int(max_concurrent or SOLVE_STREAMS), len(group)))
            while len(pool) < want - 1:
                pool.append(torch.cuda.Stream(device=dev))
            lanes = [cur] + pool[:want - 1]          # the caller's stream carries the largest solve itself
            # seconds, roughly: ~200 us of latency-bound kernels per 128-column block + the GEMM flops at 100 TFLOP/s
            def cost(g):
                own = g.members[0] if isinstance(g, _JointSolve) else g
                hess = sum(x.shape[0] for x, _, _ in own._pending) * float(g.columns) ** 2 / 4e14   # deferred updates
                return hess + 2e-4 * (g.columns / 128.0) + (2.0 / 3.0 * g.columns ** 3 + g.rows * g.columns ** 2) / 1e14
            order = sorted(group, key=lambda g: -cost(g))
            load = [0.0] * want                      # longest-processing-time-first onto the streams
            used, mine = lanes[1:], []
            for st in used:                          # before anything of this call lands on `cur`:
                st.wait_stream(cur)                  # H, layer.weight were produced on the caller's stream
            if LANE_EVENTS is not None:
                lane_ev = [[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), 0.0] for _ in lanes]
                for st, le in zip(lanes, lane_ev):
                    le[0].record(st)
            placed = []                              # (unit, lane index), widest first
            for g in order:
                k = min(range(want), key=lambda j: (load[j], j))
                load[k] += cost(g)
                placed.append((g, k))
            owner = lambda g: g.members[0] if isinstance(g, _JointSolve) else g
            # deferred Hessian updates of the units on the side lanes: ONE grouped flush on the first side lane (their
            # launches are small; together they fill the chip better), the other side lanes wait for it
            late = [owner(g) for g, k in placed if k > 0 and owner(g)._pending]
            late_done = None

            def flush_late():
                nonlocal late_done
                if not late or late_done is not None:
                    return
                st1 = lanes[1]
                with torch.cuda.stream(st1):
                    for o in late:
                        for x, _, _ in o._pending:           # inputs were allocated on the caller's stream
                            x.record_stream(st1)
                    _launch_flush(late, n_cu=int(LAZY_CUS))
                    late_done = torch.cuda.Event()
                    late_done.record(st1)
                for st in lanes[2:]:
                    st.wait_event(late_done)

            for g, k in placed:
                st = lanes[k]
                if st is not cur and g._H is not None:
                    g._H.record_stream(st)
                t_host = time.perf_counter()
                if k > 0:
                    flush_late()                     # (enqueued after the widest solve went out on the caller's stream)
                with torch.cuda.stream(st):
                    own = owner(g)
                    if own._pending:                 # the caller's-stream units fold their own, at full width ...
                        if st is not cur:            # (inputs / staged copies were allocated on the caller's stream: their
                            for x, _, _ in own._pending:      # memory must not be handed out again before this lane has read them)
                                x.record_stream(st)
                        _launch_flush([own])
                        if k == 0 and late and late_done is None and want > 1:
                            wide_done = torch.cuda.Event()      # ... and the late updates start after them
                            wide_done.record(st)
                            lanes[1].wait_event(wide_done)
                    mine.append((g, g._solve_enqueue(blocksize, percdamp, groupsize, actorder, static_groups)))
                if LANE_EVENTS is not None:
                    lane_ev[k][2] += time.perf_counter() - t_host
            if LANE_EVENTS is not None:
                for j, (st, le) in enumerate(zip(lanes, lane_ev)):
                    le[1].record(st)
                    LANE_EVENTS.append((j, le[0], le[1], le[2]))
            for st in used:
                cur.wait_stream(st)                  # everything after this call sees the results
            for g, state in mine:
                for t in state.values():
                    if isinstance(t, torch.Tensor):
                        t.record_stream(cur)
            states += mine
    for g, state in states:
        g._solve_finish(state)


def _enqueue_rows(dev, W, H, quantizer, preset_grid, blocksize, percdamp, groupsize, actorder, static_groups,
                  factored=None):
    """Enqueue one gptq_fasterquant_rows call on the CURRENT stream for the rows of `W` [R, C] (fp32, contiguous) with
    the Hessian `H` (consumed: left holding the factor, see `Hinv_form`); nothing here waits for the GPU.  `preset_grid` = (scale, zero) of a
    ready quantizer (gptq.py:181) or None.  `factored` = dict(dead, perm, info): H already holds the factor form
    (gptq_amd.parallel.rfactor_sharded) of the Hessian that gptq_solve_prepare fixed.  Returns the buffers of the call."""
    q = quantizer
    tick = time.time()
    bits = int(getattr(q, "wbits", 0)) or (int(q.maxq) + 1).bit_length() - 1
    R, C = W.shape
    G = -(-C // groupsize) if groupsize > 0 else 0
    scale = torch.zeros(R, device=dev, dtype=torch.float32)
    zero = torch.zeros(R, device=dev, dtype=torch.float32)
    if preset_grid is not None:
        scale.copy_(preset_grid[0].reshape(-1))
        zero.copy_(preset_grid[1].reshape(-1))
    gscale = torch.empty((R, G), device=dev, dtype=torch.float32) if G else None
    gzero = torch.empty((R, G), device=dev, dtype=torch.float32) if G else None
    perm = torch.empty(C, device=dev, dtype=torch.int32) if actorder else None
    if factored is not None:
        perm = factored["perm"] if actorder else None
    codes = torch.empty((R, C), device=dev, dtype=torch.uint8)
    stat = torch.zeros(2, device=dev, dtype=torch.float32)          # [error, info (int32 bits)]
    row_loss = torch.empty(R, device=dev, dtype=torch.float32)
    info = stat[1:].view(torch.int32)
    lib = _lib.load()
    nbytes = lib.gptq_fasterquant_workspace_bytes(R, C, int(blocksize), int(groupsize), int(bool(actorder)),
                                                  int(bool(static_groups)))
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        if factored is None:
            _lib.call("gptq_fasterquant_rows", _lib.ptr(W), W.stride(0), _lib.ptr(H), H.stride(0), R, C, bits,
                      int(bool(q.sym)), int(blocksize), float(percdamp), int(groupsize), int(bool(actorder)),
                      int(bool(static_groups)), _lib.ptr(scale), _lib.ptr(zero), int(preset_grid is not None),
                      _lib.ptr(gscale), _lib.ptr(gzero), _lib.ptr(perm), _lib.ptr(codes), _lib.ptr(stat),
                      _lib.ptr(row_loss), _lib.ptr(info), _lib.ptr(ws), nbytes, _lib.stream(dev))
        else:
            _lib.call("gptq_fasterquant_rows_factored", _lib.ptr(W), W.stride(0), _lib.ptr(H), H.stride(0), R, C, bits,
                      int(bool(q.sym)), int(blocksize), int(groupsize), int(bool(actorder)), int(bool(static_groups)),
                      _lib.ptr(scale), _lib.ptr(zero), int(preset_grid is not None), _lib.ptr(gscale), _lib.ptr(gzero),
                      _lib.ptr(factored["dead"]), _lib.ptr(perm), _lib.ptr(codes), _lib.ptr(stat), _lib.ptr(row_loss),
                      _lib.ptr(info), _lib.ptr(ws), nbytes, _lib.stream(dev))
            info.copy_(factored["info"])          # a non-positive pivot of the (sharded) factorization
    form = lib.gptq_fasterquant_factor_form(C, int(blocksize), int(groupsize), int(bool(static_groups)))
    return dict(tick=tick, W=W, H=H, scale=scale, zero=zero, gscale=gscale, gzero=gzero, perm=perm, codes=codes,
                stat=stat, ws=ws, row_loss=row_loss, static_groups=bool(static_groups),
                factor_form="rfactor" if form else "hinv")


def _check_solved(st):
    """Wait for the solve (one sync, like torch.cuda.synchronize() at gptq.py:292); returns its `error` scalar."""
    host = st["stat"].cpu()
    bad = int(host[1:].view(torch.int32).item())
    if bad < 0:            # chol_panel_kernel: a bounded in-kernel wait gave up (never a property of H)
        raise _lib.GptqHipError(f"fasterquant: a workgroup hand-off of the factorization timed out (code {bad})")
    if bad:
        raise torch.linalg.LinAlgError(
            f"fasterquant: the damped Hessian is not positive-definite (pivot {bad}); cf. torch.linalg.cholesky")
    return float(host[0].item())


def _publish_rows(obj, st, a, b, error):
    """Rows [a, b) of a finished solve become the results of `obj` (what callers read back, gptq.py:213, 305, plus
    what packing grouped models needs)."""
    q = obj.quantizer
    obj.error = error
    if VERBOSE:
        print('time %.2f' % (time.time() - st["tick"]))
        print('error', obj.error)
    q.maxq = q.maxq.to(obj.dev)
    q.scale = st["scale"][a:b].reshape(-1, 1)
    q.zero = st["zero"][a:b].reshape(-1, 1)
    obj.Hinv = st["H"]                   # what the solver left in H: U ("hinv") or R = U^-1 with U's diagonal
    obj.Hinv_form = st["factor_form"]    # 128-blocks ("rfactor", include/gptq_hip.h: gptq_rfactor_upper)
    obj.codes = st["codes"][a:b]
    obj.group_scale = st["gscale"][a:b] if st["gscale"] is not None else None
    obj.group_zero = st["gzero"][a:b] if st["gzero"] is not None else None
    obj.perm = st["perm"]
    obj.static_groups = st["static_groups"]
    Q = st["W"][a:b]
    if _Conv1D and isinstance(obj.layer, _Conv1D):
        Q = Q.t()
    obj.layer.weight.data = Q.reshape(obj.layer.weight.shape).to(obj.layer.weight.data.dtype)


class _JointSolve:
    """GPTQ objects sharing one running Hessian, solved as one Linear whose rows are theirs stacked (leader first)."""

    def __init__(self, members):
        self.members = members
        self.dev = members[0].dev
        self.columns = members[0].columns
        self.rows = sum(m.rows for m in members)

    @staticmethod
    def compatible(members):
        q0 = members[0].quantizer
        for m in members:
            q = m.quantizer
            if not isinstance(m.layer, nn.Linear):
                return False
            if int(q.maxq) != int(q0.maxq) or bool(q.sym) != bool(q0.sym) or int(q.maxq) < 0:
                return False
            if q.scale.numel() == m.rows and bool(q.ready()):          # a preset grid (gptq.py:181): solve alone
                return False
            if m.columns != members[0].columns or m.dev != members[0].dev or m._H is None:
                return False
        return True

    @property
    def _H(self):
        return self.members[0]._H

    def detach(self):
        L = self.members[0]
        inside = {id(m) for m in self.members}
        for f in list(L._followers):
            if id(f) not in inside:
                f._materialize()
        for m in self.members[1:]:
            L._followers.remove(m)
            m._leader = None

    def _solve_enqueue(self, blocksize, percdamp, groupsize, actorder, static_groups):
        L = self.members[0]
        W = torch.cat([m.layer.weight.data.float() for m in self.members], 0).contiguous()
        H = L._H
        for m in self.members:
            m._applied = m.nsamples if m.nsamples else m._applied
            m._H = None                  # consumed (gptq.py:141-142)
        st = _enqueue_rows(self.dev, W, H, L.quantizer, None, blocksize, percdamp, groupsize, actorder, static_groups)
        bounds, r0 = [], 0
        for m in self.members:
            bounds.append((r0, r0 + m.rows))
            r0 += m.rows
        st["bounds"] = bounds
        st["errors"] = torch.stack([st["row_loss"][a:b].sum() for a, b in bounds])   # each member's own sum(Losses)
        return st

    def _solve_finish(self, st):
        _check_solved(st)
        errors = st["errors"].cpu()
        for m, (a, b), err in zip(self.members, st["bounds"], errors):
            _publish_rows(m, st, a, b, float(err.item()))


class GPTQ:

    def __init__(self, layer):
        self.layer = layer
        self.dev = self.layer.weight.device
        _lib.require_gpu(layer.weight, "layer.weight")
        shape = layer.weight.shape
        if isinstance(layer, nn.Conv2d):
            self.rows, self.columns = shape[0], int(math.prod(shape[1:]))
        elif _Conv1D and isinstance(layer, _Conv1D):
            self.rows, self.columns = shape[1], shape[0]
        else:
            self.rows, self.columns = shape[0], shape[1]
        self._H = torch.zeros((self.columns, self.columns), device=self.dev, dtype=torch.float32)
        self._lower_stale = False   # add_batch maintains the upper triangle only
        self.nsamples = 0
        self._pending = []          # deferred (x, batch, x._version) hook inputs, see HESSIAN_DEFER
        self._ptrs = {}             # their device addresses -> (base tensor, version) (aliasing guard of add_batch)
        self._applied = 0           # samples already folded into _H
        self._fresh = True          # no update folded in yet and H never assigned
        self._sig = 0               # running identity of the deferred inputs (_input_signature)
        _LIVE.add(self)
        self._leader = None         # object whose _H this one shares (SHARE_INPUT_HESSIANS)
        self._followers = []        # objects sharing this one's _H

    # -- shared running Hessians (SHARE_INPUT_HESSIANS) -------------------------------------------
    def _materialize(self):
        """Follower: take a private copy of the leader's H (its own, so far unused, buffer)."""
        L = self._leader
        if L is None:
            return
        self._leader = None
        L._followers.remove(self)
        if self._H is not None and L._H is not None:
            self._H.copy_(L._H)
            self._lower_stale = L._lower_stale

    def _release_followers(self):
        """Leader about to overwrite / drop its H: the followers copy it first."""
        for f in list(self._followers):
            f._materialize()

    # -- H stays reachable as a full symmetric tensor; mirrored lazily -------------
    def _flush(self):
        flush_pending()

    @property
    @_locked
    def H(self):
        self._flush()
        self._materialize()
        if self._H is not None and self._lower_stale:
            with torch.cuda.device(self.dev):
                _lib.call("gptq_symmetrize", _lib.ptr(self._H), self._H.stride(0), self.columns, _lib.stream(self.dev))
            self._lower_stale = False
        return self._H

    @H.setter
    def H(self, value):
        self._pending = []
        self._ptrs = {}
        _DIRTY.pop(id(self), None)
        self._release_followers()
        if self._leader is not None:
            self._leader._followers.remove(self)
            self._leader = None
        self._fresh = False
        self._H = value
        self._lower_stale = False

    @H.deleter
    def H(self):
        self._release_followers()
        if self._leader is not None:
            self._leader._followers.remove(self)
            self._leader = None
        self._H = None

    @_locked
    def add_batch(self, inp, out):
        """Running-mean Hessian update (gptq.py:38-65): H <- H*n/(n+b) + 2/(n+b) X^T X."""
        if DEBUG:
            self.inp1 = inp
            self.out1 = out
        if len(inp.shape) == 2:
            inp = inp.unsqueeze(0)
        batch = inp.shape[0]
        if isinstance(self.layer, nn.Conv2d):
            unfold = nn.Unfold(self.layer.kernel_size, dilation=self.layer.dilation,
                               padding=self.layer.padding, stride=self.layer.stride)
            x = unfold(inp).permute(0, 2, 1).reshape(-1, self.columns)      # tokens x (C*kh*kw)
        else:
            x = inp.reshape(-1, inp.shape[-1])                              # tokens x C
        _lib.require_gpu(x, "inp")
        if x.dtype not in (torch.float16, torch.bfloat16, torch.float32):
            x = x.float()
        if x.stride(-1) != 1:
            x = x.contiguous()
        if x.shape[1] != self.columns:
            raise ValueError(f"add_batch: input has {x.shape[1]} features, layer expects {self.columns}")
        if self._pending and (self._pending[0][0].shape != x.shape or self._pending[0][0].dtype != x.dtype
                              or self._pending[0][0].stride(0) != x.stride(0)):
            flush_pending()
        defer = max(1, int(HESSIAN_DEFER))
        staged = defer <= 1 and int(STAGE_INPUTS) > 1
        if staged:                       # an owned copy (STAGE_INPUTS): nothing of the caller's is referenced after the hook
            x = x.clone(memory_format=torch.contiguous_format)
            defer = int(STAGE_INPUTS)
        if defer > 1 and len(self._pending) >= defer:
            # this object is about to exceed the batch: its lock-step peers (the other Linears hooked in
            # the same forward passes) hold exactly as many inputs, so everything goes out grouped by shape
            if staged or not LAZY_HESSIANS:
                flush_pending()
            elif 2 * self.columns >= _widest_live():          # one of the widest Linears being calibrated
                flush_pending(heavy_only=True)
            elif len(self._pending) % 16 == 0 and _retained_bytes() > LAZY_MAX_BYTES:
                flush_pending()
        base = x._base if x._base is not None else x
        if defer > 1 and self._pending and x.data_ptr() in self._ptrs:
            # the same storage again while an update that reads it is still deferred: fine only if it is provably the
            # same, unmodified tensor (shared version counter); a `.data` alias or a refilled staging buffer is not
            seen_base, seen_ver = self._ptrs[x.data_ptr()]
            if seen_base is not base or seen_ver != x._version:
                raise RuntimeError(
                    "add_batch: this activation buffer is still referenced by a deferred Hessian update "
                    "(gptq_amd.gptq.HESSIAN_DEFER > 1 keeps references, not copies); set HESSIAN_DEFER = 1 for "
                    "callers that recycle one staging buffer per sample")
        self._sig = hash((self._sig if self._pending else 0, x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, batch))
        if not self._pending:
            self._ptrs = {}
        if defer > 1:                    # (the aliasing guard is only consulted while updates are deferred)
            self._ptrs[x.data_ptr()] = (base, x._version)
        self._pending.append((x, batch, x._version))
        _DIRTY[id(self)] = self
        self.nsamples += batch
        if defer <= 1:
            # like the reference: the update for `inp` is enqueued (current stream) before the hook returns, so a
            # later in-place write to the activation is ordered behind it
            if self._leader is not None or self._followers:
                flush_pending()
            else:
                self._fresh = False
                _launch_flush([self])
        if TRACK_INPUT_MEAN:   # fork addition (gptq.py:63), unused by the default branch
            self.input = x.mean(0, dtype=torch.float32) * math.sqrt(2 / self.nsamples)

    @_locked
    def fasterquant(self, blocksize=128, percdamp=.01, groupsize=-1, actorder=False, static_groups=False,
                    model_name="opt", layer_name="layer", lut_quant=False, non_linear_quant=False,
                    columnwise=False):
        """Damped inverse factor + blocked quantize-and-compensate loop (gptq.py:126-305).

        The fork-only branches (`lut_quant`, `non_linear_quant`, `columnwise`) are outside the
        MI355X hot-path scope and raise.
        """
        if lut_quant or non_linear_quant or columnwise:
            raise NotImplementedError("lut_quant / non_linear_quant / columnwise are fork experiments outside "
                                      "the MI355X hot-path scope")
        flush_pending()
        state = self._solve_enqueue(blocksize, percdamp, groupsize, actorder, static_groups)
        self._solve_finish(state)

    def _solve_enqueue(self, blocksize, percdamp, groupsize, actorder, static_groups):
        """Enqueue the whole solve on the CURRENT stream; nothing here waits for the GPU."""
        q = self.quantizer
        if int(q.maxq) < 0:
            raise NotImplementedError("trits are outside the MI355X hot-path scope")
        if self._pending:                # (fasterquant_many folds deferred inputs lane by lane)
            _launch_flush([self._leader or self])
        self._materialize()              # shared running Hessian: this object's own copy ...
        self._release_followers()        # ... and the copies of those that share this one's
        W = self.layer.weight.data.clone()
        if isinstance(self.layer, nn.Conv2d):
            W = W.flatten(1)
        if _Conv1D and isinstance(self.layer, _Conv1D):
            W = W.t()
        W = W.float().contiguous()
        self._applied = self.nsamples if self.nsamples else self._applied
        H = self._H                      # upper triangle is all the solver reads
        self._H = None                   # consumed, like `del self.H` (gptq.py:141-142)
        if H is None:
            raise RuntimeError("fasterquant: H was already consumed or freed")
        preset = (q.scale, q.zero) if (q.scale.numel() == W.shape[0] and bool(q.ready())) else None   # gptq.py:181
        return _enqueue_rows(self.dev, W, H, q, preset, blocksize, percdamp, groupsize, actorder, static_groups)

    def _solve_finish(self, st):
        """Wait for the solve and publish its results."""
        _publish_rows(self, st, 0, st["W"].shape[0], _check_solved(st))

    @_locked
    def free(self):
        if DEBUG:
            self.inp1 = None
            self.out1 = None
        self._release_followers()
        if self._leader is not None:
            self._leader._followers.remove(self)
            self._leader = None
        self._H = None
        self._pending = []
        self._ptrs = {}
        _DIRTY.pop(id(self), None)
        _LIVE.discard(self)
        self.Hinv = None
        self.codes = None
        self.Losses = None
        self.Trace = None
        if EMPTY_CACHE_ON_FREE:
            torch.cuda.empty_cache()
