"""Sharding one transformer block's GPTQ work across the GPUs of a node.

The reference has no distributed code; this is the capability BASELINE
config 5 / north_star adds.  One process per GPU, torch.distributed ("nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU in tests).  Two layers:

  * `fasterquant_sharded` -- the data-parallel path: every rank folds ITS share of the
    calibration samples into the Hessians (add_batch), one all-reduce of H per distinct Hessian, the
    factorization chain replicated on the ranks that need it, the ROWS of W split over the ranks (rows are
    independent problems given H, gptq.py:262-276), one all-gather of the packed rows + grids.  Every rank ends
    up with bit-identical quantized weights for the post-quantization forward (opt.py:216-217).
  * `assign_units` -- longest-processing-time-first dealing of whole problems to ranks (used when there are more
    distinct Hessians than ranks: Linears hooked in one forward pass are independent problems, opt.py:189-214).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def assign_units(costs: Sequence[float], world: int, bundles: Sequence[Sequence[int]] = (),
                 shared: Sequence[float] = ()) -> List[List[int]]:
    """Longest-processing-time-first: returns, per rank, the indices of its units (deterministic).
    `bundles` lists groups of units that must land on the same rank (Linears fed the same input tensor: their
    Hessian is accumulated once); `shared[b]` is the cost a bundle saves per member beyond the first."""
    items: List[Tuple[float, List[int]]] = []
    bundled = set()
    for b, members in enumerate(bundles):
        members = list(members)
        if not members:
            continue
        save = (shared[b] if b < len(shared) else 0.0) * (len(members) - 1)
        items.append((sum(costs[i] for i in members) - save, members))
        bundled.update(members)
    items += [(costs[i], [i]) for i in range(len(costs)) if i not in bundled]
    order = sorted(range(len(items)), key=lambda k: (-items[k][0], items[k][1][0]))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for k in order:
        r = min(range(world), key=lambda j: (load[j], j))
        out[r] += items[k][1]
        load[r] += items[k][0]
    return [sorted(a) for a in out]


# ------------------------------------------------------------------------------------------------
# Data-parallel solve: all-reduce of H, row-sharded column loop, all-gather of the packed rows
# ------------------------------------------------------------------------------------------------
def tri_blocks(C: int, bs: int = 256) -> List[Tuple[int, int, int]]:
    """(r0, r1, offset) of the row blocks of the upper trapezoid layout: rows [r0, r1) x columns [r0, C), packed one
    after the other.  add_batch maintains only the upper triangle of H, so this is all an exchange has to carry:
    C^2/2 + bs*C/2 elements instead of C^2."""
    out, off = [], 0
    for r0 in range(0, C, bs):
        r1 = min(C, r0 + bs)
        out.append((r0, r1, off))
        off += (r1 - r0) * (C - r0)
    return out


def tri_numel(C: int, bs: int = 256) -> int:
    r0, r1, off = tri_blocks(C, bs)[-1]
    return off + (r1 - r0) * (C - r0)


def allreduce_hessian(H: torch.Tensor, n_local: int, group=None) -> int:
    """H <- sum_r (n_r / n) H_r over the ranks (in place, upper triangle only), n = sum_r n_r; returns n.
    H_r is the reference's running mean over the n_r samples rank r folded in (gptq.py:59-65): the weighted sum is
    the running mean over all n samples up to fp32 rounding.  One collective: the trapezoid-packed upper triangle,
    pre-scaled by n_r, with n_r itself in the last element."""
    C = H.shape[0]
    blocks = tri_blocks(C)
    numel = tri_numel(C)
    flat = torch.empty(numel + 1, device=H.device, dtype=torch.float32)
    for r0, r1, off in blocks:
        torch.mul(H[r0:r1, r0:], float(n_local), out=flat[off:off + (r1 - r0) * (C - r0)].view(r1 - r0, C - r0))
    flat[numel] = float(n_local)
    if flat.is_cuda and dist.get_backend(group) != "nccl":      # rehearsal backends (gloo): stage through the host
        host = flat.cpu()
        dist.all_reduce(host, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, group=group)
    n = int(round(float(flat[numel].item())))
    if n <= 0:
        raise RuntimeError("allreduce_hessian: no rank has folded a calibration sample into this Hessian (n = 0)")
    for r0, r1, off in blocks:
        torch.mul(flat[off:off + (r1 - r0) * (C - r0)].view(r1 - r0, C - r0), 1.0 / n, out=H[r0:r1, r0:])
    return n


def allgather_payload(send: torch.Tensor, group=None) -> torch.Tensor:
    """ONE fixed-size all-gather of every rank's int32 payload (packed rows, grids, row losses, permutation, pivot
    flag): returns [world * send.numel()] on send's device.  "nccl" (= RCCL) gathers device to device; the rehearsal
    backends (gloo) stage through the host."""
    world = dist.get_world_size(group)
    if send.is_cuda and dist.get_backend(group) != "nccl":
        host = torch.empty(world * send.numel(), dtype=send.dtype)
        dist.all_gather_into_tensor(host, send.cpu(), group=group)
        return host.to(send.device)
    recv = torch.empty(world * send.numel(), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    return recv


# ------------------------------------------------------------------------------------------------
# Factorization spread over the ranks of a bundle: outer panels dealt cyclically, one broadcast per panel
# ------------------------------------------------------------------------------------------------
# A bundle's factorization chain depends on H only, so fasterquant_sharded used to REPLICATE it on every rank that owns
# rows of the bundle -- for LLaMA-65B's down_proj (C = 22016) that is 43 of its 48 ms, of which 39 ms are the rank-512
# updates (C^3 / 3 flop).  With SHARD_CHOL the outer panels (512 columns) of the factorization are dealt to the bundle's
# ranks cyclically: the owner factorizes its panel (gptq_chol_panel), broadcasts it (rows below x 512 floats + four
# 128 x 128 inverses: C^2 / 2 floats per factorization in all), every rank updates the block columns of ITS panels
# (gptq_chol_update).  The serial chain (4 diagonal blocks per panel) stays serial; the GEMM flops divide by the ranks.
# Same kernels and the same ascending k order as gptq_rfactor_upper: the factor is bit-identical to the single-rank one.
SHARD_CHOL = __import__("os").environ.get("GPTQ_SHARD_CHOL", "1") != "0"
SHARD_CHOL_MIN_C = 8192          # below, the rank-512 updates are a fraction of a millisecond (C = 4096: 0.24 of the chain's 2.1 ms)
_SUBGROUPS = {}


def _bcast(t: torch.Tensor, src_global: int, group=None) -> None:
    if t.is_cuda and dist.get_backend(group) != "nccl":         # rehearsal backends (gloo): stage through the host
        host = t.cpu()
        dist.broadcast(host, src_global, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src_global, group=group)


def _subgroup(ranks: Sequence[int], group=None):
    """Process group of `ranks` (ranks OF `group`).  EVERY rank of `group` must call this, in the same order (new_group
    is collective); cached."""
    world = dist.get_world_size(group)
    if len(ranks) == world:
        return group
    glob = tuple(dist.get_global_rank(group, r) if group is not None else r for r in ranks)
    if glob not in _SUBGROUPS:
        _SUBGROUPS[glob] = dist.new_group(ranks=list(glob))
    return _SUBGROUPS[glob]


def rfactor_sharded(H: torch.Tensor, perm, percdamp: float, ranks: Sequence[int], group=None, sub=None) -> torch.Tensor:
    """H [C, C] (dead-column fix applied, identical on all `ranks` of `group`) <- what gptq_rfactor_upper leaves in it,
    the outer panels of the factorization dealt over `ranks` (this rank among them).  `sub`: the process group of exactly
    `ranks` (`_subgroup`).  Returns the device int32[1] pivot flag (non-zero: not positive-definite)."""
    from . import _lib
    dev = H.device
    C = H.shape[0]
    assert C % 128 == 0 and H.stride(0) >= C
    nblk = C // 128
    lib = _lib.load()
    nbytes = lib.gptq_hinv_workspace_bytes(C)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    A = ws[:C * C * 4].view(torch.float32).view(C, C)
    Linv = ws[C * C * 4:2 * C * C * 4].view(torch.float32).view(C, C)
    info = torch.zeros(1, device=dev, dtype=torch.int32)
    st = _lib.stream(dev)
    me = dist.get_rank(group) if len(ranks) > 1 else ranks[0]
    k = len(ranks)
    with torch.cuda.device(dev):
        _lib.call("gptq_chol_begin", _lib.ptr(H), H.stride(0), C, float(percdamp), _lib.ptr(perm), _lib.ptr(info),
                  _lib.ptr(ws), nbytes, st)
        npan = -(-nblk // 4)
        for j in range(npan):
            p0, p1 = 4 * j, min(4 * j + 4, nblk)
            owner = ranks[j % k]
            if owner == me:
                _lib.call("gptq_chol_panel", _lib.ptr(ws), C, p0, _lib.ptr(info), st)
            if k > 1:
                src = dist.get_global_rank(group, owner) if group is not None else owner
                panel = A[p0 * 128:, p0 * 128:p1 * 128]
                inv = [Linv[b * 128:(b + 1) * 128, b * 128:(b + 1) * 128] for b in range(p0, p1)]
                if owner == me:
                    buf = torch.cat([panel.reshape(-1)] + [x.reshape(-1) for x in inv])
                else:
                    buf = torch.empty(panel.numel() + sum(x.numel() for x in inv), device=dev, dtype=torch.float32)
                _bcast(buf, src, sub)
                if owner != me:
                    panel.copy_(buf[:panel.numel()].view(panel.shape))
                    off = panel.numel()
                    for x in inv:
                        x.copy_(buf[off:off + x.numel()].view(128, 128))
                        off += x.numel()
            for j2 in range(j + 1, npan):                       # the block columns of MY outer panels
                if ranks[j2 % k] == me:
                    _lib.call("gptq_chol_update", _lib.ptr(ws), C, p0, 4 * j2, min(4 * j2 + 4, nblk), st)
        _lib.call("gptq_chol_end", _lib.ptr(H), H.stride(0), C, _lib.ptr(ws), st)
    if k > 1:                                                   # a non-positive pivot on any rank is everybody's
        flag = info.clone()
        if flag.is_cuda and dist.get_backend(sub) != "nccl":
            host = flag.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=sub)
            flag.copy_(host)
        else:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=sub)
        info = flag
    return info


def plan_rows(bundles: Sequence[Tuple[int, int]], world: int, align: int = 128) -> List[List[Tuple[int, int, int]]]:
    """bundles: (C, R) per distinct Hessian (R = stacked rows of the Linears that share it).  Returns, per bundle, the
    slabs (rank, row_a, row_b) that cover its rows.  A bundle's ranks all replicate its factorization chain (it depends
    on H only) and split its rows; with fewer ranks than bundles whole bundles are dealt by cost.  Deterministic."""
    def chain(C):                       # seconds, roughly: latency-bound diagonal chain + 2/3 C^3 at ~80 TFLOP/s
        return 1.0e-4 * (C / 128.0) + (2.0 / 3.0) * C ** 3 / 8e13
    def loop_fixed(C):                  # the column loop's per-block latency does not shrink with fewer rows
        return 6.5e-5 * (C / 128.0)
    def rows_cost(C, R):
        return R * float(C) ** 2 / 6e13
    nb = len(bundles)
    if nb == 0:
        return []
    if world < nb:
        cost = [chain(C) + loop_fixed(C) + rows_cost(C, R) for C, R in bundles]
        owner = assign_units(cost, world)
        out: List[List[Tuple[int, int, int]]] = [[] for _ in bundles]
        for r, idxs in enumerate(owner):
            for b in idxs:
                out[b] = [(r, 0, bundles[b][1])]
        return out
    k = [1] * nb
    def t(b):
        C, R = bundles[b]
        return chain(C) + loop_fixed(C) + rows_cost(C, R) / k[b]
    for _ in range(world - nb):
        cand = [b for b in range(nb) if k[b] * align < bundles[b][1]]
        if not cand:
            break
        b = max(cand, key=lambda j: (t(j), -j))
        k[b] += 1
    out, nxt = [], 0
    for b, (C, R) in enumerate(bundles):
        blocks = -(-R // align)
        kk = min(k[b], blocks)
        slabs, a = [], 0
        for j in range(kk):
            nbk = blocks // kk + (1 if j < blocks % kk else 0)
            e = min(R, a + nbk * align)
            slabs.append((nxt % world, a, e))
            nxt += 1
            a = e
        out.append(slabs)
    return out


def fasterquant_sharded(solvers, bits: int, group=None, blocksize: int = 128, percdamp: float = .01,
                        groupsize: int = -1, actorder: bool = False, static_groups: bool = False, timings=None):
    """`fasterquant` of the Linears hooked in one forward pass, data-parallel over the ranks of `group`.

    Every rank calls this with GPTQ objects for the SAME Linears (same order, same weights), each holding the Hessian
    of the calibration samples THIS rank ran (add_batch).  Steps: (1) one all-reduce per distinct Hessian (Linears fed
    the same tensor share one, gptq.SHARE_INPUT_HESSIANS); (2) `plan_rows`: every distinct Hessian's stacked rows are
    split over ranks; (3) each rank runs gptq_fasterquant_rows on its slabs (the chain is replicated: it depends on
    H only, which is bit-identical on all ranks after the all-reduce); (4) one fixed-size all-gather of the packed
    rows, grids and per-row losses; (5) every rank rebuilds ALL the Linears from the packed form
    (gptq_dequant_packed = the solver's own scale * (code - zero)), so the weights are bit-identical everywhere.
    Publishes on every solver what `fasterquant` does (layer.weight, quantizer.scale / zero, error, perm, group
    tables) plus `qweight`; returns [(qweight, scale_table [R, G], zero_table [R, G])] per solver."""
    from . import gptq as gmod
    from .quant import dequant_packed, pack_codes
    if bits not in (3, 4):
        raise NotImplementedError("sharded runs exchange packed weights: bits must be 3 or 4")
    if groupsize > 0 and actorder and not static_groups:
        raise NotImplementedError("sharded runs need static groups with act-order (no g_idx in the packed format)")
    solvers = list(solvers)
    for s in solvers:                            # rows are sliced and written back as nn.Linear's [out, in]
        if not isinstance(s.layer, torch.nn.Linear):
            raise NotImplementedError(f"fasterquant_sharded: only nn.Linear layers are sharded (got {type(s.layer).__name__})")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = solvers[0].dev
    ev = lambda: torch.cuda.Event(enable_timing=True)
    t0, t1, t2, t3 = ev(), ev(), ev(), ev()
    # ---- distinct Hessians: a leader and the solvers that share its H ---------------------------------
    gmod.flush_pending()
    bundles = _agree_bundles(solvers, group)
    C_of = [b[0].columns for b in bundles]
    R_of = [sum(m.rows for m in b) for b in bundles]
    # ---- (1) all-reduce of H ---------------------------------------------------------------------------
    t0.record()
    for b in bundles:
        L = b[0]
        n = allreduce_hessian(L._H, L.nsamples, group)
        for m in b:
            m.nsamples = n
            m._applied = n
        L._lower_stale = True
    t1.record()
    # ---- (2) plan, (3) solve my slabs on up to SOLVE_STREAMS lanes -------------------------------------
    plan = plan_rows(list(zip(C_of, R_of)), world)
    cur = torch.cuda.current_stream(dev)
    mine = [(bi, a, e) for bi, slabs in enumerate(plan) for (r, a, e) in slabs if r == rank]
    pool = gmod._SOLVE_STREAMS.setdefault(dev, [])
    want = max(1, min(int(gmod.SOLVE_STREAMS), len(mine)))
    while len(pool) < want - 1:
        pool.append(torch.cuda.Stream(device=dev))
    lanes = [cur] + pool[:want - 1]
    for st in lanes[1:]:
        st.wait_stream(cur)
    states = {}
    # bundles whose rows are split over several ranks: the factorization is split over those ranks too (rfactor_sharded).
    # new_group is collective: EVERY rank walks the bundles in the same order, members or not.
    lib = gmod._lib.load()
    shard = {}
    for bi, slabs in enumerate(plan):
        rk = [r for (r, _, _) in slabs]
        ok = (SHARD_CHOL and len(rk) > 1 and C_of[bi] >= SHARD_CHOL_MIN_C and
              lib.gptq_fasterquant_factor_form(C_of[bi], int(blocksize), int(groupsize), int(bool(static_groups))) == 1)
        if ok:
            shard[bi] = (rk, _subgroup(rk, group))
    for (bi, a, e) in mine:
        if bi not in shard:
            continue
        rk, sub = shard[bi]
        b = bundles[bi]
        L = b[0]
        H = L._H
        Cb = C_of[bi]
        dead = torch.empty(Cb, device=dev, dtype=torch.int32)
        perm = torch.empty(Cb, device=dev, dtype=torch.int32) if actorder else None
        scratch = torch.empty(Cb, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            gmod._lib.call("gptq_solve_prepare", gmod._lib.ptr(H), H.stride(0), Cb, int(bool(actorder)), gmod._lib.ptr(dead),
                           gmod._lib.ptr(perm), gmod._lib.ptr(scratch), gmod._lib.stream(dev))
        info = rfactor_sharded(H, perm, percdamp, rk, group, sub)
        W = _stacked_rows(b, a, e)
        states[(bi, a, e)] = gmod._enqueue_rows(dev, W, H, L.quantizer, None, blocksize, percdamp, groupsize, actorder,
                                                static_groups, factored=dict(dead=dead, perm=perm, info=info))
    order = sorted([t for t in mine if t[0] not in shard], key=lambda t: -(C_of[t[0]] ** 3 + (t[2] - t[1]) * C_of[t[0]] ** 2))
    for k, (bi, a, e) in enumerate(order):
        b = bundles[bi]
        L = b[0]
        H = L._H
        st = lanes[k % want]
        if st is not cur:
            H.record_stream(st)
        with torch.cuda.stream(st):
            # the slab is gathered ON the lane that solves it: gathered on the caller's stream after the lanes' wait_stream
            # above, a side lane could start its solve before the copy had run (seen as a wrong `error` of one bundle
            # once nothing else synchronised the device between blocks)
            W = _stacked_rows(b, a, e)
            states[(bi, a, e)] = gmod._enqueue_rows(dev, W, H, L.quantizer, None, blocksize, percdamp, groupsize,
                                                    actorder, static_groups)
    for st in lanes[1:]:
        cur.wait_stream(st)
    for b in bundles:
        for m in b:
            m._H = None                          # consumed (or never needed on this rank)
            if m._leader is not None:
                m._leader = None
        b[0]._followers = []
    # ---- (4) all-gather of packed rows + grids + row losses ---------------------------------------------
    def slab_numel(bi, a, e):
        Cb = C_of[bi]
        Gb = -(-Cb // groupsize) if groupsize > 0 else 1
        return (Cb // 32 * bits) * (e - a) + (2 * Gb + 1) * (e - a) + (Cb if actorder else 0)
    per_rank = [[(bi, a, e) for bi, slabs in enumerate(plan) for (r, a, e) in slabs if r == rr] for rr in range(world)]
    width = 1 + max(sum(slab_numel(*t) for t in lst) for lst in per_rank)   # last element: first bad pivot of the rank's slabs
    send = torch.zeros(width, dtype=torch.int32, device=dev)
    off = 0
    for key in per_rank[rank]:
        st = states[key]
        # a non-positive-definite Hessian must raise on EVERY rank (the others would block in the all-gather): the
        # pivot index travels in the payload and is checked after the exchange
        bad = st["stat"][1:].view(torch.int32)
        send[width - 1:] = torch.where(send[width - 1:] != 0, send[width - 1:], bad)
        bi, a, e = key
        if groupsize > 0:
            stab, ztab = st["gscale"], st["gzero"]
        else:
            stab, ztab = st["scale"].reshape(-1, 1), st["zero"].reshape(-1, 1)
        parts = [pack_codes(st["codes"], bits).reshape(-1), stab.reshape(-1).view(torch.int32),
                 ztab.reshape(-1).view(torch.int32), st["row_loss"].view(torch.int32)]
        if actorder:
            parts.append(st["perm"])
        for t in parts:
            send[off:off + t.numel()] = t
            off += t.numel()
    t2.record()
    recv = allgather_payload(send, group)
    bad = recv.view(world, width)[:, -1]
    if bool((bad != 0).any().item()):
        raise torch.linalg.LinAlgError(
            f"fasterquant: the damped Hessian is not positive-definite (pivot {int(bad[bad != 0][0].item())}); "
            "cf. torch.linalg.cholesky")
    # ---- (5) rebuild every Linear from the packed form --------------------------------------------------
    full = []
    for bi, b in enumerate(bundles):
        Cb, Rb = C_of[bi], R_of[bi]
        Gb = -(-Cb // groupsize) if groupsize > 0 else 1
        full.append(dict(qw=torch.empty((Cb // 32 * bits, Rb), dtype=torch.int32, device=dev),
                         s=torch.empty((Rb, Gb), device=dev), z=torch.empty((Rb, Gb), device=dev),
                         loss=torch.empty(Rb, device=dev), perm=None))
    for rr in range(world):
        off = rr * width
        for (bi, a, e) in per_rank[rr]:
            Cb = C_of[bi]
            Gb = -(-Cb // groupsize) if groupsize > 0 else 1
            n = e - a
            f = full[bi]
            qh = Cb // 32 * bits
            f["qw"][:, a:e] = recv[off:off + qh * n].view(qh, n); off += qh * n
            f["s"][a:e] = recv[off:off + n * Gb].view(torch.float32).view(n, Gb); off += n * Gb
            f["z"][a:e] = recv[off:off + n * Gb].view(torch.float32).view(n, Gb); off += n * Gb
            f["loss"][a:e] = recv[off:off + n].view(torch.float32); off += n
            if actorder:
                f["perm"] = recv[off:off + Cb].clone(); off += Cb
    results = {}
    for bi, b in enumerate(bundles):
        f = full[bi]
        r0 = 0
        for m in b:
            r1 = r0 + m.rows
            qw = f["qw"][:, r0:r1].contiguous()
            stab, ztab = f["s"][r0:r1].contiguous(), f["z"][r0:r1].contiguous()
            lin = m.layer
            W = dequant_packed(qw, stab.t().contiguous(), ztab.t().contiguous(), bits, groupsize, dtype=lin.weight.dtype)
            lin.weight.data = W.reshape(lin.weight.shape)
            q = m.quantizer
            q.maxq = q.maxq.to(dev)
            q.scale = stab[:, -1:].clone()
            q.zero = ztab[:, -1:].clone()
            m.error = float(f["loss"][r0:r1].sum().item())
            m.qweight = qw
            m.codes = None
            m.group_scale = stab if groupsize > 0 else None
            m.group_zero = ztab if groupsize > 0 else None
            m.perm = f["perm"]
            m.static_groups = bool(static_groups)
            m.Hinv = None
            results[id(m)] = (qw, stab, ztab)
            r0 = r1
    t3.record()
    if timings is not None:
        torch.cuda.synchronize(dev)
        timings["exchange"] = timings.get("exchange", 0.0) + t0.elapsed_time(t1) + t2.elapsed_time(t3)
    return [results[id(s)] for s in solvers]


def _local_partition(solvers) -> List[List[int]]:
    """Which of `solvers` (by position) share one running Hessian on THIS rank, leader first; solvers whose leader is
    not part of the call stand alone.  Sorted by first member: a canonical form that ranks can compare."""
    pos = {id(s): i for i, s in enumerate(solvers)}
    parts, seen = [], set()
    for s in solvers:
        L = s._leader or s
        if id(L) in seen:
            continue
        seen.add(id(L))
        members = [m for m in [L] + list(L._followers) if id(m) in pos]
        if id(L) not in pos:
            parts += [[pos[id(m)]] for m in members]
            seen.update(id(m) for m in members)
        else:
            parts.append([pos[id(m)] for m in members])
    return sorted(parts, key=lambda p: p[0])


def _agree_bundles(solvers, group):
    """The bundle map (number and size of the all-reduces, layout of the all-gather) must be the SAME on every rank, but
    who shares a Hessian is local state: a rank that folded no sample (nsamples < world) never formed q/k/v sharing.  So
    the ranks exchange their partitions; ranks without samples abstain; if the others agree, everybody adopts that map
    (a rank without samples holds all-zero Hessians: merging them is exact), otherwise everybody falls back to one bundle
    per solver.  Returns the bundles as lists of solver objects, leader first."""
    mine = _local_partition(solvers)
    n_local = max((int(s.nsamples) for s in solvers), default=0)
    world = dist.get_world_size(group)
    views = [None] * world
    dist.all_gather_object(views, (n_local, mine), group=group)
    voters = [p for n, p in views if n > 0]
    agreed = voters[0] if voters and all(p == voters[0] for p in voters) else [[i] for i in range(len(solvers))]
    if agreed != mine:
        for s in solvers:                        # undo local sharing, then rebuild the agreed map
            s._materialize()
            s._release_followers()
        for part in agreed:
            L = solvers[part[0]]
            for i in part[1:]:
                f = solvers[i]
                if n_local > 0:                  # (only reachable in the fallback: parts are singletons there)
                    raise AssertionError("bundle map disagreement with local samples")
                f._leader = L
                L._followers.append(f)
    else:
        chosen = {id(s) for s in solvers}
        for part in agreed:                      # followers outside this call take their own copies
            L = solvers[part[0]]
            for f in list(L._followers):
                if id(f) not in chosen:
                    f._materialize()
            if L._leader is not None:
                L._materialize()
    return [[solvers[i] for i in part] for part in agreed]


def _stacked_rows(members, a: int, e: int) -> torch.Tensor:
    """Rows [a, e) of the Linears `members` stacked on top of each other, fp32 contiguous (a fresh tensor)."""
    parts, r0 = [], 0
    for m in members:
        r1 = r0 + m.rows
        lo, hi = max(a, r0), min(e, r1)
        if lo < hi:
            parts.append(m.layer.weight.data[lo - r0:hi - r0].float())
        r0 = r1
    return (torch.cat(parts, 0) if len(parts) > 1 else parts[0].clone()).contiguous()
