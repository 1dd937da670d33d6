"""Module-level sharding of one transformer block's Linears across the GPUs of a node.

The reference has no distributed code; this is the capability BASELINE
config 5 / north_star adds: the Linears hooked in one forward pass are independent GPTQ
problems (opt.py:189-214 walks them serially only for convenience), so they are dealt to
ranks by cost and the only exchange is an all-gather of the PACKED weights (+ grids) at the
block boundary -- every rank needs them to run the post-quantization forward (opt.py:216-217).
One process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm; "gloo" on CPU in tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Unit:
    """One Linear to quantize -- or a ROW SLAB of one: `name`, rows R (of the slab), in_features C, and the slab's
    first row in the Linear (`row0`; `full_rows` = the Linear's out_features, 0 when the unit is the whole Linear).
    Rows of W are independent problems given H (per-row grids, per-row error feedback, gptq.py:262-276), so a slab
    is solved exactly like a Linear with fewer outputs; its owner needs the full Hessian."""
    name: str
    rows: int
    cols: int
    row0: int = 0
    full_rows: int = 0

    @property
    def params(self) -> int:
        return self.rows * self.cols


def unit_cost(u: Unit, nsamples: int, seqlen: int, blocksize: int = 128) -> float:
    """fp32 flop model of the path for one Linear: Hessian SYRK (upper half)
    + factorization chain (2/3 C^3 here) + trailing updates (R C^2) + in-block rank-1 updates."""
    R, C = u.rows, u.cols
    return nsamples * seqlen * C * C + (2.0 / 3.0) * C ** 3 + R * C * C + R * C * blocksize


def split_rows(u: Unit, k: int, align: int = 128) -> List[Unit]:
    """`u` as k row slabs of (nearly) equal height, boundaries on multiples of `align` (fewer slabs if R is small)."""
    blocks = max(1, -(-u.rows // align))
    k = max(1, min(k, blocks))
    out, start = [], 0
    for j in range(k):
        nb = blocks // k + (1 if j < blocks % k else 0)
        stop = min(u.rows, start + nb * align)
        out.append(Unit(u.name, stop - start, u.cols, u.row0 + start, u.full_rows or u.rows))
        start = stop
    return out


def plan_units(shapes: Sequence[Tuple[str, int, int]], world: int, nsamples: int, seqlen: int, blocksize: int = 128,
               row_slabs: int = 1) -> Tuple[List[Unit], List[float], List[List[int]]]:
    """Units (Linears or row slabs), their costs and the per-rank assignment for the Linears `shapes` = (name, R, C)
    hooked in one forward pass.  row_slabs: 0 = whole Linears only; 1 = split the costliest Linear into as many slabs
    as ranks would otherwise idle (fewer Linears than ranks: LLaMA's true-sequential [o] and [down] groups);
    k >= 2 = split every Linear into k slabs.  A slab owner accumulates the full Hessian and runs the full
    factorization chain (they depend on H only), so slabs shard only the row-proportional work: the trailing
    updates R*C^2 and the column loop."""
    units = [Unit(n, r, c) for (n, r, c) in shapes]
    cost = lambda u: unit_cost(u, nsamples, seqlen, blocksize)
    if row_slabs >= 2:
        units = [s for u in units for s in split_rows(u, row_slabs)]
    elif row_slabs == 1 and world > len(units):
        big = max(range(len(units)), key=lambda i: (cost(units[i]), -i))
        units = units[:big] + split_rows(units[big], world - len(units) + 1) + units[big + 1:]
    costs = [cost(u) for u in units]
    return units, costs, assign_units(costs, world)


def hessian_cost(u: Unit, nsamples: int, seqlen: int) -> float:
    """The Hessian term of `unit_cost` (Linears fed the same input share it, gptq.SHARE_INPUT_HESSIANS)."""
    return float(nsamples) * seqlen * u.cols * u.cols


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


def packed_shapes(u: Unit, bits: int, groupsize: int) -> Tuple[Tuple[int, int], Tuple[int, int]]:
    """(qweight shape, grid-table shape) of a packed Linear."""
    groups = 1 if groupsize <= 0 else -(-u.cols // groupsize)
    return (u.cols // 32 * bits, u.rows), (u.rows, groups)


def allgather_packed(local: Dict[int, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], units: Sequence[Unit],
                     assignment: Sequence[Sequence[int]], bits: int, groupsize: int,
                     group=None, device=None) -> Dict[int, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
    """Every rank contributes (qweight int32, scales fp32, zeros fp32) for the units it owns and
    receives everybody's.  One fixed-size all-gather: each rank's tensors are flattened into a
    single int32 buffer padded to the largest rank payload (shapes are known to all ranks from
    `assignment`, so no size exchange is needed).  `device`: where the exchange buffers live -- the caller's GPU;
    it must be given when a rank may own nothing (an idle rank has no tensor to infer it from, and RCCL cannot
    gather host tensors)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)

    def payload(idx_list):
        n = 0
        for i in idx_list:
            (qh, qw), (gr, gc) = packed_shapes(units[i], bits, groupsize)
            n += qh * qw + 2 * gr * gc
        return n

    sizes = [payload(a) for a in assignment]
    width = max(max(sizes), 1)
    if device is None:
        some = next(iter(local.values()))[0] if local else None
        if some is None and dist.get_backend(group) == "nccl":
            raise ValueError("allgather_packed: this rank owns no unit, pass device= (RCCL needs device buffers)")
        device = some.device if some is not None else torch.device("cpu")
    device = torch.device(device)
    send = torch.zeros(width, dtype=torch.int32, device=device)
    off = 0
    for i in assignment[rank]:
        q, s, z = local[i]
        for t in (q.reshape(-1), s.reshape(-1).float().view(torch.int32), z.reshape(-1).float().view(torch.int32)):
            send[off:off + t.numel()] = t
            off += t.numel()
    if device.type == "cuda" and dist.get_backend(group) != "nccl":
        # rehearsal backends (gloo): stage through host memory
        host = torch.empty(world * width, dtype=torch.int32)
        dist.all_gather_into_tensor(host, send.cpu(), group=group)
        recv = host.to(device)
    else:
        recv = torch.empty(world * width, dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(recv, send, group=group)
    out: Dict[int, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}
    for r in range(world):
        off = r * width
        for i in assignment[r]:
            (qh, qw), (gr, gc) = packed_shapes(units[i], bits, groupsize)
            q = recv[off:off + qh * qw].reshape(qh, qw); off += qh * qw
            s = recv[off:off + gr * gc].view(torch.float32).reshape(gr, gc); off += gr * gc
            z = recv[off:off + gr * gc].view(torch.float32).reshape(gr, gc); off += gr * gc
            out[i] = (q, s, z)
    return out
