"""Host logic of the data-parallel path on CPU: the row plan, the trapezoid layout, and the Hessian all-reduce over
gloo (world 2).  The solve itself needs a GPU: tests/test_gpu_driver.py runs 2 ranks on the test box's one card."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gptq_amd import parallel as par


def test_lpt_assignment_balances_and_is_deterministic():
    costs = [1.0, 1.0, 1.0, 1.0, 4.0, 6.0] * 2
    a = par.assign_units(costs, 2)
    assert a == par.assign_units(costs, 2)
    assert sorted(i for r in a for i in r) == list(range(len(costs)))
    loads = [sum(costs[i] for i in r) for r in a]
    assert max(loads) / min(loads) < 1.05
    assert sorted(par.assign_units(costs, 1)[0]) == list(range(len(costs)))
    a8 = par.assign_units(costs[:3], 8)                     # more ranks than problems: the surplus ranks stay empty
    assert sum(len(r) for r in a8) == 3 and max(len(r) for r in a8) == 1
    assert par.assign_units([5.0, 3.0, 3.0, 2.0, 2.0], 2) == [[0, 3], [1, 2, 4]]
    # bundles travel together
    a = par.assign_units([1.0, 1.0, 1.0, 5.0], 2, bundles=[[0, 1, 2]], shared=[0.5])
    assert sorted(a) == [[0, 1, 2], [3]]


def test_tri_blocks_cover_the_upper_triangle_once():
    for C in (256, 300, 1024, 11008 // 8):
        blocks = par.tri_blocks(C)
        assert blocks[0][:2] == (0, min(256, C)) and blocks[-1][1] == C
        seen = torch.zeros(C, C, dtype=torch.int32)
        off_expect = 0
        for r0, r1, off in blocks:
            assert off == off_expect
            seen[r0:r1, r0:] += 1
            off_expect += (r1 - r0) * (C - r0)
        assert par.tri_numel(C) == off_expect
        up = torch.triu(torch.ones(C, C, dtype=torch.bool))
        assert bool((seen[up] == 1).all()) and int(seen.max()) == 1      # (+ the lower half of the diagonal tiles)
        assert int(seen.sum()) == off_expect


def test_plan_rows_llama_and_opt_shapes():
    # LLaMA true-sequential: one distinct Hessian per group -> every rank replicates the chain and takes rows/world
    plan = par.plan_rows([(8192, 8192)], 8)
    assert [(r, e - a) for (r, a, e) in plan[0]] == [(r, 1024) for r in range(8)]
    plan = par.plan_rows([(8192, 3 * 8192)], 8)              # q/k/v stacked
    assert sum(e - a for (_, a, e) in plan[0]) == 3 * 8192 and all(a % 128 == 0 for (_, a, _) in plan[0])
    assert sorted(r for (r, _, _) in plan[0]) == list(range(8))
    plan = par.plan_rows([(22016, 8192)], 3)
    assert [e - a for (_, a, e) in plan[0]] == [2816, 2688, 2688]
    # OPT block: four distinct Hessians; the widest (fc2) gets the spare ranks, everything is covered exactly once
    shapes = [(4096, 12288), (4096, 4096), (4096, 16384), (16384, 4096)]
    for world in (1, 2, 4, 8):
        plan = par.plan_rows(shapes, world)
        assert plan == par.plan_rows(shapes, world)
        for (C, R), slabs in zip(shapes, plan):
            assert slabs[0][1] == 0 and slabs[-1][2] == R
            assert all(s[2] == t[1] for s, t in zip(slabs, slabs[1:]))
            assert all(0 <= r < world for (r, _, _) in slabs)
        if world >= 4:
            assert len({r for slabs in plan for (r, _, _) in slabs}) == world      # nobody idles
    assert len(par.plan_rows(shapes, 8)[3]) == 5
    # tiny Linears are not cut below the alignment
    assert par.plan_rows([(128, 100)], 4) == [[(0, 0, 100)]]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for C, counts in ((300, (3, 5)), (512, (4, 4))):
            g = torch.Generator().manual_seed(7)
            xs = [torch.randn(64, C, generator=g, dtype=torch.float64) for _ in range(sum(counts))]
            # each rank: the reference's running mean over ITS samples (gptq.py:59-65), upper triangle only
            first = sum(counts[:rank])
            H, n = torch.zeros(C, C, dtype=torch.float64), 0
            for x in xs[first:first + counts[rank]]:
                H = H * (n / (n + 1)) + (2.0 / (n + 1)) * (x.t() @ x)
                n += 1
            Hl = torch.triu(H).float() + torch.tril(torch.full((C, C), 123.0), -1)     # junk below the diagonal stays
            n_all = par.allreduce_hessian(Hl, n)
            truth, m = torch.zeros(C, C, dtype=torch.float64), 0
            for x in xs:
                truth = truth * (m / (m + 1)) + (2.0 / (m + 1)) * (x.t() @ x)
                m += 1
            up = torch.triu(torch.ones(C, C, dtype=torch.bool))
            rel = float((Hl.double() - truth)[up].norm() / truth[up].norm())
            below = torch.tril(torch.ones(C, C, dtype=torch.bool), -256)             # under the diagonal TILES: untouched
            ok = ok and n_all == sum(counts) and rel < 1e-6 and bool((Hl[below] == 123.0).all())
            # bit-identical on every rank (the replicated chain relies on it)
            both = [torch.empty_like(Hl) for _ in range(world)]
            dist.all_gather(both, Hl)
            ok = ok and torch.equal(both[0], both[1])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_allreduce_hessian_world2_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
