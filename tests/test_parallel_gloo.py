"""Module-sharding host logic on CPU: LPT assignment + the packed-weight all-gather over gloo (world 2)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gptq_amd import parallel as par


def test_lpt_assignment_balances_and_is_deterministic():
    units = [par.Unit(n, r, c) for b in range(2) for (n, r, c) in
             [("q", 2048, 2048), ("k", 2048, 2048), ("v", 2048, 2048), ("o", 2048, 2048), ("fc1", 8192, 2048), ("fc2", 2048, 8192)]]
    costs = [par.unit_cost(u, 128, 2048) for u in units]
    a = par.assign_units(costs, 2)
    assert a == par.assign_units(costs, 2)
    assert sorted(i for r in a for i in r) == list(range(len(units)))
    loads = [sum(costs[i] for i in r) for r in a]
    assert max(loads) / min(loads) < 1.05
    one = par.assign_units(costs, 1)
    assert sorted(one[0]) == list(range(len(units)))
    # more ranks than units: the surplus ranks stay empty
    a8 = par.assign_units(costs[:3], 8)
    assert sum(len(r) for r in a8) == 3 and max(len(r) for r in a8) == 1


def test_cost_model_orders_the_opt_shapes():
    c_q = par.unit_cost(par.Unit("q", 2048, 2048), 128, 2048)
    c_fc1 = par.unit_cost(par.Unit("fc1", 8192, 2048), 128, 2048)
    c_fc2 = par.unit_cost(par.Unit("fc2", 2048, 8192), 128, 2048)
    assert c_q < c_fc1 < c_fc2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        units = [par.Unit("a", 64, 96), par.Unit("b", 32, 64), par.Unit("c", 128, 256), par.Unit("d", 16, 32)]
        costs = [par.unit_cost(u, 4, 16) for u in units]
        assignment = par.assign_units(costs, world)
        bits, gs = 4, 32

        def fake(i):
            (qh, qw), (gr, gc) = par.packed_shapes(units[i], bits, gs)
            g = torch.Generator().manual_seed(100 + i)
            q = torch.randint(-2 ** 31, 2 ** 31 - 1, (qh, qw), generator=g, dtype=torch.int64).to(torch.int32)
            return q, torch.rand(gr, gc, generator=g), torch.rand(gr, gc, generator=g)

        local = {i: fake(i) for i in assignment[rank]}
        everything = par.allgather_packed(local, units, assignment, bits, gs)
        ok = sorted(everything) == list(range(len(units)))
        for i in range(len(units)):
            q, s, z = fake(i)
            gq, gs_, gz = everything[i]
            ok = ok and torch.equal(gq, q) and torch.equal(gs_, s) and torch.equal(gz, z)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_allgather_packed_world2_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def test_assign_units_keeps_bundles_together():
    """Linears fed the same input travel together (their Hessian is accumulated once) and loads stay balanced."""
    from gptq_amd import parallel as par
    shapes = [("q", 2048, 2048), ("k", 2048, 2048), ("v", 2048, 2048), ("o", 2048, 2048), ("fc1", 8192, 2048),
              ("fc2", 2048, 8192)]
    for world in (1, 2, 3, 8):
        units = [par.Unit(f"b{b}.{n}", r, c) for b in range(world) for (n, r, c) in shapes]
        costs = [par.unit_cost(u, 128, 2048) for u in units]
        bundles = [[6 * b, 6 * b + 1, 6 * b + 2] for b in range(world)]
        a = par.assign_units(costs, world, bundles, [par.hessian_cost(units[m[0]], 128, 2048) for m in bundles])
        assert sorted(i for r in a for i in r) == list(range(len(units)))
        for m in bundles:
            assert len({next(r for r in range(world) if i in a[r]) for i in m}) == 1
        loads = [sum(costs[i] for i in r) for r in a]
        assert max(loads) <= 1.01 * min(loads)
    # without bundles: plain LPT, deterministic
    assert par.assign_units([5.0, 3.0, 3.0, 2.0, 2.0], 2) == [[0, 3], [1, 2, 4]]


def test_plan_units_row_slabs():
    """Row slabs (second axis): fewer Linears than ranks -> the costliest one is cut so nobody idles."""
    from gptq_amd import parallel as par
    units, costs, a = par.plan_units([("down_proj", 8192, 22016)], 8, 128, 2048)
    assert [(u.row0, u.rows, u.full_rows) for u in units] == [(1024 * j, 1024, 8192) for j in range(8)]
    assert sorted(i for r in a for i in r) == list(range(8)) and all(len(r) == 1 for r in a)
    units, _, a = par.plan_units([("o_proj", 4096, 4096)], 3, 128, 2048)
    assert [u.rows for u in units] == [1408, 1408, 1280] and sum(u.rows for u in units) == 4096
    assert all(u.row0 % 128 == 0 for u in units)
    units, _, a = par.plan_units([("q", 256, 128), ("fc1", 512, 128)], 2, 8, 128, row_slabs=0)
    assert all(u.full_rows == 0 for u in units)
    units, _, a = par.plan_units([("q", 128, 128), ("fc1", 512, 128)], 2, 8, 128, row_slabs=2)
    assert [(u.name, u.row0, u.rows) for u in units] == [("q", 0, 128), ("fc1", 0, 256), ("fc1", 256, 256)]
    qshape, gshape = par.packed_shapes(units[1], 4, 32)
    assert qshape == (128 // 32 * 4, 256) and gshape == (256, 4)
