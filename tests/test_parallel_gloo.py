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


# ---- first-run safety of the RCCL branch: what WOULD be handed to "nccl", checked with a fake process group --------
class _FakeDist:
    """Stands in for torch.distributed inside gptq_amd.parallel: claims the "nccl" backend, records every collective and
    simulates `world` identical ranks (all-reduce = x world, all-gather = world copies)."""

    def __init__(self, world, device):
        self.world, self.device, self.calls = world, device, []

    def get_backend(self, group=None):
        return "nccl"

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return 0

    def all_reduce(self, t, group=None, op=None):
        assert t.device.type == self.device and t.is_contiguous() and t.dtype == torch.float32
        self.calls.append(("all_reduce", t.numel(), t.dtype))
        t.mul_(self.world)

    def all_gather_into_tensor(self, out, inp, group=None):
        assert out.device == inp.device and out.device.type == self.device
        assert out.is_contiguous() and inp.is_contiguous() and out.dtype == inp.dtype == torch.int32
        assert out.numel() == self.world * inp.numel()
        self.calls.append(("all_gather_into_tensor", inp.numel(), inp.dtype))
        out.view(self.world, -1)[:] = inp

    def all_gather_object(self, out, obj, group=None):
        for i in range(len(out)):
            out[i] = obj


def test_nccl_branch_sizes_dtypes_and_guards(monkeypatch):
    fake = _FakeDist(8, "cpu")
    monkeypatch.setattr(par, "dist", fake)
    for C in (300, 1024):
        H = torch.triu(torch.randn(C, C))
        H0 = H.clone()
        n = par.allreduce_hessian(H, 5)
        # ONE collective of the trapezoid (+ the count) in fp32, on H's device; 8 identical ranks give back H itself
        assert fake.calls[-1] == ("all_reduce", par.tri_numel(C) + 1, torch.float32) and n == 40
        up = torch.triu(torch.ones(C, C, dtype=torch.bool))
        assert torch.allclose(H[up], H0[up], rtol=1e-6, atol=0)
    # an idle rank contributes zeros and count 0; NO rank with samples is an error on every rank, not a division by zero
    with pytest.raises(RuntimeError, match="n = 0"):
        par.allreduce_hessian(torch.zeros(256, 256), 0)
    # uneven sample counts: the weights n_r / n of the running means
    fake2 = _FakeDist(1, "cpu")
    monkeypatch.setattr(par, "dist", fake2)
    H = torch.triu(torch.ones(256, 256))
    assert par.allreduce_hessian(H, 3) == 3 and torch.equal(torch.triu(H), torch.triu(torch.ones(256, 256)))
    # the payload all-gather: int32, device to device, world * width
    monkeypatch.setattr(par, "dist", fake)
    send = torch.arange(1000, dtype=torch.int32)
    recv = par.allgather_payload(send)
    assert recv.shape == (8000,) and recv.dtype == torch.int32 and torch.equal(recv.view(8, -1)[7], send)
    assert fake.calls[-1] == ("all_gather_into_tensor", 1000, torch.int32)


class _Stub:
    """The sharing state of a GPTQ object, nothing else."""

    def __init__(self, nsamples):
        self.nsamples, self._leader, self._followers, self.copies = nsamples, None, [], 0

    def _materialize(self):
        if self._leader is not None:
            self._leader._followers.remove(self)
            self._leader = None
            self.copies += 1

    def _release_followers(self):
        for f in list(self._followers):
            f._materialize()


def _share(leader, *followers):
    for f in followers:
        f._leader = leader
        leader._followers.append(f)


def _bundle_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        # (a) rank 1 ran no sample (nsamples < world): it never formed q/k/v sharing; rank 0 did.  Both must end up with
        #     the SAME map -- rank 0's -- or they would issue 1 vs 3 all-reduces and hang.
        s = [_Stub(3 if rank == 0 else 0) for _ in range(4)]
        if rank == 0:
            _share(s[0], s[1], s[2])
        b = par._agree_bundles(s, None)
        out["idle"] = [[s.index(m) for m in part] for part in b]
        out["idle_state"] = (s[1]._leader is s[0], s[2]._leader is s[0], s[3]._leader is None, len(s[0]._followers))
        # (b) both ranks ran samples but disagree (one of them has sharing disabled): one bundle per solver everywhere
        s = [_Stub(2) for _ in range(3)]
        if rank == 1:
            _share(s[0], s[1])
        b = par._agree_bundles(s, None)
        out["disagree"] = [[s.index(m) for m in part] for part in b]
        out["disagree_state"] = all(x._leader is None and not x._followers for x in s)
        # (c) agreement: untouched
        s = [_Stub(2) for _ in range(3)]
        _share(s[1], s[2])
        b = par._agree_bundles(s, None)
        out["agree"] = [[s.index(m) for m in part] for part in b]
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bundle_map_is_rank_independent_world2_gloo():
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_bundle_worker, args=(2, port, ret), nprocs=2, join=True)
        r0, r1 = ret[0], ret[1]
    assert r0["idle"] == r1["idle"] == [[0, 1, 2], [3]]
    assert r0["idle_state"] == r1["idle_state"] == (True, True, True, 2)
    assert r0["disagree"] == r1["disagree"] == [[0], [1], [2]] and r0["disagree_state"] and r1["disagree_state"]
    assert r0["agree"] == r1["agree"] == [[0], [1, 2]]
