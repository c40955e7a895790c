"""The N>1 path on CPU: two gloo ranks shard a sweep, 'solve' their columns (here with the
oracle as a stand-in for the HIP solve, which needs a GPU) and gather to rank 0."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sosrt.dist import expected_orders, shard_indices


def test_shards_partition_the_columns():
    for n, w in ((512, 8), (13, 4), (5, 8), (64, 1)):
        cost = np.random.default_rng(n).uniform(1, 40, n)
        for c in (None, cost):
            parts = [shard_indices(n, w, r, c) for r in range(w)]
            assert sorted(np.concatenate(parts).tolist()) == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
        if n >= 4 * w:
            tot = [cost[shard_indices(n, w, r, cost)].sum() for r in range(w)]
            assert max(tot) / min(tot) < 1.25          # balanced total expected orders
    assert (np.diff(expected_orders([0.1, 0.5, 1.0], [0.2, 0.2, 0.2])) > 0).all()


def _worker(rank, world, port, q):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "sos-radiative-transfer_amd"), os.path.join(here, "..", "oracle")):
        sys.path.insert(0, p)
    import sos_oracle as O
    from sosrt.dist import gather_columns, shard_indices
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, N, B = 12, 32, 5                                # ragged: 3 + 2 columns
        mu = O.make_mu(N)
        P0, P = O.phase_isotropic(N, mu)
        mu0 = np.linspace(0.3, 0.9, B)
        taer = np.linspace(0.05, 0.4, B)
        mine = shard_indices(B, world, rank, expected_orders(0.1 + taer, np.full(B, 0.2)))
        I = np.stack([O.solve_column(O.make_column(mu0[b], 120, 60, 20, L, 0.1, taer[b], 0.2, 1.0, 1.0, N, P0, P, P0, P),
                                     literal=False).I for b in mine])
        local = {"toa_up": torch.from_numpy(I[:, 0, N:].copy()), "cols": torch.from_numpy(np.asarray(mine, dtype=np.int64))}
        res = gather_columns(local, mine, B, dst=0)
        if rank == 0:
            ref = np.stack([O.solve_column(O.make_column(mu0[b], 120, 60, 20, L, 0.1, taer[b], 0.2, 1.0, 1.0, N, P0, P, P0, P),
                                           literal=False).I[0, N:] for b in range(B)])
            ok = np.array_equal(res["toa_up"].numpy(), ref) and res["cols"].tolist() == list(range(B))
            q.put(bool(ok))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_gather_plan_is_the_same_on_every_rank_and_inverts():
    from sosrt.dist import GatherPlan
    cost = np.random.default_rng(3).uniform(1, 40, 37)
    p = GatherPlan(37, 4, cost)
    assert sum(p.counts) == 37 and sorted(p.order.tolist()) == list(range(37))
    assert np.array_equal(p.order[p.inverse], np.arange(37))
    assert all(np.array_equal(p.mine(r), shard_indices(37, 4, r, cost)) for r in range(4))
    rows = torch.arange(37, dtype=torch.float64)[torch.as_tensor(p.order)]          # what the root receives, rank-major
    assert torch.equal(p.restore(rows), torch.arange(37, dtype=torch.float64))
    q = GatherPlan(3, 8)                                                             # more ranks than columns
    assert q.counts.count(0) == 5 and sum(q.counts) == 3


def _plan_worker(rank, world, port, q):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "..", "sos-radiative-transfer_amd"))
    from sosrt.dist import GatherPlan, gather_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = 11
        cost = np.linspace(1, 3, B) ** 2
        plan = GatherPlan(B, world, cost)                  # every rank builds it locally: nothing is exchanged
        mine = plan.mine(rank)
        ok = True
        ptrs = set()
        for step in range(3):                              # the step loop: same plan, same receive buffers
            loc = torch.from_numpy(np.stack([np.full((2, 3), 100.0 * step + b) for b in mine]))
            loc_n = torch.as_tensor(mine, dtype=torch.int32)
            got = gather_rows(loc, plan, dst=0, key="I")
            got_n = gather_rows(loc_n, plan, dst=0, key="n")
            if rank == 0:
                ptrs.add(got.data_ptr())
                full = plan.restore(got)
                ok = ok and torch.equal(full[:, 0, 0], 100.0 * step + torch.arange(B, dtype=torch.float64))
                ok = ok and plan.restore(got_n).tolist() == list(range(B))
            else:
                ok = ok and got is None
        if rank == 0:
            q.put(bool(ok and len(ptrs) == 1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ragged_gather_through_a_plan(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + 7 * world) % 2000
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
