"""Multi-GPU: independent columns are dealt to the ranks (one process per GPU), every rank
solves its shard with no communication, and the results are gathered once at the end
(RCCL gather over xGMI when the backend is "nccl"; gloo on CPU for tests).

Nothing in the order loop communicates: the reference has no cross-column term anywhere in
SOS_Aer_main_specular.py:104-458.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np


def shard_indices(n_columns: int, world: int, rank: int, cost: Optional[Sequence[float]] = None) -> np.ndarray:
    """Columns of `rank`.  With `cost` (expected number of orders per column, e.g. growing with
    tau* and the surface albedo) the columns are sorted by cost and dealt in a snake so that
    every rank gets about the same total; without it they are dealt round-robin."""
    idx = np.arange(n_columns)
    if cost is not None:
        idx = idx[np.argsort(-np.asarray(cost, dtype=np.float64), kind="stable")]
    rounds = np.arange(len(idx)) // world
    pos = np.arange(len(idx)) % world
    owner = np.where(rounds % 2 == 0, pos, world - 1 - pos) if cost is not None else pos
    return np.sort(idx[owner == rank])


def expected_orders(tauStar_tot, grd_alb) -> np.ndarray:
    """Cheap monotone proxy of the order count used only for load balancing."""
    return np.asarray(tauStar_tot, dtype=np.float64) * (1.0 + 2.0 * np.asarray(grd_alb, dtype=np.float64)) + 0.1


def gather_columns(local: Dict[str, "torch.Tensor"], my_idx: np.ndarray, n_columns: int, dst: int = 0,
                   group=None) -> Optional[Dict[str, "torch.Tensor"]]:
    """Gather per-column tensors (first dimension = local columns) to `dst`, restoring the global
    column order.  Shards may be ragged; they are padded to the largest shard for the collective.
    Returns the assembled dict on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = next(iter(local.values())).device
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = len(my_idx)
    dist.all_reduce(counts, group=group)
    cmax = int(counts.max().item())
    idx_t = torch.full((cmax,), -1, dtype=torch.int64, device=dev)
    idx_t[:len(my_idx)] = torch.as_tensor(np.asarray(my_idx), dtype=torch.int64, device=dev)
    payload = {"__idx": idx_t}
    for k, v in local.items():
        pad = torch.zeros((cmax,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
        pad[:v.shape[0]] = v
        payload[k] = pad
    out = {}
    for k, v in payload.items():
        bufs = [torch.empty_like(v) for _ in range(world)] if rank == dst else None
        dist.gather(v.contiguous(), bufs, dst=dst, group=group)
        if rank == dst:
            out[k] = bufs
    if rank != dst:
        return None
    res = {}
    for k in local:
        first = out[k][0]
        full = torch.zeros((n_columns,) + tuple(first.shape[1:]), dtype=first.dtype, device=dev)
        for r in range(world):
            c = int(counts[r].item())
            full[out["__idx"][r][:c]] = out[k][r][:c]
        res[k] = full
    return res
