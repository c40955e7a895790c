"""Multi-GPU: independent columns are dealt to the ranks (one process per GPU), every rank
solves its shard with no communication, and the results are gathered once at the end
(RCCL gather over xGMI when the backend is "nccl"; gloo on CPU for tests).

Nothing in the order loop communicates: the reference has no cross-column term anywhere in
SOS_Aer_main_specular.py:104-458.

Entry points
  shard_indices / expected_orders   the deal (equal expected work per rank)
  gather_columns                    the one collective
  solve_sharded(...)                collective form of `sosrt.main.SOS_Aer_batch`: call it from every rank of an
                                    initialised process group (`torchrun --nproc-per-node G script.py`)
  solve_on_devices(devices, ...)    what `SOS_Aer_batch(..., devices=[0, 1, ...])` runs: starts one worker process
                                    per entry itself (before anything touches a GPU in them), rendezvous on
                                    127.0.0.1, returns the assembled result in the calling process
A column's result does not depend on the batch it is solved in (every tiling of the contraction uses the same
arithmetic per row), so the gathered fields equal the single-rank solve bit for bit.
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


def solve_sharded(mu0, tauStar_aer, grd_alb, *, group=None, dst=0, device=None, **kw):
    """`SOS_Aer_batch` over the ranks of a process group: the columns are dealt by expected work, every rank solves
    its shard on its own GPU (`device`, default: LOCAL_RANK), one gather assembles the fields on `dst`.
    Returns a `BatchResult` on `dst`, None elsewhere.  Keyword arguments are those of `SOS_Aer_batch`;
    per-column arrays (tauStar_atm, alb_atm, alb_aer, P0_atm, P0_aer) are sliced with the shard."""
    import os

    import torch
    import torch.distributed as dist
    from .main import BatchResult, SOS_Aer_batch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cols = ("tauStar_atm", "alb_atm", "alb_aer")
    arrs = np.broadcast_arrays(*[np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in
                                 (mu0, tauStar_aer, grd_alb) + tuple(kw.get(k, d) for k, d in zip(cols, (0.124, 1.0, 1.0)))])
    B = arrs[0].shape[0]
    mine = shard_indices(B, world, rank, expected_orders(arrs[3] + arrs[1], arrs[2]))
    sub = dict(kw)
    for k, a in zip(cols, arrs[3:]):
        sub[k] = a[mine]
    for k in ("P0_atm", "P0_aer"):
        if sub.get(k) is not None and np.ndim(sub[k]) == 2 and np.shape(sub[k])[0] == B:
            sub[k] = np.asarray(sub[k])[mine]
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    sub["raise_on_error"] = False
    if len(mine):
        r = SOS_Aer_batch(arrs[0][mine], arrs[1][mine], arrs[2][mine], device=device, **sub)
        local = {"I": r.I, "n": r.n.astype(np.int64), "status": r.status.astype(np.int64), "tau": r.tau}
        meta = (r.mu, r.idx_up, r.idx_down)
    else:                                   # more ranks than columns: an empty shard still joins the collective
        L, N = int(kw.get("nb_layers", 200)), int(kw.get("nb_angles", 128))
        local = {"I": np.zeros((0, L, 2 * N)), "n": np.zeros(0, np.int64), "status": np.zeros(0, np.int64), "tau": np.zeros((0, L))}
        meta = None
    on_gpu = dist.get_backend(group) == "nccl"
    tdev = torch.device("cuda", device) if on_gpu else torch.device("cpu")
    res = gather_columns({k: torch.from_numpy(np.ascontiguousarray(v)).to(tdev) for k, v in local.items()}, mine, B, dst=dst, group=group)
    if rank != dst:
        return None
    if meta is None:
        from .inputs import direction_grid, slab_indices
        L, N = int(kw.get("nb_layers", 200)), int(kw.get("nb_angles", 128))
        meta = (direction_grid(N),) + slab_indices(kw.get("z0", 120), kw.get("z_up", 25), kw.get("z_down", 17), L)
    out = {k: v.cpu().numpy() for k, v in res.items()}
    return BatchResult(I=out["I"], n=out["n"].astype(np.int32), status=out["status"].astype(np.int32), tau=out["tau"],
                       mu=meta[0], idx_up=meta[1], idx_down=meta[2])


def _device_worker(rank, devices, port, args, kw, out_path):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(len(devices)),
                      LOCAL_RANK=str(devices[rank]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    distinct = len(set(devices)) == len(devices)
    if distinct:
        torch.cuda.set_device(devices[rank])
        dist.init_process_group("nccl", rank=rank, world_size=len(devices), device_id=torch.device("cuda", devices[rank]))
    else:                                   # several ranks on one GPU (tests): RCCL refuses that, gloo carries the gather
        dist.init_process_group("gloo", rank=rank, world_size=len(devices))
    try:
        r = solve_sharded(*args, device=devices[rank], **kw)
        if rank == 0:
            np.savez(out_path, I=r.I, n=r.n, status=r.status, tau=r.tau, mu=r.mu, idx=np.array([r.idx_up, r.idx_down]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def solve_on_devices(devices, mu0, tauStar_aer, grd_alb, **kw):
    """One worker process per entry of `devices` (GPU ordinals of this node), started here with the `spawn` method
    so that no worker inherits an initialised GPU runtime; rank 0 hands the assembled result back through a file
    in /dev/shm (a 512-column C4 field is 210 MB)."""
    import multiprocessing as mp
    import os
    import socket
    import tempfile
    from .main import BatchResult
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("devices must name at least one GPU")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    fd, path = tempfile.mkstemp(suffix=".npz", prefix="sosrt_", dir=tmpdir)
    os.close(fd)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_device_worker, args=(r, devices, port, (mu0, tauStar_aer, grd_alb), kw, path))
             for r in range(len(devices))]
    try:
        for p in procs:
            p.start()
        for p in procs:
            p.join()
        bad = [p.exitcode for p in procs if p.exitcode != 0]
        if bad:
            raise RuntimeError("sharded solve failed: worker exit codes %s" % [p.exitcode for p in procs])
        with np.load(path) as d:
            return BatchResult(I=d["I"], n=d["n"], status=d["status"], tau=d["tau"], mu=d["mu"], idx_up=int(d["idx"][0]),
                               idx_down=int(d["idx"][1]))
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        if os.path.exists(path):
            os.unlink(path)
