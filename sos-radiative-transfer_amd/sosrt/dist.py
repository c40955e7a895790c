"""Multi-GPU: independent columns are dealt to the ranks (one process per GPU), every rank
solves its shard with no communication, and the results are gathered once at the end
(RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

Nothing in the order loop communicates: the reference has no cross-column term anywhere in
SOS_Aer_main_specular.py:104-458.

Entry points
  shard_indices / expected_orders   the deal (equal expected work per rank)
  GatherPlan                        the layout of the one collective: every rank computes it from the deal alone (no
                                    communication, no host synchronisation), once per sweep; receive buffers are allocated once
  gather_rows                       the collective itself: ragged blocks, no padding -- point-to-point sends to the root in one
                                    group (torch.distributed P2P = ncclSend / ncclRecv under RCCL: every sender on its own xGMI
                                    link to the root), or the C ABI's sosrt_gather (`via="abi"`)
  solve_sharded(...)                collective form of `sosrt.main.SOS_Aer_batch`: call it from every rank of an
                                    initialised process group (`torchrun --nproc-per-node G script.py`); the field stays on
                                    the device from the solve to the root's receive buffer
  solve_on_devices(devices, ...)    what `SOS_Aer_batch(..., devices=[0, 1, ...])` runs: starts one worker process
                                    per entry itself (before anything touches a GPU in them), rendezvous on
                                    127.0.0.1, returns the assembled result in the calling process
A column's result does not depend on the batch it is solved in (every tiling of the contraction uses the same
arithmetic per row), so the gathered fields equal the single-rank solve bit for bit.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

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


class GatherPlan:
    """Who holds which columns, and where they land on the root.  A pure function of (n_columns, world, cost): every rank
    builds the same plan locally, so nothing is exchanged to agree on counts (round 2 ran an all_reduce and a host
    synchronisation for that inside every step).  The root receives the ranks' blocks one after the other (`order` = the
    global column of every received row); `restore` puts them back in global order with one indexed copy."""

    def __init__(self, n_columns: int, world: int, cost: Optional[Sequence[float]] = None):
        self.n_columns, self.world = int(n_columns), int(world)
        self.parts: List[np.ndarray] = [shard_indices(n_columns, world, r, cost) for r in range(world)]
        self.counts = [len(p) for p in self.parts]
        self.offsets = np.concatenate(([0], np.cumsum(self.counts))).astype(np.int64)
        self.order = np.concatenate(self.parts) if n_columns else np.zeros(0, dtype=np.int64)
        self.inverse = np.argsort(self.order, kind="stable")        # row of global column c in the received buffer
        self._buffers: Dict[tuple, "torch.Tensor"] = {}
        self._inv_t: Dict[str, "torch.Tensor"] = {}

    def mine(self, rank: int) -> np.ndarray:
        return self.parts[rank]

    def buffer(self, like: "torch.Tensor", key: str = "") -> "torch.Tensor":
        """The root's receive buffer for rows shaped like `like[0]`: allocated on first use, reused by every later step."""
        import torch
        k = (key, tuple(like.shape[1:]), like.dtype, str(like.device))
        b = self._buffers.get(k)
        if b is None:
            b = self._buffers[k] = torch.empty((self.n_columns,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        return b

    def restore(self, received: "torch.Tensor") -> "torch.Tensor":
        """Rank-major rows -> global column order."""
        import torch
        d = str(received.device)
        if d not in self._inv_t:
            self._inv_t[d] = torch.as_tensor(self.inverse, dtype=torch.int64, device=received.device)
        return received.index_select(0, self._inv_t[d])


def gather_rows(local: "torch.Tensor", plan: GatherPlan, dst: int = 0, group=None, key: str = "", via: str = "torch",
                solver=None) -> Optional["torch.Tensor"]:
    """Send this rank's rows (`local`, first dimension = its columns in ascending global order) to `dst`; returns the
    root's buffer `[n_columns, ...]` in rank-major order there (see `GatherPlan.restore`), None elsewhere.  Blocks are
    ragged and unpadded.  `via="torch"`: one group of point-to-point operations through torch.distributed (RCCL ncclSend /
    ncclRecv on GPU tensors, gloo on CPU tensors), asynchronous on the current stream for RCCL.  `via="abi"`: the library's
    own `sosrt_gather` on `solver`'s communicator and stream (float64 device tensors)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert world == plan.world and local.shape[0] == plan.counts[rank], (world, plan.world, local.shape, plan.counts)
    local = local.contiguous()
    recv = plan.buffer(local, key) if rank == dst else None
    if via == "abi":
        if local.dtype != torch.float64 or not local.is_cuda:
            raise ValueError("the C-ABI gather moves float64 device tensors")
        per = int(np.prod(local.shape[1:], dtype=np.int64))
        solver.gather_device(dst, [c * per for c in plan.counts], local.data_ptr() if local.numel() else 0,
                             recv.data_ptr() if recv is not None else 0)
        return recv
    ops = []
    if rank == dst:
        for r in range(world):
            c = plan.counts[r]
            if c == 0:
                continue
            blk = recv[int(plan.offsets[r]):int(plan.offsets[r]) + c]
            if r == rank:
                blk.copy_(local)
            else:
                ops.append(dist.P2POp(dist.irecv, blk, r if group is None else dist.get_global_rank(group, r), group))
    elif plan.counts[rank] > 0:
        ops.append(dist.P2POp(dist.isend, local, dst if group is None else dist.get_global_rank(group, dst), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()                    # (RCCL: orders the current stream behind the transfer; gloo: blocks)
    return recv


def gather_columns(local: Dict[str, "torch.Tensor"], my_idx: np.ndarray, n_columns: int, dst: int = 0,
                   group=None, plan: Optional[GatherPlan] = None) -> Optional[Dict[str, "torch.Tensor"]]:
    """Gather per-column tensors (first dimension = local columns) to `dst`, restoring the global column order.  The
    layout comes from `plan` (the deal; built once by the caller) -- without one, a plan is derived here from every rank's
    `my_idx`, which costs one small all_gather and is meant for one-off calls, not for a step loop.
    Returns the assembled dict on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if plan is None:
        parts: List[Optional[np.ndarray]] = [None] * world
        dist.all_gather_object(parts, np.asarray(my_idx, dtype=np.int64), group=group)
        plan = GatherPlan(0, world)
        plan.n_columns, plan.parts, plan.counts = int(n_columns), [np.asarray(p) for p in parts], [len(p) for p in parts]
        plan.offsets = np.concatenate(([0], np.cumsum(plan.counts))).astype(np.int64)
        plan.order = np.concatenate(plan.parts)
        plan.inverse = np.argsort(plan.order, kind="stable")
    out = {}
    for k, v in local.items():
        got = gather_rows(v, plan, dst=dst, group=group, key=k)
        if rank == dst:
            out[k] = plan.restore(got)
    return out if rank == dst else None


def _comm_for(solver, group, dst):
    """RCCL communicator of `solver` over the ranks of `group` (made once per solver): the root's unique id travels through
    the process group's store."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if getattr(solver, "_comm_world", None) == (world, rank):
        return
    ids = [solver.comm_unique_id() if rank == dst else None]
    dist.broadcast_object_list(ids, src=dst if group is None else dist.get_global_rank(group, dst), group=group)
    solver.comm_init(rank, world, ids[0])
    solver._comm_world = (world, rank)


def solve_sharded(mu0, tauStar_aer, grd_alb, *, group=None, dst=0, device=None, gather="torch", **kw):
    """`SOS_Aer_batch` over the ranks of a process group: the columns are dealt by expected work, every rank solves
    its shard on its own GPU (`device`, default: LOCAL_RANK) into device memory, ONE gather of ragged blocks assembles the
    fields in the root's device buffer (no padding, no intermediate host copy under RCCL; `gather="abi"` takes the C ABI's
    sosrt_gather instead of torch.distributed), and the root copies the result to the host once.
    Returns a `BatchResult` on `dst`, None elsewhere.  Keyword arguments are those of `SOS_Aer_batch`;
    per-column arrays (tauStar_atm, alb_atm, alb_aer, P0_atm, P0_aer) are sliced with the shard."""
    import os

    import torch
    import torch.distributed as dist
    from .main import BatchResult, get_solver, solve_batch_device
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cols = ("tauStar_atm", "alb_atm", "alb_aer")
    arrs = np.broadcast_arrays(*[np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in
                                 (mu0, tauStar_aer, grd_alb) + tuple(kw.get(k, d) for k, d in zip(cols, (0.124, 1.0, 1.0)))])
    B = arrs[0].shape[0]
    plan = GatherPlan(B, world, expected_orders(arrs[3] + arrs[1], arrs[2]))
    mine = plan.mine(rank)
    sub = dict(kw)
    for k, a in zip(cols, arrs[3:]):
        sub[k] = a[mine]
    for k in ("P0_atm", "P0_aer"):
        if sub.get(k) is not None and np.ndim(sub[k]) == 2 and np.shape(sub[k])[0] == B:
            sub[k] = np.asarray(sub[k])[mine]
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    # what the device-resident solve does not carry must not be dropped silently
    if sub.get("first_order", "coded") != "coded":
        raise ValueError("solve_sharded solves with the coded first order only (first_order=%r)" % (sub["first_order"],))
    if sub.get("save_orders"):
        raise ValueError("solve_sharded does not return I_saved (save_orders=True): the per-order fields stay on the ranks")
    for k in ("raise_on_error", "save_orders", "devices", "first_order"):
        sub.pop(k, None)
    L, N = int(kw.get("nb_layers", 200)), int(kw.get("nb_angles", 128))
    tdev = torch.device("cuda", device)
    if len(mine):
        loc, meta = solve_batch_device(arrs[0][mine], arrs[1][mine], arrs[2][mine], device=device, **sub)
    else:                                   # more ranks than columns: an empty shard still joins the collective
        loc = {"I": torch.zeros((0, L, 2 * N), dtype=torch.float64, device=tdev), "n": torch.zeros(0, dtype=torch.int32, device=tdev),
               "status": torch.zeros(0, dtype=torch.int32, device=tdev), "tau": torch.zeros((0, L), dtype=torch.float64, device=tdev)}
        meta = None
    on_gpu = dist.get_backend(group) == "nccl"
    if gather == "abi" and not on_gpu:
        raise ValueError("gather='abi' needs one GPU per rank (RCCL); the gloo rehearsal uses gather='torch'")
    out = {}
    if gather == "abi":
        s = get_solver(L, N, max(1, len(mine)), int(kw.get("max_orders", 256)), device)
        s.set_stream(torch.cuda.current_stream(tdev).cuda_stream)
        _comm_for(s, group, dst)
        # one float64 block per rank: field rows, then tau, order count and status of the column
        pack = torch.cat([loc["I"].reshape(len(mine), -1), loc["tau"], loc["n"].to(torch.float64)[:, None],
                          loc["status"].to(torch.float64)[:, None]], dim=1)
        got = gather_rows(pack, plan, dst=dst, group=group, key="pack", via="abi", solver=s)
        # the cached solver goes back to its own stream (as solve_batch_device leaves it), once the gather has left this one
        torch.cuda.current_stream(tdev).synchronize()
        s.set_stream(None)
        if rank == dst:
            got = plan.restore(got)
            LD = L * 2 * N
            out = {"I": got[:, :LD].reshape(B, L, 2 * N), "tau": got[:, LD:LD + L], "n": got[:, LD + L].to(torch.int32),
                   "status": got[:, LD + L + 1].to(torch.int32)}
    else:
        for k, v in loc.items():
            got = gather_rows(v if on_gpu else v.cpu(), plan, dst=dst, group=group, key=k)
            if rank == dst:
                out[k] = plan.restore(got)
    if rank != dst:
        if on_gpu:
            torch.cuda.current_stream(tdev).synchronize()        # the sends have left before the tensors go out of scope
        return None
    if meta is None:
        from .inputs import direction_grid, slab_indices
        meta = (direction_grid(N),) + slab_indices(kw.get("z0", 120), kw.get("z_up", 25), kw.get("z_down", 17), L)
    host = {k: v.cpu().numpy() for k, v in out.items()}       # the one device-to-host copy, on the root
    return BatchResult(I=host["I"], n=host["n"].astype(np.int32), status=host["status"].astype(np.int32), tau=host["tau"],
                       mu=meta[0], idx_up=meta[1], idx_down=meta[2])


def _device_worker(rank, devices, port, args, kw, out_path):
    import datetime
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(len(devices)),
                      LOCAL_RANK=str(devices[rank]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    distinct = len(set(devices)) == len(devices)
    limit = datetime.timedelta(seconds=float(os.environ.get("SOSRT_DIST_TIMEOUT", "600")))
    if distinct:
        torch.cuda.set_device(devices[rank])
        dist.init_process_group("nccl", rank=rank, world_size=len(devices), device_id=torch.device("cuda", devices[rank]), timeout=limit)
    else:                                   # several ranks on one GPU (tests): RCCL refuses that, gloo carries the gather
        dist.init_process_group("gloo", rank=rank, world_size=len(devices), timeout=limit)
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
    in /dev/shm (a 512-column C4 field is 210 MB).  The workers are watched: when one of them dies (bad input, no such
    device) the others -- which would wait for it in the collective until the process group's timeout -- are ended, and
    the error names every exit code."""
    import multiprocessing as mp
    import os
    import socket
    import tempfile
    import time
    from .main import BatchResult
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("devices must name at least one GPU")
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    fd, path = tempfile.mkstemp(suffix=".npz", prefix="sosrt_", dir=tmpdir)
    os.close(fd)
    ctx = mp.get_context("spawn")
    # the rendezvous port: held open (SO_REUSEADDR) until the workers are about to start, so that nothing else is handed it
    sk = socket.socket()
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    procs = [ctx.Process(target=_device_worker, args=(r, devices, port, (mu0, tauStar_aer, grd_alb), kw, path))
             for r in range(len(devices))]
    try:
        sk.close()
        for p in procs:
            p.start()
        deadline = time.monotonic() + float(os.environ.get("SOSRT_DIST_TIMEOUT", "600")) + 60
        while any(p.is_alive() for p in procs):
            if any(p.exitcode not in (None, 0) for p in procs) or time.monotonic() > deadline:
                break
            time.sleep(0.05)
        codes = [p.exitcode for p in procs]
        if any(c != 0 for c in codes):
            raise RuntimeError("sharded solve failed: worker exit codes %s (None = still running when the others were ended)" % codes)
        with np.load(path) as d:
            return BatchResult(I=d["I"], n=d["n"], status=d["status"], tau=d["tau"], mu=d["mu"], idx_up=int(d["idx"][0]),
                               idx_down=int(d["idx"][1]))
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(10)
        if os.path.exists(path):
            os.unlink(path)
