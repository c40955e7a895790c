#!/usr/bin/env python3
"""Throughput of the SOS hot path: columns/s to 1e-4 convergence at (N_tau=200, N_mu=128).

One step = one pass of the hot path (I1 -> [Jn -> In] x orders, convergence test included) over
one batch of synthetic columns whose inputs are already resident in HBM.  The workload is the
BASELINE C4 sweep shape: 512 columns = 8 mu0 x 8 tau*_aer x 8 grd_alb, L=200, N=128, Rayleigh
atmosphere + the aerosol of `--aerosol` (default `eva`: the log-normal Mie ensemble of the reference's EVA scenario,
README.md:95-102, tabulated by sosrt/mie.py -- `miepython` is unavailable offline, so the table is parity-unpinned, the
solve on it is checked against the oracle on identical inputs; `wildfire`: README.md:104-111; `hg`: the HG(0.7) stand-in
of rounds 1-2), specular surface, fp64.  P0(mu, mu0) of every column is built on the device (sosrt_phase_p0_dev) before
the timed region.

Several GPUs (`--gpus N`): one process per GPU.  When no launcher environment is present (no WORLD_SIZE)
the N rank processes are started from here through torch.distributed.run, before anything touches a GPU.
  --scaling weak (default): every rank solves its own 512-column sweep (rank r uses a different aerosol
      single-scattering albedo); the TOA / surface radiances and order counts are gathered to rank 0 once
      per step.
  --scaling strong: ONE sweep (BASELINE configs[3]: 512 columns over the node, 64 per GPU at N = 8) dealt to
      the ranks by expected work (sosrt.dist.GatherPlan) and the whole fields gathered to rank 0 once per
      step -- ragged blocks, no padding, layout and receive buffers made once outside the step loop; `--gather field`:
      torch.distributed point-to-point (ncclSend / ncclRecv under RCCL), `--gather abi`: the library's own sosrt_gather.
No collective runs inside the order loop.

Prints ONE JSON line on rank 0 (see the contract in the task description).  Beside the headline (N = 1 only, outside its
timed region, `--no-extras` skips them): `extras.c2` / `extras.c3` = one EVA column at N_mu = 128 / 256 (BASELINE
configs[1], [2]: single-column latency), `extras.c5` = the 4096-column wildfire sweep at L = 400, N = 256 (configs[4]),
each with its own check against the oracle; `extras.c4_shard` = the 64 columns rank 0 gets when the C4 sweep is dealt to 8 GPUs
(BASELINE configs[3]: what bounds the 8-GPU strong-scaling number, measurable on one GPU); `extras.c4_hg` = the headline sweep
with the HG(0.7) stand-in of rounds 1-2 (continuity).  With several GPUs the line also carries `strong`: the ONE-sweep-over-the-node
measurement of configs[3] (fields gathered through the C ABI under RCCL) beside the weak-scaling headline.  `--groups 1` (default) runs the headline's order loop as one column group on one
stream, so that every kernel is timed alone on the GPU (the roofline objects are per kernel); the library by itself takes two
groups on two streams for a batch of more than 256 columns, which is faster: `two_groups` holds that measurement (same sweep, same bits).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))

import numpy as np

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix, vendor datasheet; the microarch guide lists no f64 row
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md
PMC_FILE = "r04_pmc_traffic.json"

# what the sweeps vary around (README.md:95-111): slab altitudes, range of tau*_aer, aerosol single-scattering albedo
SCENARIOS = {
    "hg": dict(z=(25, 17), taer=(0.01, 1.0), alb_aer=0.97, label="HG(0.7) aerosol stand-in"),
    "eva": dict(z=(25, 17), taer=(0.01, 1.0), alb_aer=0.97, label="EVA log-normal Mie aerosol (sigma=1.2, r_m=0.506 um, n=1.44; own Mie series, table parity-unpinned)"),
    "wildfire": dict(z=(15, 14), taer=(0.0075 / 8, 0.0075 * 8), alb_aer=0.97,
                     label="wildfire log-normal Mie aerosol (sigma=1.5, r_m=0.065 um, n=1.7+0.03j; own Mie series, table parity-unpinned)"),
}


def kernel_sources_sha():
    """Digest of the kernel sources the PMC traffic figures belong to."""
    import hashlib
    h = hashlib.sha256()
    for name in ("jn_gemm.hip", "jn_gemm_tile.hpp", "transport_ring.hip", "transport_scan.hip", "transport_scan_body.hpp", "order_loop.hip",
                 "kernels.hpp", "transport_util.hpp"):
        with open(os.path.join(ROOT, "sos-radiative-transfer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def aerosol_phase(aerosol):
    """(device kind, g, table or None) of the aerosol's scalar phase function."""
    from sosrt import inputs
    if aerosol == "hg":
        return "hg", 0.7, None
    return "table", 0.0, inputs.scenario_table(aerosol)


def build_sweep(n_columns, L, N, rank, world, vary_albedo=True, aerosol="hg"):
    """Host-side description of one sweep (the per-column P0 rows are built on the device by the caller)."""
    from sosrt import inputs
    sc = SCENARIOS[aerosol]
    side = max(1, round(n_columns ** (1 / 3)))
    mu0 = np.linspace(0.2, 1.0, side)
    taer = np.geomspace(sc["taer"][0], sc["taer"][1], side)
    rho = np.linspace(0.0, 0.8, side)
    M0, TA, RH = (x.ravel()[:n_columns] for x in np.meshgrid(mu0, taer, rho, indexing="ij"))
    B = len(M0)
    mu = inputs.direction_grid(N)
    iu, idn = inputs.slab_indices(120, sc["z"][0], sc["z"][1], L)
    tau_atm = 0.124
    alb_aer = float(np.linspace(sc["alb_aer"], sc["alb_aer"] - 0.07, world)[rank]) if (world > 1 and vary_albedo) else sc["alb_aer"]
    tau = np.stack([inputs.tau_profile(tau_atm, t, 120, sc["z"][0], sc["z"][1], L) for t in TA])
    P_atm = inputs.phase_function("rayleigh", N, mu, 0.5)[1]
    P_aer = inputs.phase_function(aerosol, N, mu, 0.5, 0.7)[1]
    return dict(B=B, L=L, N=N, mu=mu, tau=tau, P_atm=P_atm, P_aer=P_aer, mu0=M0, taer=TA, rho=RH, aerosol=aerosol,
                idx_up=iu, idx_down=idn, z=sc["z"], tau_atm=tau_atm, alb_aer=alb_aer)


def host_p0(w):
    """(P0_atm, P0_aer) rows [B, 2N] on the host (tools and tests; bench itself builds them on the device)."""
    from sosrt import inputs
    cache = {}
    for m in np.unique(w["mu0"]):
        cache[float(m)] = (inputs.phase_function("rayleigh", w["N"], w["mu"], m)[0],
                           inputs.phase_function(w.get("aerosol", "hg"), w["N"], w["mu"], m, 0.7)[0])
    return np.stack([cache[float(m)][0] for m in w["mu0"]]), np.stack([cache[float(m)][1] for m in w["mu0"]])


def take(w, idx):
    """The sub-sweep of the columns `idx` (strong scaling: this rank's shard)."""
    out = dict(w)
    for k in ("tau", "mu0", "taer", "rho"):
        out[k] = np.ascontiguousarray(w[k][idx])
    out["B"] = len(idx)
    return out


def oracle_p0(O, w, b):
    kind, g, tab = aerosol_phase(w.get("aerosol", "hg"))
    return (O.phase_p0("rayleigh", w["N"], w["mu"], float(w["mu0"][b])), O.phase_p0(kind, w["N"], w["mu"], float(w["mu0"][b]), g, tab))


def oracle_column(O, w, b, P0a, P0r):
    return O.Column(tau=w["tau"][b], mu=w["mu"], N=w["N"], idx_up=w["idx_up"], idx_down=w["idx_down"], mu0=float(w["mu0"][b]),
                    grd_alb=float(w["rho"][b]), alb_atm=1.0, alb_aer=w["alb_aer"], dtau_atm=w["tau_atm"] / w["L"],
                    dtau_aer=float(w["taer"][b]) / (w["idx_down"] + 1 - w["idx_up"]),
                    tauStar_tot=w["tau_atm"] + float(w["taer"][b]), P0_atm=P0a, P_atm=w["P_atm"], P0_aer=P0r, P_aer=w["P_aer"])


def _baseline_one(args):
    """One column of the sweep through the oracle in literal (reference-cost) mode; runs in a worker process."""
    w, b = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sos_oracle as O
    P0a, P0r = oracle_p0(O, w, b)
    t0 = time.perf_counter()
    s = O.solve_column(oracle_column(O, w, b, P0a, P0r), literal=True)
    return s.n - 1, time.perf_counter() - t0


def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def cpu_baseline(w, seconds_budget=20.0):
    """The oracle in its literal (reference-cost) mode on the host cores of this box, on a bounded sample of the same
    sweep.  `value`: ONE core (the reference is single-threaded), columns taken evenly across the sweep until the
    budget is used.  `all_cores`: one process per usable core over independent columns (how a user of the reference
    would fill the box), one column per process."""
    B = w["B"]
    model, logical, usable = cpu_info()
    order = np.linspace(0, B - 1, min(B, 8)).astype(int)
    done, orders, t0 = 0, 0, time.perf_counter()
    for b in order:
        k, _ = _baseline_one((w, int(b)))
        done += 1
        orders += k
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "columns/s", "cores": 1, "kind": "port",
           "sample": "%d columns of the same sweep (evenly spaced), %d orders, oracle literal mode, %.1f s" % (done, orders, dt),
           "cpu_model": model, "host_cores_logical": logical, "host_cores_usable": usable}
    procs = max(1, min(usable, B, 64))
    if procs > 1:
        import multiprocessing as mp
        cols = np.linspace(0, B - 1, procs).astype(int)
        wl = {k: w[k] for k in ("tau", "mu", "N", "L", "idx_up", "idx_down", "mu0", "rho", "alb_aer", "tau_atm", "taer", "P_atm", "P_aer", "aerosol")}
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        t1 = time.perf_counter()
        with mp.get_context("spawn").Pool(procs) as pool:
            res = pool.map(_baseline_one, [(wl, int(b)) for b in cols], chunksize=1)
        dtp = time.perf_counter() - t1
        out["all_cores"] = {"value": procs / dtp, "unit": "columns/s", "cores": procs,
                            "sample": "%d columns, one process each, %d orders, %.1f s wall (slowest column %.1f s)" % (
                                procs, sum(r[0] for r in res), dtp, max(r[1] for r in res))}
    return out


class Lane:
    """One handle + HIP stream + resident buffers for a sweep: inputs uploaded, P0 rows built on the device, `solve()`
    enqueues one step."""

    def __init__(self, w, dev, local_rank, max_orders, d_shared=None):
        import torch
        from sosrt.solver import Solver
        B, L, N = w["B"], w["L"], w["N"]
        self.w, self.dev, self.B, self.L, self.N = w, dev, B, L, N
        self.stream = torch.cuda.Stream(device=dev)
        self.s = Solver(L, N, max_batch=B, max_orders=max_orders, device=local_rank)
        self.s.set_stream(self.stream.cuda_stream)
        self.s.set_grid(w["mu"])
        self.s.set_phase(w["P_atm"], w["P_aer"])
        self.s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
                           w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
        if d_shared is None:
            # the inputs of the path: tau uploaded; one P0 row per column built where the solve reads them
            # (phase:86-103,148-165,245-262 on the device; outside the timed region, like the other inputs)
            d_tau = torch.from_numpy(w["tau"]).to(dev)
            d_mu0 = torch.from_numpy(np.ascontiguousarray(w["mu0"])).to(dev)
            d_P0a = torch.empty((B, 2 * N), dtype=torch.float64, device=dev)
            d_P0r = torch.empty((B, 2 * N), dtype=torch.float64, device=dev)
            torch.cuda.synchronize(dev)
            kind, g, tab = aerosol_phase(w.get("aerosol", "hg"))
            if tab is not None:
                self.s.set_phase_table(*tab)
            self.s.phase_p0_device("rayleigh", d_mu0.data_ptr(), d_P0a.data_ptr(), B)
            self.s.phase_p0_device(kind, d_mu0.data_ptr(), d_P0r.data_ptr(), B, g=g)
            self.s.synchronize()
            d_shared = (d_tau, d_P0a, d_P0r)
        self.d_tau, self.d_P0a, self.d_P0r = d_shared
        self.I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev)
        self.n = torch.zeros(B, dtype=torch.int32, device=dev)
        self.st = torch.zeros(B, dtype=torch.int32, device=dev)

    def shared(self):
        return self.d_tau, self.d_P0a, self.d_P0r

    def solve(self):
        self.s.solve_device(self.d_tau.data_ptr(), self.d_P0a.data_ptr(), self.d_P0r.data_ptr(), self.I.data_ptr(), tol=1e-4,
                            d_n_orders=self.n.data_ptr(), d_status=self.st.data_ptr())

    def check(self, O, cols, max_orders=10000):
        """Columns `cols` of the field the last step left on the device against the oracle (vectorised mode, same arithmetic as
        the reference to rounding) on IDENTICAL inputs: the device-built P0 rows go into the oracle; they are compared with the
        oracle's own evaluation beside that.  `max_orders`: the order budget the lane was made with, where it matters."""
        w, N = self.w, self.N
        n_host = self.n.cpu().numpy()
        worst, same_n, p0_err = 0.0, True, 0.0
        for b in cols:
            P0a_d, P0r_d = self.d_P0a[b].cpu().numpy(), self.d_P0r[b].cpu().numpy()
            P0a, P0r = oracle_p0(O, w, b)
            p0_err = max(p0_err, float(np.max(np.abs(P0a_d - P0a) / P0a)), float(np.max(np.abs(P0r_d - P0r) / P0r)))
            ref = O.solve_column(oracle_column(O, w, b, P0a_d, P0r_d), literal=False, max_orders=max_orders)
            got = self.I[b].cpu().numpy()
            scale = np.max(np.abs(ref.I))
            sig = np.abs(ref.I) > 1e-9 * scale
            err = max(float(np.max(np.abs(got - ref.I)[sig] / np.abs(ref.I)[sig])), float(np.max(np.abs(got - ref.I)) / scale))
            worst = max(worst, err)
            same_n = same_n and int(n_host[b]) == ref.n
        return {"max_rel_err_vs_oracle": worst, "orders_match": bool(same_n), "p0_max_rel_err_vs_oracle": p0_err,
                "tolerance": 1e-10, "ok": bool(worst <= 1e-10 and same_n and p0_err <= 1e-12)}

    def close(self):
        self.s.close()


def extra_case(O, dev, local_rank, n_columns, L, N, aerosol, steps, check_cols, max_orders=256, sweep=None, rho=0.15, tau_atm=0.124,
               check_orders=None):
    """One more configuration of BASELINE.json beside the headline, outside its timed region: `steps` solves one at a time on
    one stream, wall time, its own check.  `sweep`: a prepared sweep (build_sweep / take) instead of n_columns."""
    import torch
    w = sweep if sweep is not None else (build_sweep(n_columns, L, N, 0, 1, aerosol=aerosol) if n_columns > 1 else None)
    if w is None:          # one column at the scenario's own values (README.md:95-111)
        from sosrt import inputs
        sc = SCENARIOS[aerosol]
        taer = 0.120 if aerosol != "wildfire" else 0.0075
        mu = inputs.direction_grid(N)
        iu, idn = inputs.slab_indices(120, sc["z"][0], sc["z"][1], L)
        w = dict(B=1, L=L, N=N, mu=mu, tau=inputs.tau_profile(tau_atm, taer, 120, sc["z"][0], sc["z"][1], L)[None],
                 P_atm=inputs.phase_function_device("rayleigh", N, mu, 0.5, device=local_rank)[1],
                 P_aer=inputs.phase_function_device(aerosol, N, mu, 0.5, 0.7, device=local_rank)[1],
                 mu0=np.array([0.5]), taer=np.array([taer]), rho=np.array([float(rho)]), aerosol=aerosol, idx_up=iu, idx_down=idn,
                 z=sc["z"], tau_atm=tau_atm, alb_aer=sc["alb_aer"])
    ln = Lane(w, dev, local_rank, max_orders)
    try:
        ln.solve(); torch.cuda.synchronize(dev)                # warm-up
        t0 = time.perf_counter()
        for _ in range(steps):
            ln.solve()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        n = ln.n.cpu().numpy()
        orders = int((n - 1).sum())
        out = {"workload": "%d column%s, L=%d, N=%d, Rayleigh + %s" % (w["B"], "" if w["B"] == 1 else "s", L, N, SCENARIOS[aerosol]["label"]),
               "ms_per_solve": dt * 1e3, "columns_per_s": w["B"] / dt, "orders": orders, "max_order": int(n.max()),
               "us_per_order": dt * 1e6 / max(int(n.max()) - 1, 1), "not_converged": int((ln.st.cpu().numpy() != 0).sum()),
               "steps": steps}
        ol = ln.s.order_loop_stats(True)
        out["order_loop_launches"] = {"launches": ol[0], "refused": ol[1], "column_orders": int(ol[2]),
                                      "note": "orders of the last solve that ran inside order-loop launches (csrc/order_loop.hip)"}
        # single-GPU roofline of SURVEY 8(d) for this shape: max(t_flop, t_byte), both flop conventions
        D = 2 * N
        t_flop = 2.0 * L * D * D * orders / (FP64_MFMA_PEAK_TFLOPS * 1e12)
        t_byte = 8.0 * L * D * (4 * orders + 2 * w["B"]) / (HBM_PEAK_GBS * 1e9)
        out["roofline_frac_full_product"] = max(t_flop, t_byte) / dt
        out["roofline_frac_executed"] = max(t_flop / 2, t_byte) / dt
        if check_cols and check_orders:
            # a size at which the oracle takes seconds per order: the check is of the first `check_orders` orders -- a second
            # lane with that order budget against the oracle with the same budget
            lc = Lane(w, dev, local_rank, check_orders, ln.shared())
            try:
                lc.solve(); torch.cuda.synchronize(dev)
                out["check"] = dict(lc.check(O, check_cols, max_orders=check_orders), columns=[int(b) for b in check_cols],
                                    orders_checked=check_orders)
            finally:
                lc.close()
        elif check_cols:
            out["check"] = dict(ln.check(O, check_cols), columns=[int(b) for b in check_cols])
        return out
    finally:
        ln.close()


def measure_strong(a, dev, local_rank, rank, world, on_gpu):
    """BASELINE configs[3] beside a weak-scaling headline: ONE sweep of --columns columns dealt to the ranks, every rank solves its
    shard, the fields are gathered to rank 0 once per step (no collective in the order loop).  Timed like the headline: barrier +
    synchronize on both sides of exactly --steps steps, max over ranks.  Returns the object on rank 0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    from sosrt import dist as sdist
    torch.cuda.set_device(local_rank)                          # (the current device is per thread: this runs in one of its own)
    w1 = build_sweep(a.columns, a.layers, a.angles, 0, 1, vary_albedo=False, aerosol=a.aerosol)
    plan = sdist.GatherPlan(w1["B"], world, sdist.expected_orders(w1["tau_atm"] + w1["taer"], w1["rho"]))
    mine = plan.mine(rank)
    if len(mine) == 0:
        return {"status": "skipped: fewer columns than ranks"} if rank == 0 else None
    ws = take(w1, mine)
    via = "abi" if on_gpu else "torch"
    os.environ.pop("SOSRT_GROUPS", None)                      # the library's own order loop for a shard of this size
    ln = Lane(ws, dev, local_rank, a.max_orders)
    os.environ["SOSRT_GROUPS"] = a.groups if a.groups != "auto" else os.environ.get("SOSRT_GROUPS", "")
    if os.environ["SOSRT_GROUPS"] == "":
        os.environ.pop("SOSRT_GROUPS")
    got = {}
    try:
        if via == "abi":
            sdist._comm_for(ln.s, None, 0)

        def step():
            ln.solve()
            if via == "abi":
                with torch.cuda.stream(ln.stream):
                    got["I"] = sdist.gather_rows(ln.I, plan, dst=0, key="sI", via="abi", solver=ln.s)
                    got["n"] = sdist.gather_rows(ln.n.to(torch.float64)[:, None].contiguous(), plan, dst=0, key="sn", via="abi", solver=ln.s)
            else:
                ln.stream.synchronize()
                got["I"] = sdist.gather_rows(ln.I if on_gpu else ln.I.cpu(), plan, dst=0, key="sI")
                got["n"] = sdist.gather_rows(ln.n if on_gpu else ln.n.cpu(), plan, dst=0, key="sn")

        def sync():
            torch.cuda.synchronize(dev)
            dist.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(max(1, a.warmup)):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        sync()
        dts = time.perf_counter() - t0
        cpu_t = dev if on_gpu else "cpu"
        t = torch.tensor([dts], dtype=torch.float64, device=cpu_t)
        pr = torch.zeros(world, 2, dtype=torch.float64, device=cpu_t)
        pr[rank, 0] = len(mine)
        pr[rank, 1] = float((ln.n.cpu().numpy() - 1).sum())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(pr)
        if rank != 0:
            return None
        dts = float(t.item())
        pr = pr.cpu().numpy()
        gI = plan.restore(got["I"])
        sel = torch.as_tensor(np.asarray(mine)[[0, len(mine) - 1]], device=gI.device)
        placed = bool(torch.equal(gI[sel], torch.stack([ln.I[0], ln.I[len(mine) - 1]]).to(gI.device)))
        return {"metric": "SOS columns/sec to 1e-4 convergence, ONE sweep over the node (BASELINE configs[3])", "scaling": "strong",
                "value": w1["B"] * a.steps / dts, "unit": "columns/s", "n_gpus": world, "steps": a.steps, "ms_per_step": dts / a.steps * 1e3,
                "columns": int(w1["B"]), "columns_per_gpu": [int(x) for x in pr[:, 0]], "orders_per_step_per_rank": [int(x) for x in pr[:, 1]],
                "gather": "whole fields to rank 0 once per step, " + ("the C ABI's sosrt_gather (ncclSend / ncclRecv over RCCL)" if via == "abi"
                                                                       else "torch.distributed point-to-point (gloo rehearsal)"),
                "gather_places_columns": placed}
    finally:
        ln.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--columns", type=int, default=512, help="columns per GPU")
    ap.add_argument("--layers", type=int, default=200)
    ap.add_argument("--angles", type=int, default=128)
    ap.add_argument("--aerosol", choices=tuple(SCENARIOS), default="eva")
    ap.add_argument("--max-orders", type=int, default=256)
    ap.add_argument("--inflight", type=int, default=1, help="independent solves in flight (own handle + stream each)")
    ap.add_argument("--groups", choices=("1", "2", "auto"), default="1",
                    help="column groups of the order loop in the headline's timed region (SOSRT_GROUPS).  1 (default): one stream, "
                         "every kernel has the GPU to itself, so the per-kernel roofline below is what the kernel does; auto: the "
                         "library's own choice (two groups on two streams for a batch of this size: contraction and transport of "
                         "the halves side by side, faster, per-launch durations no longer those of a kernel alone) -- measured "
                         "beside the headline as `two_groups`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the C2 / C3 / C5 figures beside the headline")
    ap.add_argument("--pipelined", type=int, default=3,
                    help="after the timed region, also measure the throughput with this many steps in flight on "
                         "separate streams (0: skip); reported beside the headline value, never instead of it")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--gather", choices=("digest", "field", "abi"), default=None,
                    help="what rank 0 receives per step: TOA/surface rows + order counts (digest), or the whole fields through "
                         "torch.distributed point-to-point (field) or through the C ABI's sosrt_gather (abi) "
                         "(default: digest for weak scaling, field for strong)")
    ap.add_argument("--check-columns", type=int, default=3, help="columns compared with the oracle after the timed region")
    a = ap.parse_args()
    if a.gather is None:
        a.gather = "field" if a.scaling == "strong" else "digest"
    if a.scaling == "strong" and a.gather == "digest" and a.gpus > 1:
        raise SystemExit("--scaling strong deals ragged shards: gather the fields (--gather field | abi), not equal-sized digests")

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, before this process touches a GPU
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    # the driver launches one rank per GPU over RCCL; SOSRT_BENCH_BACKEND=gloo + SOSRT_BENCH_SHARE_GPU=1 lets the
    # tests rehearse the multi-rank path with two ranks on a one-GPU box
    backend = os.environ.get("SOSRT_BENCH_BACKEND", "nccl")
    if os.environ.get("SOSRT_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if a.gather == "abi" and world > 1 and backend != "nccl":
        raise SystemExit("--gather abi needs RCCL (one GPU per rank); the gloo rehearsal uses --gather field")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    ge.build()
    from sosrt import _lib
    from sosrt import dist as sdist

    strong = a.scaling == "strong" and world > 1
    w_all = build_sweep(a.columns, a.layers, a.angles, rank, world, vary_albedo=not strong, aerosol=a.aerosol)
    n_global = w_all["B"]
    plan = None
    if strong:
        # ONE sweep over the node: columns dealt by expected work, every rank gets about the same sum of orders.  The layout
        # of the gather follows from the deal alone: computed here, once, on every rank, without communication
        plan = sdist.GatherPlan(n_global, world, sdist.expected_orders(w_all["tau_atm"] + w_all["taer"], w_all["rho"]))
        mine = plan.mine(rank)
        w = take(w_all, mine)
    else:
        mine = np.arange(n_global)
        w = w_all
        if world > 1 and a.gather in ("field", "abi"):
            plan = sdist.GatherPlan(0, world)             # weak scaling, whole fields: every rank sends all its B columns
            plan.n_columns, plan.counts = world * n_global, [n_global] * world
            plan.offsets = np.arange(world + 1, dtype=np.int64) * n_global
    B, L, N = w["B"], w["L"], w["N"]
    D = 2 * N
    if B == 0:
        raise SystemExit("rank %d has no columns: --columns must be at least --gpus" % rank)

    # `inflight` independent solves may be in flight at once, each on its own handle and HIP stream:
    # the order loop of a sweep ends in a long tail of launches over its few slowest-converging
    # columns, which leaves most CUs idle; the next sweep's dense launches fill them.
    torch.cuda.synchronize(dev)
    if a.groups == "auto":
        os.environ.pop("SOSRT_GROUPS", None)
    else:
        os.environ["SOSRT_GROUPS"] = a.groups                     # (read by sosrt_create)
    lanes = [Lane(w, dev, local_rank, a.max_orders)]
    while len(lanes) < max(1, a.inflight):
        lanes.append(Lane(w, dev, local_rank, a.max_orders, lanes[0].shared()))
    main_stream = torch.cuda.current_stream(dev)
    on_gpu = backend == "nccl"
    if world > 1 and a.gather == "abi":
        for ln in lanes:
            sdist._comm_for(ln.s, None, 0)

    # per-step digest of a lane (weak scaling): equal blocks, receive buffers made once
    dig_bufs = None
    if world > 1 and a.gather == "digest" and rank == 0:
        dig_bufs = [torch.empty((B, 2 * N + 1), dtype=torch.float64, device=dev if on_gpu else "cpu") for _ in range(world)]

    from concurrent.futures import ThreadPoolExecutor
    execs = [ThreadPoolExecutor(max_workers=1) for _ in lanes]       # a lane runs its steps in order
    gathered = {}

    def lane_step(i):
        """Enqueue one step on lane i's stream; returns (what rank 0 is to receive or None, completion event)."""
        torch.cuda.set_device(local_rank)
        ln = lanes[i]
        ln.solve()
        send = None
        if world > 1:
            with torch.cuda.stream(ln.stream):
                if a.gather == "digest":
                    send = torch.cat([ln.I[:, 0, N:], ln.I[:, L - 1, :N], ln.n.to(torch.float64)[:, None]], dim=1).contiguous()
                elif a.gather == "abi":
                    # the C ABI's gather, on the lane's own stream behind the solve: ragged counts, the root's buffer made once
                    got = sdist.gather_rows(ln.I, plan, dst=0, key="I%d" % i, via="abi", solver=ln.s)
                    got_n = sdist.gather_rows(ln.n.to(torch.float64)[:, None].contiguous(), plan, dst=0, key="n%d" % i, via="abi", solver=ln.s)
                    send = (got, got_n)
        ev = torch.cuda.Event()
        ev.record(ln.stream)
        return send, ev

    def run_steps(k):
        """k steps dealt round-robin to the lanes.  With several ranks the results of every step are gathered to
        rank 0 -- the only collective -- in step order, behind the step's completion event: per-column digests (TOA-up row,
        surface-down row, order count) or the whole fields."""
        futs = [(step % len(lanes), execs[step % len(lanes)].submit(lane_step, step % len(lanes))) for step in range(k)]
        for li, f in futs:
            send, ev = f.result()
            if world == 1:
                continue
            main_stream.wait_event(ev)
            ln = lanes[li]
            if a.gather == "abi":
                if rank == 0:
                    gathered["I"], gathered["n"] = send
            elif a.gather == "field":
                if not on_gpu:                    # gloo carries host tensors
                    ev.synchronize()
                got = sdist.gather_rows(ln.I if on_gpu else ln.I.cpu(), plan, dst=0, key="I")
                got_n = sdist.gather_rows(ln.n if on_gpu else ln.n.cpu(), plan, dst=0, key="n")
                if rank == 0:
                    gathered["I"], gathered["n"] = got, got_n
            else:
                if not on_gpu:
                    ev.synchronize()
                    send = send.cpu()
                dist.gather(send, dig_bufs, dst=0)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    run_steps(a.warmup)
    sync_all()
    for ln in lanes:
        ln.s.profile_enable(True)
        ln.s.profile_reset()
    t0 = time.perf_counter()
    run_steps(a.steps)
    sync_all()
    dt = time.perf_counter() - t0
    n_host = lanes[0].n.cpu().numpy()
    st_host = lanes[0].st.cpu().numpy()
    orders_per_step = int((n_host - 1).sum())
    gemm_ms = tr_ms = fo_ms = ol_ms = 0.0
    gemm_launches = tr_launches = ol_launches = 0
    for ln in lanes:
        ms, cnt = ln.s.profile_get(_lib.K_GEMM); gemm_ms += ms; gemm_launches += cnt
        ms, cnt = ln.s.profile_get(_lib.K_TRANSPORT); tr_ms += ms; tr_launches += cnt
        ms, cnt = ln.s.profile_get(_lib.K_ORDER_LOOP); ol_ms += ms; ol_launches += cnt
        fo_ms += ln.s.profile_get(_lib.K_FIRST)[0]
        ln.s.profile_enable(False)
    # (column, order) pairs of a step that ran inside order-loop launches (the last orders of the last live columns: one launch
    # holds the transport and the contraction of all of them) -- the same in every step
    ol_stats = lanes[0].s.order_loop_stats(True)
    ol_orders_per_step = int(ol_stats[2])

    # Outside the timed region: sampled columns of the field the last timed step left on the device against the oracle.
    # Rank 0, its own columns.
    check = None
    O = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import sos_oracle as O
    if rank == 0 and a.check_columns > 0:
        last = lanes[(a.steps - 1) % len(lanes)]
        cols = sorted(set(np.linspace(0, B - 1, min(B, a.check_columns)).astype(int).tolist()))
        check = dict(last.check(O, cols), columns=[int(mine[b]) for b in cols])
        if world > 1 and a.gather in ("field", "abi") and gathered:
            # the gathered buffer holds the ranks' blocks one after the other; in global order (plan.restore) rank 0's own
            # columns must be where they belong
            gI = gathered["I"]
            if strong:
                gI = plan.restore(gI)
                sel = torch.as_tensor(np.asarray(mine)[cols], device=gI.device)
            else:
                sel = torch.as_tensor(np.asarray(cols), device=gI.device)
            mineI = torch.stack([last.I[b] for b in cols]).to(gI.device)
            check["gather_places_columns"] = bool(torch.equal(gI[sel], mineI))
            check["ok"] = bool(check["ok"] and check["gather_places_columns"])

    # Beside the headline too: the same sweep with the order loop the library chooses by itself for a batch of this size -- two
    # column groups on two streams -- one step at a time; the field must have the headline's bits.
    two = None
    # (measured before the `pipelined` lanes exist: HIP maps streams onto a few hardware queues, and with eight streams alive the
    # two streams of this handle can land on one queue and serialise)
    if world == 1 and a.groups == "1" and not a.no_extras:
        os.environ.pop("SOSRT_GROUPS", None)
        l2 = Lane(w, dev, local_rank, a.max_orders, lanes[0].shared())
        os.environ["SOSRT_GROUPS"] = a.groups
        try:
            for _ in range(2):
                l2.solve()
            torch.cuda.synchronize(dev)
            tsteps = max(a.steps, 10)
            t20 = time.perf_counter()
            for _ in range(tsteps):
                l2.solve()
            torch.cuda.synchronize(dev)
            dt2 = (time.perf_counter() - t20) / tsteps
            two = {"order_loop": "SOSRT_GROUPS unset: the library's choice (two column groups on two streams for a batch of more than 256 columns)", "steps": tsteps, "ms_per_step": dt2 * 1e3,
                   "value": B / dt2, "unit": "columns/s",
                   "same_bits_as_headline": bool(torch.equal(l2.I, lanes[0].I) and torch.equal(l2.n, lanes[0].n))}
        finally:
            l2.close()

    # Beside the headline (one step at a time, one stream: every kernel timing above is of a launch that
    # has the GPU to itself): the same steps with `--pipelined` of them in flight on separate streams and
    # handles.  The tail of one sweep's order loop then overlaps the dense launches of the next.
    pipe = None
    if a.pipelined > 1 and len(lanes) == 1 and world == 1:
        while len(lanes) < a.pipelined:
            lanes.append(Lane(w, dev, local_rank, a.max_orders, lanes[0].shared()))
            execs.append(ThreadPoolExecutor(max_workers=1))
        psteps = max(a.steps, 2 * a.pipelined)
        run_steps(a.pipelined)
        sync_all()
        tp0 = time.perf_counter()
        run_steps(psteps)
        sync_all()
        dtp = time.perf_counter() - tp0
        pipe = {"steps_in_flight": a.pipelined, "steps": psteps, "value": B * psteps / dtp, "unit": "columns/s",
                "ms_per_step": dtp / psteps * 1e3}

    # Several GPUs, weak-scaling headline: the same command also measures BASELINE configs[3] -- ONE sweep over the node, dealt to
    # the ranks by expected work (sosrt.dist.GatherPlan), the whole fields gathered to rank 0 once per step (through the C ABI's
    # sosrt_gather under RCCL; torch.distributed point-to-point in the gloo rehearsal) -- so that one driver run yields both curves.
    # (measured AFTER every collective of the weak-scaling headline below, under a watchdog: the headline must not depend on it)
    strong_obj = None

    cpu_t = dev if on_gpu else "cpu"
    t = torch.tensor([dt], dtype=torch.float64, device=cpu_t)
    per_rank = torch.zeros(world, 2, dtype=torch.float64, device=cpu_t)
    per_rank[rank, 0] = B
    per_rank[rank, 1] = orders_per_step
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank)
    dt = float(t.item())
    per_rank = per_rank.cpu().numpy()

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        total_columns = int(per_rank[:, 0].sum())
        value = total_columns * a.steps / dt
        # The Jn contraction.  Flops of the algorithm that runs, per live (column, order) pair: 2 L D^2 for the full
        # product (no credit for the second slab matrix or for padding); L D^2 when the folded matrices are
        # flip-symmetric and the library runs the two N x N products (sosrt.h, sosrt_set_contraction) -- `achieved` counts
        # those, the rate in units of the full product is reported beside it.
        asym, uses_sym = lanes[0].s.phase_asymmetry()
        KO_all = orders_per_step * a.steps                        # column.orders in the timed region (this rank)
        KO_ol = ol_orders_per_step * a.steps                      # ... of which inside order-loop launches
        KO = KO_all - KO_ol                                       # ... and as two launches per order: the work of the two kernels below
        full_flops = 2.0 * L * D * D * KO
        flops = full_flops / 2 if uses_sym else full_flops
        achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # The transport (Jn not fused): per element and order it reads Jn and I, writes In and I = 32 bytes.  (Its
        # attenuation tables are shared per optical-depth profile and served from cache: not counted -- round 2 counted them.)
        tr_bytes = 32.0 * L * D * KO
        tr_gbs = tr_bytes / (tr_ms * 1e-3) / 1e9 if tr_ms > 0 else 0.0
        out = {
            "metric": "SOS columns/sec to 1e-4 convergence (Ntau=200, Nmu=128)",
            "value": value, "unit": "columns/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4 sweep: %s = mu0 x tau*_aer x grd_alb grid, L=%d, N=%d (D=%d), "
                                   "Rayleigh atm + %s, specular surface, tol 1e-4" % (
                                       ("%d columns over %d GPUs" % (n_global, world)) if strong else ("%d columns/GPU" % B), L, N, D,
                                       SCENARIOS[a.aerosol]["label"]),
                       "aerosol": a.aerosol,
                       "columns_per_gpu": [int(x) for x in per_rank[:, 0]] if world > 1 else B,
                       "orders_per_step": orders_per_step, "orders_per_step_per_rank": [int(x) for x in per_rank[:, 1]],
                       "max_order": int(n_host.max()),
                       "not_converged": int((st_host != 0).sum()), "inflight_solves": max(1, a.inflight),
                       "p0": "built on the device (sosrt_phase_p0_dev), outside the timed region",
                       "library_default": bool(a.groups == "auto" or B <= 48),
                       "order_loop": {"1": "one column group, one stream (SOSRT_GROUPS=1): no kernel runs beside another; the library "
                                           "alone takes two groups at this size: `two_groups` is what a default library call executes",
                                      "2": "two column groups on two streams (SOSRT_GROUPS=2)",
                                      "auto": "the library's choice (SOSRT_GROUPS unset)"}[a.groups],
                       "gather": ({"digest": "digest (TOA / surface rows, order counts) to rank 0 once per step, torch.distributed gather",
                                   "field": "whole fields to rank 0 once per step, ragged point-to-point blocks (torch.distributed)",
                                   "abi": "whole fields to rank 0 once per step through the C ABI (sosrt_gather: ncclSend / ncclRecv)"}[a.gather])
                       if world > 1 else "none",
                       "parallelism": ("one sweep dealt to %d ranks by expected orders, gather only" % world) if strong else
                                      ("columns sharded x%d, gather only" % world)},
            "roofline": None, "roofline_other": None,
            "kernel_ms_per_step": {"k_jn_gemm": gemm_ms / a.steps, "k_transport": tr_ms / a.steps,
                                   "k_first_order": fo_ms / a.steps, "k_order_loop": ol_ms / a.steps},
        }
        if check is not None:
            out["check"] = check
        r_gemm = {"bound": "mfma", "kernel": "k_jn_gemm", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                  "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                  "avg_launch_ms": gemm_ms / max(gemm_launches, 1), "launches": gemm_launches,
                  "total_ms_per_step": gemm_ms / a.steps,
                  "work_per_launch": flops / max(gemm_launches, 1), "work_unit": "flop", "column_orders_per_step": KO // max(a.steps, 1),
                  "flops_per_column_order": flops / max(KO, 1),
                  "form": ("flip-symmetric: two N x N products per row, L D^2 flops" if uses_sym else "full 2N x 2N product, 2 L D^2 flops"),
                  "matrix_asymmetry": asym,
                  "full_product_equivalent_tflops": full_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0}
        r_tr = {"bound": "hbm", "kernel": "k_transport_ring + k_transport_scan", "achieved": tr_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": tr_gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": tr_ms / max(tr_launches, 1),
                "launches": tr_launches, "total_ms_per_step": tr_ms / a.steps,
                "work_per_launch": tr_bytes / max(tr_launches, 1), "work_unit": "bytes (32 L D per column.order: read Jn, I; write In, I)",
                "column_orders_per_step": KO // max(a.steps, 1)}
        # The order-loop launches: the last orders of the last live columns, transport and contraction of all of them in one
        # launch (csrc/order_loop.hip).  Latency-bound by construction (a few columns on 256 CUs): priced like the transport, on
        # the 32 L D bytes per column.order it moves, with the flops of its contraction role beside that.
        ol_bytes = 32.0 * L * D * KO_ol
        ol_flops = (L * D * D if uses_sym else 2.0 * L * D * D) * KO_ol
        ol_gbs = ol_bytes / (ol_ms * 1e-3) / 1e9 if ol_ms > 0 else 0.0
        r_ol = {"bound": "hbm", "kernel": "k_order_loop (transport + contraction roles of one launch)", "achieved": ol_gbs, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ol_gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ol_ms / max(ol_launches, 1),
                "launches": ol_launches, "total_ms_per_step": ol_ms / a.steps, "work_per_launch": ol_bytes / max(ol_launches, 1),
                "work_unit": "bytes (32 L D per column.order)", "column_orders_per_step": ol_orders_per_step,
                "orders_per_launch_longest_column": None,
                "contraction_tflops": ol_flops / (ol_ms * 1e-3) / 1e12 if ol_ms > 0 else 0.0,
                "refused_launches": int(ol_stats[1]),
                "note": "latency-bound: the columns' serial chains, not the bytes"}
        # HBM bytes per launch come from separate rocprofv3 --pmc passes of the same workload (the counters cannot be
        # read from inside this process).  The committed file names the source revision of the kernels it was
        # measured on; it is used only while those sources are unchanged, otherwise traffic stays null.
        try:
            with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
                pmc = json.load(f)
            if (B == 512 and L == 200 and N == 128 and world == 1 and pmc.get("aerosol", "hg") == a.aerosol
                    and pmc.get("kernel_sources_sha") == kernel_sources_sha()):
                r_gemm["traffic"] = pmc["k_jn_gemm"]["hbm_bytes_per_launch"]
                r_tr["traffic"] = pmc["k_transport"]["hbm_bytes_per_launch"]
                r_gemm["traffic_unit"] = r_tr["traffic_unit"] = "bytes/launch (rocprofv3 PMC, profiles/%s)" % PMC_FILE
                r_tr["frac_on_pmc_traffic"] = r_tr["traffic"] / (r_tr["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        except (OSError, KeyError, ValueError):
            pass
        # the roofline object is the kernel with the largest share of the timed region
        ranked = sorted([(gemm_ms, 0, r_gemm), (tr_ms, 1, r_tr), (ol_ms, 2, r_ol)], key=lambda x: (-x[0], x[1]))
        out["roofline"], out["roofline_other"] = ranked[0][2], ranked[1][2]
        if ol_launches:
            out["roofline_order_loop"] = r_ol
        # SURVEY 8(d), column level: the step against max(t_flop, t_byte) of one GPU's share.  t_flop in both conventions
        # (2 L D^2 per column.order as 8d writes it / L D^2 as the symmetric form executes); t_byte = the compulsory
        # 8 L D (4 K + 2) with Jn fused (+ 16 L D K while Jn round-trips HBM, as it does here).
        ko1 = float(per_rank[:, 1].max())
        b1 = float(per_rank[:, 0].max())
        t_flop = 2.0 * L * D * D * ko1 / (FP64_MFMA_PEAK_TFLOPS * 1e12) * 1e3
        t_byte = 8.0 * L * D * (4 * ko1 + 2 * b1) / (HBM_PEAK_GBS * 1e9) * 1e3
        t_byte_unfused = t_byte + 16.0 * L * D * ko1 / (HBM_PEAK_GBS * 1e9) * 1e3
        out["roofline_step"] = {"ms_per_step": ms_per_step, "t_flop_full_product_ms": t_flop, "t_flop_executed_ms": t_flop / (2 if uses_sym else 1),
                                "t_byte_compulsory_ms": t_byte, "t_byte_with_jn_round_trip_ms": t_byte_unfused,
                                "frac_full_product": max(t_flop, t_byte) / ms_per_step,
                                "frac_executed": max(t_flop / (2 if uses_sym else 1), t_byte) / ms_per_step,
                                "peaks": "%.1f TFLOP/s FP64 matrix, %.0f GB/s HBM" % (FP64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS)}
        if pipe:
            out["pipelined"] = pipe
        if two:
            out["two_groups"] = two
        if world == 1 and not a.no_extras:
            # the other single-GPU configurations of BASELINE.json, outside the headline's timed region
            for ln in lanes:
                ln.close()
            lanes.clear()
            torch.cuda.empty_cache()
            os.environ.pop("SOSRT_GROUPS", None)          # (the library's own order loop: one group for one column and for the 64-column shard, two for the 4096 columns of C5)
            ex = {}
            try:
                ex["c2"] = extra_case(O, dev, local_rank, 1, 200, 128, "eva", 20, [0])
                ex["c3"] = extra_case(O, dev, local_rank, 1, 200, 256, "eva", 20, [0])
                ex["c5"] = extra_case(O, dev, local_rank, 4096, 400, 256, "wildfire", 2, [16 * 16 * 5 + 16 * 9 + 4])
                # BASELINE configs[3] on one GPU: the shard rank 0 gets when the C4 sweep is dealt to 8 ranks (sosrt.dist.GatherPlan:
                # sorted by expected orders, dealt in a snake), solved like the other extras.  512 columns / its time is the most
                # an 8-GPU strong-scaling run of that sweep can reach before its gather.
                w4 = build_sweep(512, 200, 128, 0, 1, aerosol=a.aerosol)
                plan8 = sdist.GatherPlan(w4["B"], 8, sdist.expected_orders(w4["tau_atm"] + w4["taer"], w4["rho"]))
                shard = np.asarray(plan8.mine(0))
                c4s = extra_case(O, dev, local_rank, len(shard), 200, 128, a.aerosol, 20, [0, len(shard) - 1], sweep=take(w4, shard))
                c4s["columns_of_the_sweep"] = [int(x) for x in shard[:4]] + ["..."] + [int(x) for x in shard[-2:]]
                c4s["implied_8gpu_strong_ceiling"] = {"columns_per_s": 512 / (c4s["ms_per_solve"] * 1e-3),
                                                      "speedup_over_this_gpu": 512 / (c4s["ms_per_solve"] * 1e-3) / value,
                                                      "efficiency": 512 / (c4s["ms_per_solve"] * 1e-3) / value / 8,
                                                      "note": "512 columns / the slowest shard's solve, gather not included; NOT a measurement of 8 GPUs"}
                ex["c4_shard"] = c4s
                # the reference's shipped size and literals (spec:23-96: L = 800, N = 501, grd_alb = 1, tau*_atm = 0.104, EVA aerosol): one
                # column; the transport is the WIDE instantiation of the chunk-parallel kernel, eight workgroups per column
                ex["shipped"] = extra_case(O, dev, local_rank, 1, 800, 501, "eva", 5, [0], rho=1.0, tau_atm=0.104, check_orders=6)
                if a.aerosol != "hg":                    # rounds 1-2 measured the HG(0.7) stand-in: the same sweep shape, for continuity
                    ex["c4_hg"] = extra_case(O, dev, local_rank, 512, 200, 128, "hg", 5, [0])
                    ex["c4_hg"]["order_loop"] = "the library's choice (two column groups)"
            except Exception as e:                      # an extra must not take the headline line with it
                ex["error"] = "%s: %s" % (type(e).__name__, e)
            out["extras"] = ex
        if world == 1:
            out["strong"] = {"status": "unmeasured: one GPU.  With --gpus N this object holds the one-sweep-over-the-node measurement of "
                                       "BASELINE configs[3] (512 columns dealt to the ranks, fields gathered to rank 0)",
                             "single_gpu_proxy": "extras.c4_shard"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
    stuck = False
    if world > 1 and not strong:
        # The one-sweep-over-the-node measurement, in a thread the main thread waits for with a limit: its gather (the C ABI's
        # sosrt_gather under RCCL) has never run with more than one rank on this pool's one-GPU boxes, and a collective that does not
        # complete must cost the line its `strong` object, not the line.
        import threading
        box = {}

        def _strong():
            try:
                box["obj"] = measure_strong(a, dev, local_rank, rank, world, on_gpu)
            except Exception as e:                      # (on every rank alike, or the others run into the limit)
                box["obj"] = {"status": "failed: %s: %s" % (type(e).__name__, e)}

        th = threading.Thread(target=_strong, daemon=True)
        th.start()
        th.join(float(os.environ.get("SOSRT_BENCH_STRONG_LIMIT", "240")))
        stuck = th.is_alive()
        strong_obj = {"status": "timed out: the strong-scaling measurement did not complete within its limit"} if stuck else box.get("obj")
    if rank == 0:
        if world > 1 and strong_obj is not None:
            out["strong"] = strong_obj
        print(json.dumps(out), flush=True)
    if stuck:
        os._exit(0)                                     # (a collective is still pending: no orderly shutdown of the process group)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
