#!/usr/bin/env python3
"""Throughput of the SOS hot path: columns/s to 1e-4 convergence at (N_tau=200, N_mu=128).

One step = one pass of the hot path (I1 -> [Jn -> In] x orders, convergence test included) over
one batch of synthetic columns whose inputs are already resident in HBM.  The workload is the
BASELINE C4 sweep shape: 512 columns = 8 mu0 x 8 tau*_aer x 8 grd_alb, L=200, N=128, Rayleigh
atmosphere + HG(g=0.7) aerosol stand-in (the EVA log-normal Mie phase function needs miepython,
unavailable offline), specular surface, fp64.  With N GPUs every rank solves its own 512-column
sweep (weak scaling; rank r uses a different aerosol single-scattering albedo) and the TOA /
surface radiances and order counts are gathered to rank 0 over RCCL once per step.

Prints ONE JSON line on rank 0 (see the contract in the task description).
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


def build_sweep(n_columns, L, N, rank, world):
    from sosrt import inputs
    side = max(1, round(n_columns ** (1 / 3)))
    mu0 = np.linspace(0.2, 1.0, side)
    taer = np.geomspace(0.01, 1.0, side)
    rho = np.linspace(0.0, 0.8, side)
    M0, TA, RH = (x.ravel()[:n_columns] for x in np.meshgrid(mu0, taer, rho, indexing="ij"))
    B = len(M0)
    mu = inputs.direction_grid(N)
    iu, idn = inputs.slab_indices(120, 25, 17, L)
    tau_atm = 0.124
    alb_aer = float(np.linspace(0.97, 0.90, world)[rank]) if world > 1 else 0.97
    tau = np.stack([inputs.tau_profile(tau_atm, t, 120, 25, 17, L) for t in TA])
    P_atm = inputs.phase_function("rayleigh", N, mu, 0.5)[1]
    P_aer = inputs.phase_function("hg", N, mu, 0.5, 0.7)[1]
    cache = {}
    for m in np.unique(M0):
        cache[float(m)] = (inputs.phase_function("rayleigh", N, mu, m)[0], inputs.phase_function("hg", N, mu, m, 0.7)[0])
    P0a = np.stack([cache[float(m)][0] for m in M0])
    P0r = np.stack([cache[float(m)][1] for m in M0])
    return dict(B=B, L=L, N=N, mu=mu, tau=tau, P_atm=P_atm, P_aer=P_aer, P0a=P0a, P0r=P0r, mu0=M0, taer=TA, rho=RH,
                idx_up=iu, idx_down=idn, tau_atm=tau_atm, alb_aer=alb_aer)


def cpu_baseline(w, seconds_budget=25.0):
    """The oracle in its literal (reference-cost) mode on one host core, on a bounded sample of the
    same sweep: columns are taken evenly across the sweep until the budget is used."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sos_oracle as O
    B = w["B"]
    order = np.linspace(0, B - 1, min(B, 8)).astype(int)
    done, orders, t0 = 0, 0, time.perf_counter()
    for b in order:
        col = O.Column(tau=w["tau"][b], mu=w["mu"], N=w["N"], idx_up=w["idx_up"], idx_down=w["idx_down"], mu0=float(w["mu0"][b]),
                       grd_alb=float(w["rho"][b]), alb_atm=1.0, alb_aer=w["alb_aer"], dtau_atm=w["tau_atm"] / w["L"],
                       dtau_aer=float(w["taer"][b]) / (w["idx_down"] + 1 - w["idx_up"]),
                       tauStar_tot=w["tau_atm"] + float(w["taer"][b]), P0_atm=w["P0a"][b], P_atm=w["P_atm"],
                       P0_aer=w["P0r"][b], P_aer=w["P_aer"])
        s = O.solve_column(col, literal=True)
        done += 1
        orders += s.n - 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "columns/s", "cores": 1, "kind": "port",
            "sample": "%d columns of the same sweep (evenly spaced), %d orders, oracle literal mode, %.1f s" % (done, orders, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--columns", type=int, default=512, help="columns per GPU")
    ap.add_argument("--layers", type=int, default=200)
    ap.add_argument("--angles", type=int, default=128)
    ap.add_argument("--max-orders", type=int, default=256)
    ap.add_argument("--inflight", type=int, default=1, help="independent solves in flight (own handle + stream each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipelined", type=int, default=3,
                    help="after the timed region, also measure the throughput with this many steps in flight on "
                         "separate streams (0: skip); reported beside the headline value, never instead of it")
    a = ap.parse_args()

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
    from sosrt.solver import Solver

    w = build_sweep(a.columns, a.layers, a.angles, rank, world)
    B, L, N = w["B"], w["L"], w["N"]
    D = 2 * N
    d_tau = torch.from_numpy(w["tau"]).to(dev)
    d_P0a = torch.from_numpy(w["P0a"]).to(dev)
    d_P0r = torch.from_numpy(w["P0r"]).to(dev)

    # `inflight` independent solves may be in flight at once, each on its own handle and HIP stream:
    # the order loop of a sweep ends in a long tail of launches over its few slowest-converging
    # columns, which leaves most CUs idle; the next sweep's dense launches fill them.
    class Lane:
        def __init__(self):
            self.stream = torch.cuda.Stream(device=dev)
            self.s = Solver(L, N, max_batch=B, max_orders=a.max_orders, device=local_rank)
            self.s.set_stream(self.stream.cuda_stream)
            self.s.set_grid(w["mu"])
            self.s.set_phase(w["P_atm"], w["P_aer"])
            self.s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
                               w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
            self.I = torch.empty((B, L, D), dtype=torch.float64, device=dev)
            self.n = torch.zeros(B, dtype=torch.int32, device=dev)
            self.st = torch.zeros(B, dtype=torch.int32, device=dev)
            self.done = torch.cuda.Event()

        def solve(self):
            """Enqueue one step on this lane's stream; returns (digest tensor or None, completion event)."""
            self.s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), self.I.data_ptr(), tol=1e-4,
                                d_n_orders=self.n.data_ptr(), d_status=self.st.data_ptr())
            dig = None
            if world > 1:
                with torch.cuda.stream(self.stream):
                    dig = torch.cat([self.I[:, 0, N:], self.I[:, L - 1, :N], self.n.to(torch.float64)[:, None]], dim=1).contiguous()
            ev = torch.cuda.Event()
            ev.record(self.stream)
            return dig, ev

    torch.cuda.synchronize(dev)
    lanes = [Lane() for _ in range(max(1, a.inflight))]
    main_stream = torch.cuda.current_stream(dev)

    from concurrent.futures import ThreadPoolExecutor
    execs = [ThreadPoolExecutor(max_workers=1) for _ in lanes]       # a lane runs its steps in order

    def lane_step(i):
        torch.cuda.set_device(local_rank)
        return lanes[i].solve()

    def run_steps(k):
        """k steps dealt round-robin to the lanes.  With several ranks the per-column digests (TOA-up row,
        surface-down row, order count) of every step are gathered to rank 0 -- the only collective --
        from this thread, in step order, behind the step's completion event."""
        futs = [execs[step % len(lanes)].submit(lane_step, step % len(lanes)) for step in range(k)]
        for f in futs:
            dig, ev = f.result()
            if world > 1:
                main_stream.wait_event(ev)
                if backend != "nccl":                 # gloo gathers host tensors
                    ev.synchronize()
                    dig = dig.cpu()
                bufs = [torch.empty_like(dig) for _ in range(world)] if rank == 0 else None
                dist.gather(dig, bufs, dst=0)

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
    gemm_ms = tr_ms = fo_ms = 0.0
    gemm_launches = tr_launches = 0
    for ln in lanes:
        ms, cnt = ln.s.profile_get(_lib.K_GEMM); gemm_ms += ms; gemm_launches += cnt
        ms, cnt = ln.s.profile_get(_lib.K_TRANSPORT); tr_ms += ms; tr_launches += cnt
        fo_ms += ln.s.profile_get(_lib.K_FIRST)[0]
        ln.s.profile_enable(False)

    # Beside the headline (one step at a time, one stream: every kernel timing above is of a launch that
    # has the GPU to itself): the same steps with `--pipelined` of them in flight on separate streams and
    # handles.  The tail of one sweep's order loop then overlaps the dense launches of the next.
    pipe = None
    if a.pipelined > 1 and len(lanes) == 1 and world == 1:
        while len(lanes) < a.pipelined:
            lanes.append(Lane())
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

    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = world * B * a.steps / dt
        # dominant kernel: the Jn contraction.  Algorithmic flops per launch group = 2 L D^2 per live
        # (column, order) pair (no credit for the second slab matrix or for padding).
        flops = 2.0 * L * D * D * orders_per_step * a.steps
        achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        tr_gbs = 40.0 * L * D * orders_per_step * a.steps / (tr_ms * 1e-3) / 1e9 if tr_ms > 0 else 0.0
        out = {
            "metric": "SOS columns/sec to 1e-4 convergence (Ntau=200, Nmu=128)",
            "value": value, "unit": "columns/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4 sweep: %d columns/GPU = mu0 x tau*_aer x grd_alb grid, L=%d, N=%d (D=%d), "
                                   "Rayleigh atm + HG(0.7) aerosol stand-in, specular surface, tol 1e-4" % (B, L, N, D),
                       "columns_per_gpu": B, "orders_per_step": orders_per_step, "max_order": int(n_host.max()),
                       "not_converged": int((st_host != 0).sum()), "inflight_solves": max(1, a.inflight),
                       "parallelism": "columns sharded x%d, gather only" % world},
            "roofline": None, "roofline_other": None,
            "kernel_ms_per_step": {"k_jn_gemm": gemm_ms / a.steps, "k_transport": tr_ms / a.steps,
                                   "k_first_order": fo_ms / a.steps},
        }
        r_gemm = {"bound": "mfma", "kernel": "k_jn_gemm", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                  "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                  "avg_launch_ms": gemm_ms / max(gemm_launches, 1), "launches": gemm_launches,
                  "total_ms_per_step": gemm_ms / a.steps}
        # HBM-bound: reads Jn, E, I and writes In, I = 40 L D bytes per column.order
        r_tr = {"bound": "hbm", "kernel": "k_transport_ring", "achieved": tr_gbs, "peak": 8000.0, "unit": "GB/s",
                "frac": tr_gbs / 8000.0, "traffic": None, "avg_launch_ms": tr_ms / max(tr_launches, 1),
                "launches": tr_launches, "total_ms_per_step": tr_ms / a.steps}
        # HBM bytes per launch from the committed PMC passes of the same workload (profiles/README.md): the
        # counters cannot be read from inside this process
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if B == 512 and L == 200 and N == 128 and world == 1:
                r_gemm["traffic"] = pmc["k_jn_gemm"]["hbm_bytes_per_launch"]
                r_tr["traffic"] = pmc["k_transport_ring"]["hbm_bytes_per_launch"]
                r_gemm["traffic_unit"] = r_tr["traffic_unit"] = "bytes/launch (rocprofv3 PMC, profiles/r01_pmc_traffic.json)"
        except (OSError, KeyError, ValueError):
            pass
        # the roofline object is the kernel with the larger share of the timed region
        out["roofline"], out["roofline_other"] = (r_gemm, r_tr) if gemm_ms >= tr_ms else (r_tr, r_gemm)
        if pipe:
            out["pipelined"] = pipe
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
