#!/usr/bin/env python3
"""Bit-reproducibility of repeated solves on one handle, with the device memory poisoned beforehand (diagnostic).
SYNC_AFTER_ZERO=0 shows what a fill on torch's default stream (handle 0) does under a solve on the handle's own stream.
env: LL, NN (shape), TRIALS, REPS, MAXO; argument: columns."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np, torch
import bench
from sosrt.solver import Solver
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)

def poison(gb=60, val=0x7ff8dead):
    # raw hipMalloc blocks filled with a NaN pattern, then freed: what the library's hipMalloc gets next
    ptrs = []
    for _ in range(gb):
        p = ctypes.c_void_p()
        if hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 30)) != 0: break
        hip.hipMemsetD32(p, ctypes.c_int(val), ctypes.c_size_t((1 << 30) // 4))
        ptrs.append(p)
    hip.hipDeviceSynchronize()
    for p in ptrs: hip.hipFree(p)
    return len(ptrs)

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L, N = int(os.environ.get("LL", "400")), int(os.environ.get("NN", "256"))
w = bench.build_sweep(cols, L, N, 0, 1)
B = w["B"]
for trial in range(int(os.environ.get("TRIALS", "3"))):
    print("poisoned GB:", poison(val=0x7ff8dead + trial), flush=True)
    s = Solver(L, N, max_batch=B, max_orders=int(os.environ.get("MAXO", "128")))
    s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    s.set_grid(w["mu"]); s.set_phase(w["P_atm"], w["P_aer"])
    s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
                  w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
    d_tau = torch.from_numpy(w["tau"]).to(dev)
    d_mu0 = torch.from_numpy(np.ascontiguousarray(w["mu0"])).to(dev)
    d_P0a = torch.empty((B, 2 * N), dtype=torch.float64, device=dev); d_P0r = torch.empty((B, 2 * N), dtype=torch.float64, device=dev)
    s.phase_p0_device("rayleigh", d_mu0.data_ptr(), d_P0a.data_ptr(), B)
    s.phase_p0_device("hg", d_mu0.data_ptr(), d_P0r.data_ptr(), B, g=0.7)
    d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    ref = None
    for rep in range(int(os.environ.get("REPS", "3"))):
        d_I.zero_()
        if os.environ.get("SYNC_AFTER_ZERO", "0") == "1":     # (round 2: needed, set_stream(0) meant a non-blocking stream; round 3: 0 = the default stream)
            torch.cuda.synchronize()
        s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr(), d_status=d_st.data_ptr())
        torch.cuda.synchronize()
        cur = (d_I.clone(), d_n.clone())
        if ref is None:
            ref = cur
            print("trial %d: status sum %d, n max %d, finite %s" % (trial, int(d_st.abs().sum()), int(d_n.max()), bool(torch.isfinite(d_I).all())), flush=True)
        else:
            same = torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1])
            if not same:
                diff = (ref[0] != cur[0])
                idx = diff.nonzero()
                print("trial %d rep %d: DIFFERENT in %d elements; first %s; columns %s" % (trial, rep, int(diff.sum()), idx[0].tolist(), torch.unique(idx[:, 0])[:20].tolist()), flush=True)
                print("   values", ref[0][tuple(idx[0])].item(), cur[0][tuple(idx[0])].item(), "n", ref[1][idx[0][0]].item(), cur[1][idx[0][0]].item())
            else:
                print("trial %d rep %d: same bits" % (trial, rep), flush=True)
    s.close(); del d_I, ref, cur
    torch.cuda.empty_cache()
