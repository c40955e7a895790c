#!/usr/bin/env python3
"""fp64 -> fp32 tolerance study of the Jn contraction ON THE DEVICE (BASELINE configs[4]).

The same sweep is solved twice through the product path: with the default v_mfma_f64 contraction and with the opt-in
float contraction (sosrt_set_contraction: float operands, v_mfma_f32 with a float accumulator; transport, running total
and convergence test stay fp64).  Reported: error of the converged field, order counts, time of the contraction
launches.  Run on the GPU box:  python tests/study_mixed_precision_gpu.py > profiles/r02_mixed_precision_gpu.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np


def study(columns=512, L=200, N=128, out=sys.stdout, steps=3):
    import torch
    import bench
    from sosrt import _lib
    from sosrt.solver import Solver
    w = bench.build_sweep(columns, L, N, 0, 1)
    B = w["B"]
    dev = torch.device("cuda", 0)
    P0a, P0r = bench.host_p0(w)
    d_tau = torch.from_numpy(w["tau"]).to(dev); d_P0a = torch.from_numpy(P0a).to(dev); d_P0r = torch.from_numpy(P0r).to(dev)
    res = {}
    for mode in ("f64", "f32"):
        s = Solver(L, N, max_batch=B, max_orders=256)
        s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        s.set_grid(w["mu"]); s.set_phase(w["P_atm"], w["P_aer"])
        s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
                      w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
        s.set_contraction(mode)
        d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev)
        d_n = torch.zeros(B, dtype=torch.int32, device=dev)
        for it in range(steps + 1):
            if it == 1:
                s.profile_enable(True); s.profile_reset()
                torch.cuda.synchronize(); t0 = time.perf_counter()
            s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr())
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        gemm_ms, launches = s.profile_get(_lib.K_GEMM)
        res[mode] = dict(I=d_I.cpu().numpy(), n=d_n.cpu().numpy(), ms=dt * 1e3, gemm_ms=gemm_ms / steps, launches=launches // steps)
        s.close()
    a, b = res["f64"], res["f32"]
    scale = np.max(np.abs(a["I"]), axis=(1, 2), keepdims=True)
    err = np.max(np.abs(a["I"] - b["I"]) / scale, axis=(1, 2))
    sig = np.abs(a["I"]) > 1e-9 * scale
    rel = np.max(np.where(sig, np.abs(a["I"] - b["I"]) / np.maximum(np.abs(a["I"]), 1e-300), 0.0), axis=(1, 2))
    print("sweep: %d columns, L=%d, N=%d (D=%d)" % (B, L, N, 2 * N), file=out)
    print("orders: sum %d (f64) vs %d (f32), columns whose order count differs: %d" % (
        int((a["n"] - 1).sum()), int((b["n"] - 1).sum()), int((a["n"] != b["n"]).sum())), file=out)
    print("converged field, f32 contraction vs f64 contraction: max |dI| / max I = %.3e (median over columns %.3e); "
          "max element-wise relative error where |I| > 1e-9 max I = %.3e" % (err.max(), np.median(err), rel.max()), file=out)
    d = np.abs(a["I"] - b["I"]) / scale
    big = d > 1e-5
    bb, tt, mm = np.nonzero(big)
    print("elements more than 1e-5 away: %d of %d (%.2e), all in %d of the %d columns.  One application of the contraction to the "
          "same field differs by 1e-6 at most (step-level test); the isolated large deviations build up over the orders and are "
          "consistent with the data-dependent second-difference search of spec:403-406 (absolute threshold 1e-4): a 1e-7 "
          "perturbation can move where a search stops, the blended directions of that row then change at the 1e-2 level and "
          "the next orders spread the change."
          % (int(big.sum()), d.size, big.mean(), len(np.unique(bb)), B), file=out)
    print("99.9th percentile of |dI| / max I over all elements: %.3e" % np.quantile(d, 0.999), file=out)
    print("parity bar of the north star: 1e-10  ->  the float contraction misses it by a factor %.0f" % (rel.max() / 1e-10), file=out)
    print("contraction launches per solve: %.3f ms (f64, %d launches: live-column tilings) vs %.3f ms (f32, %d launches: dense tiling only)"
          % (a["gemm_ms"], a["launches"], b["gemm_ms"], b["launches"]), file=out)
    print("solve: %.3f ms (f64) vs %.3f ms (f32)" % (a["ms"], b["ms"]), file=out)
    return dict(err=float(err.max()), med=float(np.median(err)), p999=float(np.quantile(d, 0.999)), rel=float(rel.max()), n_equal=bool((a["n"] == b["n"]).all()), gemm_f64=a["gemm_ms"], gemm_f32=b["gemm_ms"])


if __name__ == "__main__":
    study(512, 200, 128)
    print()
    study(256, 400, 256)
