#!/usr/bin/env python3
"""Phase timing of k_transport_scan from in-kernel cycle stamps (diagnostic build: SOSRT_CXXFLAGS=-DSOSRT_SCAN_STAMPS)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np, torch
import bench
from sosrt.solver import Solver
from sosrt._lib import lib, check
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w = bench.build_sweep(cols, 200, 128, 0, 1)
B, L, N = w["B"], w["L"], w["N"]
dev = torch.device("cuda", 0)
s = Solver(L, N, max_batch=B, max_orders=256)
s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
s.set_grid(w["mu"]); s.set_phase(w["P_atm"], w["P_aer"])
s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
              w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
P0a, P0r = bench.host_p0(w)
d_tau = torch.from_numpy(w["tau"]).to(dev); d_P0a = torch.from_numpy(P0a).to(dev); d_P0r = torch.from_numpy(P0r).to(dev)
d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev); d_n = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.zeros((B, 2, 16), dtype=torch.int64, device=dev)
for it in range(2):
    check(lib().sosrt_debug_stamps(s._h, ctypes.c_void_p(st.data_ptr())))
    s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr())
    torch.cuda.synchronize()
n = d_n.cpu().numpy(); x = st.cpu().numpy()
order = np.argsort(-n)[:4]
for b in list(order) + [int(np.argmin(n))]:
    for wv in range(2):
        t = x[b, wv]
        d = np.diff(t[:6])
        print("col %3d n=%2d %s wave: tables %6d  down %7d  barrier %6d  surface %5d  up %7d  barrier %6d | waiting: carried values %7d, stage %7d cycles (total %.1f us @2.4GHz)" % (
            b, n[b], "first" if wv == 0 else "last ", d[0], d[1], d[2], 0, d[3], d[4], t[6], t[7], (t[5] - t[0]) / 2400.0))
        print("      per chunk (%d chunks): before the carried value %5d, after it %5d cycles" % (2 * ((25 + 3 - (0 if wv == 0 else 3)) // 4), t[8] / max(1, 2 * ((25 + 3 - (0 if wv == 0 else 3)) // 4)), t[9] / max(1, 2 * ((25 + 3 - (0 if wv == 0 else 3)) // 4))))
