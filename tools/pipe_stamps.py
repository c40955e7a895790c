#!/usr/bin/env python3
"""Busy cycles per wave of k_transport_pipe (diagnostic): python3 tools/pipe_stamps.py [columns]"""
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
st = torch.zeros((B, 16), dtype=torch.int64, device=dev)
for it in range(2):
    check(lib().sosrt_debug_stamps(s._h, ctypes.c_void_p(st.data_ptr())))
    s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr())
    torch.cuda.synchronize()
n = d_n.cpu().numpy(); x = st.cpu().numpy()
# the stamps hold the LAST order each column ran through the pipeline kernel; the slowest columns ran it nearly alone
names = ["chain0", "chain1", "load0", "load1", "treat", "store0", "store1", "store2", "store3"]
for b in list(np.argsort(-n)[:3]):
    ticks = 2 * ((L + 7) // 8) + 2
    print("col %3d n=%2d total %d cycles (%.1f us at 2.4 GHz), %d ticks -> %.0f cycles/tick; busy cycles/tick by wave:" % (
        b, n[b], x[b, 15], x[b, 15] / 2400.0, ticks, x[b, 15] / ticks))
    print("   " + "  ".join("%s %.0f" % (names[i], x[b, i] / ticks) for i in range(9)))
