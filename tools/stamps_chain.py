#!/usr/bin/env python3
"""Per-chunk timeline of the chunk-parallel transport kernel for a lone column (diagnostic build: SOSRT_CXXFLAGS=-DSOSRT_SCAN_CHAIN).
For every chunk of the two sweeps: when its wave entered it, had its stage, finished the chunk-local work, received the carried
value (= published the next one) and finished; and the hop from one carried value to the next."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np, torch
import bench
from sosrt.solver import Solver
from sosrt._lib import lib, check
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w = bench.build_sweep(1, 200, N, 0, 1, aerosol=os.environ.get("AEROSOL", "eva"))
B, L = w["B"], w["L"]
dev = torch.device("cuda", 0)
s = Solver(L, N, max_batch=B, max_orders=256)
s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
s.set_grid(w["mu"]); s.set_phase(w["P_atm"], w["P_aer"])
s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
              w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
P0a, P0r = bench.host_p0(w)
d_tau = torch.from_numpy(w["tau"]).to(dev); d_P0a = torch.from_numpy(P0a).to(dev); d_P0r = torch.from_numpy(P0r).to(dev)
d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev); d_n = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.zeros((B, 4096), dtype=torch.int64, device=dev)
for it in range(2):
    check(lib().sosrt_debug_stamps(s._h, ctypes.c_void_p(st.data_ptr())))
    s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr())
    torch.cuda.synchronize()
x = st.cpu().numpy()[0]
NCH = (L + 7) // 8
parts = (N + 63) // 64
for part in range(parts):
    t = x[part * 640:(part * 128 + 2 * NCH) * 5].reshape(2 * NCH, 5).astype(np.int64)
    t0 = t[t > 0].min()
    print("part %d: chunk  wave | enter  stage   pre  carry   done | hop   (cycles from the part's first stamp; wave = chunk mod 8)" % part)
    prev = None
    for q in range(2 * NCH):
        r = t[q] - t0
        hop = (r[3] - prev) if prev is not None and q != NCH else 0
        print("   %s %2d    %d  | %6d %6d %6d %6d %6d | %5d" % ("dn" if q < NCH else "up", q % NCH, (q % NCH) % 8, r[0], r[1], r[2], r[3], r[4], hop))
        prev = r[3]
    print("  downward sweep %d cycles, upward %d" % (t[NCH - 1][4] - t[0][0], t[2 * NCH - 1][4] - t[NCH][0]))
