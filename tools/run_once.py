#!/usr/bin/env python3
"""One solve of the bench sweep (for rocprofv3 runs): python3 tools/run_once.py [columns] [solves] [angles]
(environment: AEROSOL = eva | wildfire | hg, default eva as in bench.py; LAYERS, default 200)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np
import torch

os.environ.setdefault("SOSRT_GROUPS", "1")      # one column group as in bench.py's headline: every kernel alone on the GPU

import bench
from sosrt.solver import Solver

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 512
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NANG = int(sys.argv[3]) if len(sys.argv) > 3 else 128
w = bench.build_sweep(cols, int(os.environ.get("LAYERS", "200")), NANG, 0, 1, aerosol=os.environ.get("AEROSOL", "eva"))
B, L, N = w["B"], w["L"], w["N"]
dev = torch.device("cuda", 0)
s = Solver(L, N, max_batch=B, max_orders=256)
s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
s.set_grid(w["mu"]); s.set_phase(w["P_atm"], w["P_aer"])
s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
              w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
P0a, P0r = bench.host_p0(w)
d_tau = torch.from_numpy(w["tau"]).to(dev); d_P0a = torch.from_numpy(P0a).to(dev); d_P0r = torch.from_numpy(P0r).to(dev)
d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev)
d_n = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(solves):
    s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr())
    torch.cuda.synchronize()
print("orders", int((d_n.cpu().numpy() - 1).sum()), "max", int(d_n.max().item()))
n = d_n.cpu().numpy()
print("active columns per order (2..max):", [int((n >= k).sum()) for k in range(2, int(n.max()) + 1)])
