#!/usr/bin/env python3
"""Experiment: the bench sweep with its columns sorted by order count, solved as two column groups -- the short
columns on the caller's stream, the long ones on the internal (optionally high-priority) stream.
python3 tools/sorted_groups.py [columns] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np
import torch

import bench
from sosrt.solver import Solver

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w = bench.build_sweep(cols, 200, 128, 0, 1)
B, L, N = w["B"], w["L"], w["N"]
dev = torch.device("cuda", 0)
P0a, P0r = bench.host_p0(w)


def run(w, P0a, P0r, env, label, ref=None):
    for k in ("SOSRT_GROUPS", "SOSRT_GROUP_SPLIT", "SOSRT_GROUP_PRIO", "SOSRT_GEMM_PAD_LDS", "SOSRT_GROUP_RING_SLOTS", "SOSRT_SPLIT_MIN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    s = Solver(L, N, max_batch=B, max_orders=256)
    s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    s.set_grid(w["mu"]); s.set_phase(w["P_atm"], w["P_aer"])
    s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
                  w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
    d_tau = torch.from_numpy(w["tau"]).to(dev); d_P0a = torch.from_numpy(P0a).to(dev); d_P0r = torch.from_numpy(P0r).to(dev)
    d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    ts = []
    for i in range(steps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr())
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[2:]) * 1e3
    same = "" if ref is None else (" same_bits=%s" % bool(torch.equal(ref, d_I)))
    print("%-58s %.3f ms (min %.3f)%s" % (label, ts.mean(), ts.min(), same), flush=True)
    n = d_n.cpu().numpy()
    s.close()
    return n, d_I


n, _ = run(w, P0a, P0r, {}, "grid order, one group")
order = np.argsort(n, kind="stable")
ws = bench.take(w, order)
P0a_s, P0r_s = P0a[order], P0r[order]
ns, ref = run(ws, P0a_s, P0r_s, {}, "sorted by n, one group")
print("n sorted: min %d max %d; columns with n > 13: %d" % (ns.min(), ns.max(), int((ns > 13).sum())))
for T in (64, 96, 128, 160, 200, 256):
    for prio in (0, 1):
        for pad in ("27008", "0"):
            env = {"SOSRT_GROUPS": "2", "SOSRT_GROUP_SPLIT": str(B - T), "SOSRT_GROUP_PRIO": str(prio), "SOSRT_GEMM_PAD_LDS": pad,
                   "SOSRT_GROUP_RING_SLOTS": "2" if pad != "0" else "0", "SOSRT_SPLIT_MIN": "2"}
            run(ws, P0a_s, P0r_s, env, "sorted, long group %3d, prio %d, pad %s" % (T, prio, pad), ref)
