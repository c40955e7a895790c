#!/usr/bin/env python3
"""Experiment: the bench sweep with its columns ordered by (expected) order count and solved as two column groups of unequal
size -- the long columns a group of their own, whose latency-bound chain of orders starts at time 0 and runs beside the
throughput-bound orders of the short columns.  Needs a -DSOSRT_DIAG build (SOSRT_GROUP_SPLIT): SOSRT_LIB=.../libsosrt_diag.so.
python3 tools/sorted_groups.py [columns] [steps] [aerosol]   -> table on stdout, n and the sweep's axes in gpurun_out/sorted_n.npz"""
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
from sosrt.dist import expected_orders

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
aerosol = sys.argv[3] if len(sys.argv) > 3 else "eva"
w = bench.build_sweep(cols, 200, 128, 0, 1, aerosol=aerosol)
B, L, N = w["B"], w["L"], w["N"]
dev = torch.device("cuda", 0)
P0a, P0r = bench.host_p0(w)
KNOBS = ("SOSRT_GROUPS", "SOSRT_GROUP_SPLIT", "SOSRT_GROUP_PRIO", "SOSRT_SPLIT_MIN")


def run(w, P0a, P0r, env, label, ref=None):
    for k in KNOBS:
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
    for i in range(steps + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr())
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[3:]) * 1e3
    same = "" if ref is None else (" same_bits=%s" % bool(torch.equal(ref, d_I)))
    print("%-64s %.3f ms (min %.3f)%s" % (label, ts.mean(), ts.min(), same), flush=True)
    n = d_n.cpu().numpy()
    s.close()
    return n, d_I


def permuted(order):
    return bench.take(w, order), P0a[order], P0r[order]


n, _ = run(w, P0a, P0r, {"SOSRT_GROUPS": "1"}, "grid order, one group")
run(w, P0a, P0r, {}, "grid order, library default (two halves)")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "sorted_n_%s_%d.npz" % (aerosol, B)), n=n, mu0=w["mu0"], taer=w["taer"], rho=w["rho"],
         tau_atm=w["tau_atm"], alb_aer=w["alb_aer"])
proxy = expected_orders(w["tau_atm"] + w["taer"], w["rho"])
keys = {"true n": n.astype(np.float64), "proxy": proxy}
print("n: min %d max %d; columns with n > 13: %d; rank correlation of the proxy with n: %.3f"
      % (n.min(), n.max(), int((n > 13).sum()), np.corrcoef(np.argsort(np.argsort(proxy)), np.argsort(np.argsort(n)))[0, 1]))
for name, key in keys.items():
    asc = np.argsort(key, kind="stable")
    ws, a, r = permuted(asc)
    _, ref = run(ws, a, r, {"SOSRT_GROUPS": "1"}, "ascending by %s, one group" % name)
    run(ws, a, r, {}, "ascending by %s, two halves" % name, ref)
    for T in (64, 96, 128, 160, 192, 224):
        if T >= B:
            continue
        run(ws, a, r, {"SOSRT_GROUPS": "2", "SOSRT_GROUP_SPLIT": str(B - T), "SOSRT_SPLIT_MIN": "2"},
            "ascending by %s, long group of %3d second" % (name, T), ref)
    desc = asc[::-1].copy()
    ws, a, r = permuted(desc)
    _, ref = run(ws, a, r, {"SOSRT_GROUPS": "1"}, "descending by %s, one group" % name)
    for T in (64, 96, 128, 160, 192, 224):
        if T >= B:
            continue
        run(ws, a, r, {"SOSRT_GROUPS": "2", "SOSRT_GROUP_SPLIT": str(T), "SOSRT_SPLIT_MIN": "2"},
            "descending by %s, long group of %3d first" % (name, T), ref)
