#!/usr/bin/env python3
"""A/B of the contraction's register-resident tile for the last few live columns (jn_gemm_tile.hpp: gemm_tile_lone,
SOSRT_PLAN_GEMM_LIVE16_REGS) against the staged live-column tilings: ms per solve and us per order for batches of B columns of
the headline sweep's shape, SOSRT_GEMM_REGS = 0 / <cap> alternating on one box; bits compared.
usage: tools/ab_gemm_regs.py [N [L [cap [B ...]]]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np
import torch

import bench


def run(B, L, N, cap, reps=3, steps=10, aerosol="eva", groups="1"):
    os.environ["SOSRT_GEMM_REGS"] = str(cap)
    os.environ["SOSRT_GROUPS"] = groups
    dev = torch.device("cuda", 0)
    w = bench.build_sweep(512, L, N, 0, 1, aerosol=aerosol)
    idx = np.linspace(0, 511, B).astype(int)
    w = bench.take(w, idx)
    ln = bench.Lane(w, dev, 0, 256)
    best = 1e9
    try:
        ln.solve(); torch.cuda.synchronize(dev)
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(steps):
                ln.solve()
            torch.cuda.synchronize(dev)
            best = min(best, (time.perf_counter() - t0) / steps)
        n = ln.n.cpu().numpy()
        return best * 1e3, int(n.max()), ln.I.clone(), n
    finally:
        ln.close()


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    cap = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    Bs = [int(x) for x in sys.argv[4:]] or [1, 2, 4, 8, 16, 32, 64, 512]
    import __graft_entry__ as ge
    ge.build()
    print("N=%d L=%d cap=%d    B   staged ms (us/order) x2   register tile ms (us/order) x2   same bits" % (N, L, cap))
    for B in Bs:
        g = "0" if B > 256 else "1"
        a = run(B, L, N, 0, groups=g)
        b = run(B, L, N, cap, groups=g)
        a2 = run(B, L, N, 0, groups=g)
        b2 = run(B, L, N, cap, groups=g)
        same = bool(torch.equal(a[2], b[2]) and np.array_equal(a[3], b[3]))
        o = max(a[1] - 1, 1)
        print("%20d   %8.3f (%6.1f) %8.3f (%6.1f)    %8.3f (%6.1f) %8.3f (%6.1f)   %s   max order %d" % (
            B, a[0], a[0] * 1e3 / o, a2[0], a2[0] * 1e3 / o, b[0], b[0] * 1e3 / o, b2[0], b2[0] * 1e3 / o, same, a[1]), flush=True)


if __name__ == "__main__":
    main()
