#!/usr/bin/env python3
"""A/B of the order-loop kernel (csrc/order_loop.hip) against the two-launch order loop: ms per solve and us per order for
batches of B columns of the headline sweep's shape, SOSRT_ORDER_LOOP=0 / 1 alternating on one box.
usage: tools/ab_order_loop.py [N [L [B ...]]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np
import torch

import bench


def run(B, L, N, on, reps=3, steps=10, aerosol="eva"):
    os.environ["SOSRT_ORDER_LOOP"] = "1" if on else "0"
    os.environ["SOSRT_GROUPS"] = "1"
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
        st = ln.s.order_loop_stats(True)
        return best * 1e3, int(n.max()), int((n - 1).sum()), st, ln.I.clone(), n
    finally:
        ln.close()


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    Bs = [int(x) for x in sys.argv[3:]] or [1, 4, 16, 32, 64, 96, 128]
    import __graft_entry__ as ge
    ge.build()
    print("N=%d L=%d   B   two-launch ms (us/order)   order-loop ms (us/order)   launches/refused/col.orders  same bits" % (N, L))
    for B in Bs:
        a = run(B, L, N, False)
        b = run(B, L, N, True)
        same = bool(torch.equal(a[4], b[4]) and np.array_equal(a[5], b[5]))
        print("%14d   %8.3f (%6.1f)          %8.3f (%6.1f)          %s   %s   max order %d  sum %d" % (
            B, a[0], a[0] * 1e3 / max(a[1] - 1, 1), b[0], b[0] * 1e3 / max(b[1] - 1, 1), b[3], same, a[1], a[2]), flush=True)


if __name__ == "__main__":
    main()
