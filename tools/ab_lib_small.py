#!/usr/bin/env python3
"""ms per solve (and a digest of the field and the order counts) for small batches of the headline sweep's columns with the library
named by SOSRT_LIB (default: the tree's): run it once per build, alternating, to A/B two builds.  usage: tools/ab_lib_small.py [N [B ...]]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np
import torch

import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Bs = [int(x) for x in sys.argv[2:]] or [1, 8, 64]
dev = torch.device("cuda", 0)
for B in Bs:
    w = bench.build_sweep(512, 200, N, 0, 1, aerosol="eva")
    w = bench.take(w, np.linspace(0, 511, B).astype(int))
    ln = bench.Lane(w, dev, 0, 256)
    ln.solve(); torch.cuda.synchronize(dev)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            ln.solve()
        torch.cuda.synchronize(dev)
        best = min(best, (time.perf_counter() - t0) / 10)
    n = ln.n.cpu().numpy()
    dig = hashlib.sha256(ln.I.cpu().numpy().tobytes() + n.tobytes()).hexdigest()[:12]
    print("%-22s N=%d B=%3d: %.4f ms per solve, %.2f us per order, digest %s" % (os.environ.get("SOSRT_LIB", "tree")[-20:], N, B, best * 1e3, best * 1e6 / max(int(n.max()) - 1, 1), dig), flush=True)
    ln.close()
