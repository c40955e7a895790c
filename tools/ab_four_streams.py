#!/usr/bin/env python3
"""What more than two column groups would buy on the headline sweep, measured without building them: the 512 columns as ONE handle
(the library's two groups of 256 on two streams) against TWO handles of 256 columns each solving side by side (each takes two
groups of 128: four streams), and against FOUR handles of 128 (eight streams) -- alternating, same box.  The fields are compared
bit for bit with the one-handle solve (a column's bits do not depend on its batch).
    python3 tools/ab_four_streams.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np
import torch
import bench
from concurrent.futures import ThreadPoolExecutor


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    os.environ.pop("SOSRT_GROUPS", None)
    dev = torch.device("cuda:0")
    w = bench.build_sweep(512, 200, 128, 0, 1, aerosol="eva")
    B = w["B"]
    sets = {}
    for parts, how in ((1, "blocks"), (2, "blocks"), (2, "dealt"), (4, "blocks"), (4, "dealt")):
        idx = [np.arange(B)[p::parts] if how == "dealt" else np.arange(B)[p * B // parts:(p + 1) * B // parts] for p in range(parts)]
        sets[(parts, how)] = (idx, [bench.Lane(bench.take(w, ix), dev, 0, 256) for ix in idx])
    pools = [ThreadPoolExecutor(max_workers=1) for _ in range(4)]

    def step(lanes):
        def go(ln):
            torch.cuda.set_device(0)
            ln.solve()
        futs = [pools[i].submit(go, ln) for i, ln in enumerate(lanes)]
        for f in futs:
            f.result()

    def timed(lanes, k):
        for _ in range(3):
            step(lanes)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(k):
            step(lanes)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / k * 1e3

    ref = None
    for rep in range(3):
        for key, (idx, lanes) in sets.items():
            ms = timed(lanes, steps)
            I = torch.empty((B, 200, 256), dtype=torch.float64, device=dev)
            for ix, ln in zip(idx, lanes):
                I[torch.from_numpy(ix).to(dev)] = ln.I
            if ref is None:
                ref = I.clone()
            same = bool(torch.equal(I, ref))
            print("round %d  %d handle(s), columns in %-6s  %.3f ms per sweep  (%6.1fk columns/s)  same bits as one handle: %s" % (
                rep, key[0], key[1], ms, B / ms, same), flush=True)


if __name__ == "__main__":
    main()
