#!/usr/bin/env python3
"""How do the two kernels of an order scale with the CUs they are given?  A solve of B columns of the headline sweep on a stream
created with hipExtStreamCreateWithCUMask over Y of the 256 CUs (every (256/Y)-th bit of the mask), one column group: total
contraction / transport time per solve (HIP events of the handle's profiling) and ms per solve; then two handles of B columns
each on complementary masks, side by side.  The question behind it: would an order loop whose contraction and transport launches
live on disjoint sets of CUs overlap them (on shared CUs they do not)?
usage: tools/cu_mask_probe.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np
import torch

os.environ["SOSRT_GROUPS"] = "1"
import bench

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[0] * 8)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return st


def lane(B, stream):
    dev = torch.device("cuda", 0)
    w = bench.build_sweep(512, 200, 128, 0, 1, aerosol="eva")
    w = bench.take(w, np.linspace(0, 511, B).astype(int))
    ln = bench.Lane(w, dev, 0, 256)
    torch.cuda.synchronize(dev)
    if stream is not None:
        ln.s.set_stream(stream.value)
    return ln


def timed(lanes, steps=6):
    for ln in lanes:
        ln.solve()
    for ln in lanes:
        ln.s.synchronize()
    for ln in lanes:
        ln.s.profile_enable(True); ln.s.profile_reset()
    import threading

    def work(ln):                                        # (sosrt_solve_dev holds its host thread for the whole order loop)
        for _ in range(steps):
            ln.solve()
        ln.s.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(ln,)) for ln in lanes]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = (time.perf_counter() - t0) / steps * 1e3
    out = []
    for ln in lanes:
        g, t = ln.s.profile_get(0), ln.s.profile_get(1)
        out.append((g[0] / steps, t[0] / steps))
        ln.s.profile_enable(False)
    return dt, out


def main():
    # (the mask is honoured for a contiguous range of bits: bits [0, Y) are Y CUs spread evenly over the eight XCCs; a sparse pattern
    # enables every CU -- tools/cu_mask_where.hip)
    for B, Y in ((512, 256), (512, 192), (512, 128), (256, 256), (256, 192), (256, 128), (256, 96), (256, 64), (128, 64), (128, 32)):
        st = None if Y == 256 else masked_stream(range(0, Y))
        ln = lane(B, st)
        dt, k = timed([ln])
        print("%4d columns on %3d CUs: %.3f ms per solve, contraction %.3f ms, transport %.3f ms" % (B, Y, dt, k[0][0], k[0][1]), flush=True)
        ln.close()
    # two handles side by side: same CUs, then disjoint parts
    for name, m0, m1 in (("both on all CUs", None, None), ("CUs 0..127 / 128..255", range(0, 128), range(128, 256)),
                         ("CUs 0..191 / 64..255 (128 shared)", range(0, 192), range(64, 256))):
        a = lane(256, None if m0 is None else masked_stream(m0))
        b = lane(256, masked_stream(range(256)) if m1 is None else masked_stream(m1))
        dt, k = timed([a, b])
        print("2 x 256 columns, %s: %.3f ms per pair of solves (contraction %.3f + %.3f, transport %.3f + %.3f)" % (
            name, dt, k[0][0], k[1][0], k[0][1], k[1][1]), flush=True)
        a.close(); b.close()


if __name__ == "__main__":
    main()
