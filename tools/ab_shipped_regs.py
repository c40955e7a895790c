#!/usr/bin/env python3
"""One column at the reference's shipped size (L = 800, N = 501), EVA: the contraction's register-resident tile (SOSRT_GEMM_REGS=1)
against the staged tilings (=0), alternating; also L = 200 at N = 128 / 256 / 512 for reference.  usage: tools/ab_shipped_regs.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import torch

import bench


def main():
    import __graft_entry__ as ge
    ge.build()
    dev = torch.device("cuda", 0)
    for L, N in ((800, 501), (200, 512), (400, 256), (200, 128)):
        for regs in ("0", "1", "0", "1"):
            os.environ["SOSRT_GEMM_REGS"] = regs
            r = bench.extra_case(None, dev, 0, 1, L, N, "eva", 5, [])
            print("SOSRT_GEMM_REGS=%s L=%d N=%d: %.3f ms per solve, %d orders, %.1f us per order" % (regs, L, N, r["ms_per_solve"], r["max_order"] - 1, r["us_per_order"]), flush=True)


if __name__ == "__main__":
    main()
