#!/usr/bin/env python3
"""One column at the reference's shipped size (L = 800, N = 501, spec:33,57), EVA aerosol: ms per solve and us per order with the
default launch plan (chunk-parallel transport, WIDE instantiation: eight workgroups per column) and with the register-streaming
kernel of rounds 1-3 (SOSRT_TRANSPORT=fast).  usage: tools/time_shipped.py [L [N]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import torch

import bench


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 501
    import __graft_entry__ as ge
    ge.build()
    dev = torch.device("cuda", 0)
    for mode in ("auto", "fast", "auto", "fast"):
        os.environ["SOSRT_TRANSPORT"] = mode
        r = bench.extra_case(None, dev, 0, 1, L, N, "eva", 5, [])
        print("%-5s L=%d N=%d: %.3f ms per solve, %d orders, %.1f us per order" % (mode, L, N, r["ms_per_solve"], r["max_order"] - 1, r["us_per_order"]), flush=True)


if __name__ == "__main__":
    main()
