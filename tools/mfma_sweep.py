#!/usr/bin/env python3
"""FP64 MFMA rate vs independent accumulators per wave and workgroups (4 waves) per CU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
from sosrt.solver import Solver
s = Solver(8, 8, max_batch=1, max_orders=1)
for nacc_code, nacc in ((0, 2), (1, 4), (2, 8), (3, 16)):
    print("acc/wave %2d:" % nacc, "  ".join("wg/CU %d: %5.1f TF" % (bpc, s.microbench(10 + nacc_code * 10 + bpc)) for bpc in (1, 2, 4, 8)))
