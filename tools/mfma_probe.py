#!/usr/bin/env python3
"""Runs the FP64-MFMA microbenchmark and one solve (for a rocprofv3 --pmc pass)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
from sosrt.solver import Solver
s = Solver(8, 8, max_batch=1, max_orders=1)
print("mfma TF", s.microbench(0))
