#!/usr/bin/env python3
"""Measure the machine peaks the roofline fractions are quoted against (FP64 MFMA, FP64 FMA,
streaming copy) on the current device."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
from sosrt.solver import Solver  # noqa: E402

s = Solver(8, 8, max_batch=1, max_orders=1)
print(json.dumps({"mfma_f64_tflops": s.microbench(0), "copy_gbs_read_plus_write": s.microbench(1),
                  "fma_f64_tflops": s.microbench(2)}))
