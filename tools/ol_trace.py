#!/usr/bin/env python3
"""Event log of order-loop launches (-DSOSRT_OL_STAMPS build): B columns of the headline sweep, two solves, the log of the last one.
usage: SOSRT_CXXFLAGS=-DSOSRT_OL_STAMPS tools/ol_trace.py B [out dir]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
B = int(sys.argv[1])
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
log = os.path.join(out, "ol_log_%d.txt" % B)
if os.path.exists(log):
    os.remove(log)
os.environ["SOSRT_ORDER_LOOP"] = "1"
os.environ["SOSRT_GROUPS"] = "1"
os.environ["SOSRT_OL_LOG"] = log
import numpy as np
import torch

import bench
import __graft_entry__ as ge
ge.build()
dev = torch.device("cuda", 0)
w = bench.build_sweep(512, 200, 128, 0, 1, aerosol="eva")
w = bench.take(w, np.linspace(0, 511, B).astype(int))
ln = bench.Lane(w, dev, 0, 256)
for _ in range(3):
    ln.solve()
    torch.cuda.synchronize(dev)
ln.close()
subprocess.call([sys.executable, os.path.join(ROOT, "tools", "ol_timeline.py"), log])
