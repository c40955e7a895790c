#!/usr/bin/env python3
"""What more zones cost: a batch of two-layer (five-zone) columns -- the zone-table instantiation of the ring / chunk-parallel
kernels; round 2: the general kernel, 2.16 vs 0.50 ms at 64 columns -- against the same batch with one layer (three zones).  python3 tools/time_zones.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np
from sosrt import _lib
from sosrt.main import SOS_Aer_layers, get_solver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(5)
mu0 = rng.uniform(0.3, 1.0, B); rho = rng.uniform(0.0, 0.6, B)
for name, slabs in (("one layer (3 zones)", [(25, 17, 0.32, 0.97)]), ("two layers (5 zones)", [(60, 50, 0.12, 0.90), (25, 17, 0.20, 0.97)])):
    r = SOS_Aer_layers(mu0, rho, slabs, nb_layers=200, nb_angles=128, aer_phase_fun="hg", g_aer=0.7)      # warm-up, builds the solver
    s = get_solver(200, 128, B, 256)
    s.profile_enable(True); s.profile_reset()
    r = SOS_Aer_layers(mu0, rho, slabs, nb_layers=200, nb_angles=128, aer_phase_fun="hg", g_aer=0.7)
    g_ms, g_n = s.profile_get(_lib.K_GEMM); t_ms, t_n = s.profile_get(_lib.K_TRANSPORT)
    s.profile_enable(False)
    print("%-22s B=%d  contraction %.3f ms in %d launches, transport %.3f ms in %d launches (HIP events)  orders %d (max %d)" % (
        name, B, g_ms, g_n, t_ms, t_n, int((r.n - 1).sum()), r.n.max()))
