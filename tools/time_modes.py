#!/usr/bin/env python3
"""Wall time of a solve under the transport kernels that can take a shape: python3 tools/time_modes.py L N B [modes...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np, torch
L, N, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
modes = sys.argv[4:] or ["auto", "fast", "general"]
from sosrt import inputs
from sosrt.solver import Solver
mu = inputs.direction_grid(N)
P_atm = inputs.phase_function_device("rayleigh", N, mu, 0.5)[1]
P_aer = inputs.phase_function_device("hg", N, mu, 0.5, 0.7)[1]
rng = np.random.default_rng(1)
mu0 = rng.uniform(0.3, 1.0, B); taer = rng.choice([0.05, 0.12, 0.5], B); rho = rng.uniform(0.0, 0.6, B)
iu, idn = inputs.slab_indices(120, 25, 17, L)
tau = np.stack([inputs.tau_profile(0.124, t, 120, 25, 17, L) for t in taer])
dev = torch.device("cuda", 0)
ref = None
for mode in modes:
    os.environ["SOSRT_TRANSPORT"] = mode
    s = Solver(L, N, max_batch=B, max_orders=128)
    st = torch.cuda.Stream(device=dev); s.set_stream(st.cuda_stream)
    s.set_grid(mu); s.set_phase(P_atm, P_aer)
    s.set_columns(np.full(B, iu), np.full(B, idn), mu0, rho, 1.0, 0.97, 0.124 / L, taer / (idn + 1 - iu), 0.124 + taer)
    d_tau = torch.from_numpy(tau).to(dev); d_mu0 = torch.from_numpy(mu0).to(dev)
    d_P0a = torch.empty((B, 2 * N), dtype=torch.float64, device=dev); d_P0r = torch.empty_like(d_P0a)
    torch.cuda.synchronize()
    s.phase_p0_device("rayleigh", d_mu0.data_ptr(), d_P0a.data_ptr(), B); s.phase_p0_device("hg", d_mu0.data_ptr(), d_P0r.data_ptr(), B, g=0.7)
    d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev); d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = d_n.cpu().numpy(); I = d_I.cpu().numpy()
    if ref is None: ref = I
    print("L=%d N=%d B=%d %-8s %8.3f ms  orders %d (max %d)  max |I - first mode| / max %.2e" % (L, N, B, mode, dt * 1e3, int((n - 1).sum()), n.max(), np.max(np.abs(I - ref)) / np.max(np.abs(ref))))
    s.close()
