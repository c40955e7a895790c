#!/usr/bin/env python3
"""fp64 -> fp32 tolerance study for the source-function contraction (BASELINE configs[4] / SURVEY 8d C5:
"fp64 -> fp32 mixed with tolerance study").

The one place where a lower precision would buy throughput is Jn = diag(c) In_1 W (FP64 MFMA: 49 TFLOP/s
measured; the FP32 / split-BF16 matrix paths are several times faster).  This script reruns the whole order
loop of a column with that contraction done four ways and reports what reaches the converged radiance field:

    fp64        the shipped arithmetic (operands and accumulation in double)
    fp32/fp64   operands rounded to float, products accumulated in double ("fp32 storage")
    fp32        operands and accumulation in float (an FP32 MFMA)
    2xfp32      operands split into float hi + float lo, three products accumulated in double
                (what a split scheme could reach with exact accumulation)
    2xfp32/f32  the same split with each product accumulated in float, as an FP32 MFMA does: the
                accumulator, not the operand width, is then the limit

It is CPU-only test infrastructure (NumPy + the oracle); nothing here is on the product path.

    python tests/study_mixed_precision.py [L N n_columns]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sos_oracle as O            # noqa: E402
import gpu_model as M             # noqa: E402


def contraction(mode):
    def f32(x):
        return x.astype(np.float32)

    def mm(A, W):
        if mode == "fp64":
            return A @ W
        if mode == "fp32/fp64":
            return f32(A).astype(np.float64) @ f32(W).astype(np.float64)
        if mode == "fp32":
            return (f32(A) @ f32(W)).astype(np.float64)
        if mode == "2xfp32":
            Ah, Wh = f32(A), f32(W)
            Al, Wl = f32(A - Ah), f32(W - Wh)
            d = np.float64
            return Ah.astype(d) @ Wh.astype(d) + Ah.astype(d) @ Wl.astype(d) + Al.astype(d) @ Wh.astype(d)
        if mode == "2xfp32/f32":
            Ah, Wh = f32(A), f32(W)
            Al, Wl = f32(A - Ah), f32(W - Wh)
            d = np.float64
            return (Ah @ Wh).astype(d) + (Ah @ Wl).astype(d) + (Al @ Wh).astype(d)
        raise ValueError(mode)
    return mm


def solve(c, mode):
    """solve_column with the source function as the folded contraction in the given arithmetic."""
    mm = contraction(mode)
    Wa, Wr = M.fold_weights(c.P_atm, c.mu), M.fold_weights(c.P_aer, c.mu)
    L = len(c.tau)
    rows = np.arange(L)
    slab = (rows >= c.idx_up) & (rows <= c.idx_down)
    ca = np.where(slab, (c.alb_atm / 4) * c.f_atm, c.alb_atm / 4)
    cr = np.where(slab, (c.alb_aer / 4) * c.f_aer, 0.0)
    orig = O.source_function
    O.source_function = lambda col, In_1: ca[:, None] * mm(In_1, Wa) + cr[:, None] * mm(In_1, Wr)
    try:
        return O.solve_column(c, literal=False)
    finally:
        O.source_function = orig


def study(L=200, N=128, ncol=3, out=sys.stdout):
    mu = O.make_mu(N)
    rng = np.random.default_rng(20250905)
    res = {}
    for i in range(ncol):
        mu0 = float(rng.uniform(0.2, 1.0))
        taer = float(10 ** rng.uniform(-2, 0))
        rho = float(rng.uniform(0, 0.8))
        P0a, Pa = O.phase_rayleigh(N, mu, mu0)
        P0r, Pr = O.phase_hg(N, mu, mu0, 0.7)
        c = O.make_column(mu0, 120, 25, 17, L, 0.124, taer, rho, 1.0, 0.97, N, P0a, Pa, P0r, Pr)
        ref = solve(c, "fp64")
        scale = np.abs(ref.I).max()
        for mode in ("fp32/fp64", "fp32", "2xfp32", "2xfp32/f32"):
            s = solve(c, mode)
            err = np.abs(s.I - ref.I).max() / scale if s.n == ref.n else float("nan")
            r = res.setdefault(mode, {"err": 0.0, "dn": 0})
            r["err"] = max(r["err"], err) if err == err else r["err"]
            r["dn"] += int(s.n != ref.n)
        print("column %d: mu0=%.3f tau*_aer=%.3f rho=%.2f  n=%d" % (i, mu0, taer, rho, ref.n), file=out)
    print("\n| contraction arithmetic | max |dI| / max I over %d columns (L=%d, N=%d) | columns whose order count changed |" % (ncol, L, N), file=out)
    print("|---|---|---|", file=out)
    for mode, r in res.items():
        print("| %s | %.2e | %d |" % (mode, r["err"], r["dn"]), file=out)
    return res


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]]
    study(*a)
