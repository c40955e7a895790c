"""Host-side logic that needs no GPU: the C ABI loads and exports every declared symbol, the
plan tables (trapezoid weights, folded phase matrices, extrapolation maps) match NumPy, the
argument checks raise the reference's exception types, the input builders match the goldens."""
import ctypes
import os
import re

import numpy as np
import pytest

import gpu_model as M
import sos_oracle as O
from sosrt import _lib, inputs
from sosrt.solver import Solver, fix_count
from util import assert_close, golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sosrt.h")).read()
    declared = set(re.findall(r"\b(sosrt_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.lib().sosrt_version() >= 100


@pytest.mark.parametrize("N", [32, 100, 128, 256, 501])
def test_plan_tables(N):
    s = Solver(10, N, device=-1)
    mu = inputs.direction_grid(N)
    s.set_grid(mu)
    y = np.random.default_rng(N).uniform(0.1, 1.0, 2 * N)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    assert s.plan_weights() @ y == pytest.approx(trapz(y, mu), rel=1e-13)
    P = np.random.default_rng(1).uniform(0.5, 2.0, (2 * N, 2 * N))
    s.set_phase(P)
    W = s.plan_fold(0)
    assert_close(y @ W, trapz(P[:, ::-1] * y, mu, axis=1), 1e-13, "fold")
    assert np.array_equal(W, M.fold_weights(P, mu))
    row = np.exp(-0.3 / np.maximum(-mu[:N], 1e-3)) * (1 + 0.1 * np.random.default_rng(2).standard_normal(N))
    for tau_ref in (0.05, 0.5, 2.0, 4.5):
        idx = fix_count(tau_ref, N)
        assert idx == O.a4b_count(tau_ref, N)
        if idx == 0:
            continue
        s0, ns, C = s.plan_fix_table(idx)
        ref = np.array([O.improved_limit_mu_down(row, mu[:N], N, idx, i) for i in range(idx)])
        got = C @ row[s0:s0 + ns]
        # 1e-10 of the radiance the values are extrapolated from (the row carries 10 % noise, which an extrapolation over
        # idx grid steps amplifies for the reference's np.polyfit and for the table alike) ...
        assert np.max(np.abs(got - ref)) <= 1e-10 * np.max(np.abs(row[s0:s0 + ns])), "fix table N=%d idx=%d" % (N, idx)
        # ... and element-wise 1e-10 on a smooth row, the shape of a radiance field next to mu = 0
        smooth = np.exp(-0.3 / np.maximum(-mu[:N], 1e-3))
        ref_s = np.array([O.improved_limit_mu_down(smooth, mu[:N], N, idx, i) for i in range(idx)])
        assert_close(C @ smooth[s0:s0 + ns], ref_s, 1e-10, "fix table, smooth row, N=%d idx=%d" % (N, idx))
    s.close()


def test_fix_count_matches_python_int_truncation():
    for N in range(4, 1025):
        for t in (0.0625, 0.06251, 1.0, 1.0001, 3.999, 4.0):
            assert fix_count(t, N) == O.a4b_count(t, N)


def test_argument_errors():
    with pytest.raises(ValueError):
        Solver(1, 32, device=-1)
    with pytest.raises(ValueError):
        Solver(10, 2, device=-1)
    with pytest.raises(ValueError, match="at most 2686 layers"):      # three values per layer in the LDS of the first-order kernel
        Solver(2687, 128, device=-1)
    Solver(2686, 128, device=-1).close()
    s = Solver(10, 8, device=-1)
    with pytest.raises(ValueError):
        s.set_grid(np.zeros(15))
    bad = inputs.direction_grid(8)
    bad[3] = 0.5
    with pytest.raises(ValueError):
        s.set_grid(bad)
    with pytest.raises(_lib.SosrtError):          # phase before grid
        s.set_phase(np.ones((16, 16)))
    s.set_grid(inputs.direction_grid(8))
    with pytest.raises(ValueError):
        s.set_phase(np.ones((16, 15)))
    with pytest.raises(_lib.SosrtError):          # host-only handle cannot compute
        s.set_columns_single_slab(0.5, 1.0, 0.2)
    s.close()


def test_input_builders():
    for path in golden("g5_phase_N32_*.npz"):
        d = np.load(path)
        N, mu, mu0 = int(d["N"]), d["mu"], float(d["mu0"])
        assert np.array_equal(inputs.direction_grid(N), mu)
        for nm, name, g in (("ray", "rayleigh", 0), ("hg07", "hg", 0.7), ("hg03", "hg", 0.3), ("iso", "iso", 0), ("fwc", "fwc", 0)):
            P0, P = inputs.phase_function(name, N, mu, mu0, g)
            assert_close(P0, d[nm + "_P0"], 1e-13, nm)
            assert_close(P, d[nm + "_P"], 1e-13, nm)
    d = np.load(golden("g5_phase_N128.npz")[0])
    for k, m0 in enumerate(d["mu0"]):          # the BASELINE angular resolution
        for nm, name, g in (("ray", "rayleigh", 0), ("hg07", "hg", 0.7), ("fwc", "fwc", 0)):
            assert_close(inputs.phase_function(name, 128, d["mu"], float(m0), g)[0] if k else
                         inputs._azimuth_averaged(inputs._scalar_phase(name, g)[0], d["mu"], float(m0))[0], d[nm + "_P0"][k], 1e-13, nm)
    with pytest.raises(ValueError):
        inputs.phase_function("nope", 8, inputs.direction_grid(8), 0.5)
    d = np.load(golden("g5_tau_profile.npz")[0])
    for i in range(int(d["n"])):
        ta, tr, z0, zu, zd, L = d["p%d" % i]
        assert_close(inputs.tau_profile(ta, tr, z0, zu, zd, int(L)), d["tau%d" % i], 1e-15, "tau")
        assert inputs.slab_indices(z0, zu, zd, int(L)) == O.slab_indices(z0, zu, zd, int(L))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sos-radiative-transfer_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "sos_oracle" not in txt and "oracle/" not in txt, f


def test_mie_series_and_log_normal_ensemble():
    """Own Mie series (miepython is absent: parity unpinned) against published values and identities."""
    from sosrt import mie
    trapz = getattr(np, "trapezoid", None) or np.trapz
    # Bohren & Huffman appendix A test case: m = 1.55, x = 2 pi 0.525 / 0.6328
    qe, qs, qb, g = mie.efficiencies(1.55 + 0j, 2 * np.pi * 0.525 / 0.6328)
    assert (round(qe, 5), round(qs, 5), round(qb, 5), round(g, 5)) == (3.10543, 3.10543, 2.92534, 0.63314)
    # Wiscombe (1979, NCAR/TN-140+STR, the MIEV0 test cases; the same numbers Bohren & Huffman-style codes are checked with):
    # (Q_ext, Q_sca, g) to six digits.  The absorbing cases are written m = n - ik there; this module's series takes n + ik.
    for m, x, (qe_, qs_, g_) in (((1.5 + 0j), 10.0, (2.881999, 2.881999, 0.742913)),
                                 ((0.75 + 0j), 10.0, (2.232265, 2.232265, 0.896473)),
                                 ((0.75 + 0j), 1000.0, (1.997908, 1.997908, 0.844944)),
                                 ((1.5 + 1j), 1.0, (2.336321, 0.663454, 0.192136)),
                                 ((1.5 + 1j), 100.0, (2.097502, 1.283697, 0.850252)),
                                 ((10 + 10j), 1.0, (2.532993, 2.049405, -0.110664)),
                                 ((10 + 10j), 100.0, (2.071124, 1.836785, 0.556215))):
        qe, qs, _, g = mie.efficiencies(m, x)
        assert (round(qe, 6), round(qs, 6), round(g, 6)) == (qe_, qs_, g_), (m, x, qe, qs, g)
    assert mie.efficiencies(0.75 + 0j, 0.099)[0] == pytest.approx(7.417859e-06, rel=1e-6)       # Rayleigh regime
    assert mie.efficiencies(1.5 + 1j, 0.055)[:2] == pytest.approx((0.101491, 1.1e-05), abs=5e-7)
    # the sign of Im(m): by default any imaginary part is absorption, so both ways of writing the wildfire aerosol's index
    # (README.md:109 writes 1.7 + 0.03j; miepython documents n - ik) are the same, absorbing, sphere; the literal conventions
    # are explicit options, and the wrong one is a gain medium (Q_abs < 0) -- which is why it is never the default
    a = mie.efficiencies(1.7 + 0.03j, 2.0)
    assert a == mie.efficiencies(1.7 - 0.03j, 2.0) == mie.efficiencies(1.7 + 0.03j, 2.0, "n+ik") == mie.efficiencies(1.7 - 0.03j, 2.0, "n-ik")
    assert a[0] > a[1] > 0                                                        # extinction > scattering: it absorbs
    wrong = mie.efficiencies(1.7 - 0.03j, 2.0, "n+ik")
    assert wrong[0] < wrong[1]                                                    # "absorption" < 0
    assert mie.refractive_index(1.7 - 0.03j) == 1.7 + 0.03j and mie.refractive_index(1.7 - 0.03j, "n+ik") == 1.7 - 0.03j
    with pytest.raises(ValueError):
        mie.refractive_index(1.5, "whatever")
    assert np.array_equal(inputs.phase_function("wildfire", 8, inputs.direction_grid(8), 0.5, indx=1.7 - 0.03j)[0],
                          inputs.phase_function("wildfire", 8, inputs.direction_grid(8), 0.5)[0])
    # Rayleigh limit: p ~ 1 + mu^2
    mu = np.linspace(-1, 1, 9)
    p = mie.i_unpolarized(1.5 + 0j, 0.02, mu)
    assert np.allclose(p / p[4], 1 + mu ** 2, rtol=2e-3)
    th = np.linspace(0, np.pi, 20001)
    for m, x in ((1.44 + 0j, 5.0), (1.7 + 0.03j, 2.0), (1.33 + 0.1j, 30.0)):
        qe, qs, _, g = mie.efficiencies(m, x)
        S1, _ = mie.amplitudes(m, x, np.array([1.0]))
        assert qe == pytest.approx(4 / x ** 2 * S1[0].real, rel=1e-12)            # optical theorem
        pv = mie.i_unpolarized(m, x, np.cos(th))
        assert trapz(pv * np.sin(th), th) * 2 * np.pi == pytest.approx(qs / qe, rel=1e-5)   # integrates to the albedo
        assert trapz(pv * np.cos(th) * np.sin(th), th) * 2 * np.pi / (qs / qe) == pytest.approx(g, rel=1e-5)
    assert mie.efficiencies(1.5 + 0j, 2000.0)[0] == pytest.approx(2.0, abs=0.03)    # extinction paradox
    # ensemble phase functions of the reference's two scenarios: normalisations of phase:103,131
    N = 16
    mug = inputs.direction_grid(N)
    for name in ("eva", "wildfire"):
        P0, P = inputs.phase_function(name, N, mug, 0.5)
        assert trapz(P0, mug) == pytest.approx(2.0, rel=1e-12)
        assert np.allclose(trapz(P, mug, axis=0), 4.0, rtol=1e-12)
        assert (P > 0).all() and (P0 > 0).all()
    P0e, _ = inputs.phase_function("eva", N, mug, 0.5)
    assert P0e[:N].sum() > 2 * P0e[N:].sum()        # micron-size sulphate scatters forward: sunlight keeps going down



def test_launch_plan_of_the_order_loop():
    """sosrt_plan_launch: the one function the order loop of sosrt_solve_dev decides with (csrc/api.hip: plan_order), on
    host-only handles -- over the direction count, the live count, the surface and the zone count."""
    T, G = _lib, _lib
    def plan(N, L, batch, live, order_loop=True, **kw):
        s = Solver(L, N, device=-1)
        s.set_grid(inputs.direction_grid(N))
        s.set_order_loop(order_loop)              # (off by default: measured slower than two launches per order)
        p = s.plan_launch(batch, live, **kw)
        s.close()
        return p
    assert plan(128, 200, 1, 1, order_loop=False)["order_loop"] == 0 and plan(128, 200, 512, 20, order_loop=False)["order_loop"] == 0
    # BASELINE C4 shape (L = 200, N = 128), the 512-column sweep: two column groups of 256
    p = plan(128, 200, 512, 512)
    assert (p["groups"], p["gemm"], p["transport"], p["order_loop"]) == (2, G.PLAN_GEMM_DENSE, T.PLAN_TRANSPORT_RING, 0)
    p = plan(128, 200, 512, 230)                    # a few converged (more than 60 % of the group live): the dense tiling skips their tiles
    assert (p["gemm"], p["tail_cols"], p["transport"]) == (G.PLAN_GEMM_DENSE, 0, T.PLAN_TRANSPORT_RING)
    p = plan(128, 200, 512, 180)                    # ... and writes the live list: the transport's choice follows the live count, not the group's size
    assert (p["gemm"], p["tail_cols"], p["transport"], p["parts"]) == (G.PLAN_GEMM_DENSE, 0, T.PLAN_TRANSPORT_SCAN, 1)
    p = plan(128, 200, 512, 150)                    # contraction over the live columns, chunk-parallel transport
    assert (p["gemm"], p["tail_cols"], p["transport"], p["parts"], p["order_loop"]) == (G.PLAN_GEMM_LIVE32, 150, T.PLAN_TRANSPORT_SCAN, 1, 0)
    p = plan(128, 200, 512, 100)                    # two workgroups per column while they fit the CUs
    assert (p["transport"], p["parts"], p["order_loop"]) == (T.PLAN_TRANSPORT_SCAN, 2, 0)
    p = plan(128, 200, 512, 60)                     # the group's share is 128 CUs: 60 columns x 1 workgroup <= half of it
    assert (p["order_loop"], p["ol_parts"], p["ol_grid"]) == (1, 1, 128)
    p = plan(128, 200, 512, 30)                     # ... x 2 workgroups
    assert (p["order_loop"], p["ol_parts"], p["ol_grid"]) == (1, 2, 128)
    # the last few live columns: the register-resident 16-row tile while its workgroups make at most a round and a half (13
    # columns at N = 128, 6 at N = 256); N = 32 rounds up to half a register block: the staged tile
    assert [plan(128, 200, 512, k)["gemm"] for k in (40, 14, 13, 1)] == [G.PLAN_GEMM_LIVE32, G.PLAN_GEMM_LIVE32_DEEP, G.PLAN_GEMM_LIVE16_REGS, G.PLAN_GEMM_LIVE16_REGS]
    assert [plan(256, 200, 8, k)["gemm"] for k in (7, 6, 1)] == [G.PLAN_GEMM_LIVE32_DEEP, G.PLAN_GEMM_LIVE16_REGS, G.PLAN_GEMM_LIVE16_REGS]
    assert plan(501, 800, 1, 1)["gemm"] == G.PLAN_GEMM_LIVE32_DEEP      # the shipped size: 408 such workgroups for ONE column, slower (144 against 133 us per order)
    assert plan(512, 200, 3, 3)["gemm"] == G.PLAN_GEMM_LIVE16_REGS and plan(256, 400, 3, 3)["gemm"] == G.PLAN_GEMM_LIVE16_REGS
    assert plan(32, 50, 1, 1)["gemm"] == G.PLAN_GEMM_LIVE32_DEEP and plan(64, 50, 1, 1)["gemm"] == G.PLAN_GEMM_LIVE16_REGS
    # BASELINE C2 / C3: one column is in the order-loop launch from the second order on, on the whole device
    assert (plan(128, 200, 1, 1)["order_loop"], plan(128, 200, 1, 1)["ol_parts"], plan(128, 200, 1, 1)["ol_grid"]) == (1, 2, 256)
    p = plan(256, 200, 1, 1)
    assert (p["transport"], p["parts"], p["order_loop"], p["ol_parts"]) == (T.PLAN_TRANSPORT_SCAN, 4, 1, 4)
    assert plan(256, 200, 40, 40)["order_loop"] == 0 and plan(256, 200, 40, 32)["order_loop"] == 1     # 4 x 40 > 128 >= 4 x 32
    # a batch of up to 48 columns is one column group; the 64-column shard of the C4 sweep on one of 8 GPUs is two groups of 32
    p = plan(128, 200, 40, 40)
    assert (p["groups"], p["order_loop"], p["ol_parts"], p["ol_grid"]) == (1, 1, 2, 256)
    p = plan(128, 200, 64, 32)
    assert (p["groups"], p["order_loop"], p["ol_parts"], p["ol_grid"]) == (2, 1, 2, 128)
    assert plan(128, 200, 200, 65)["order_loop"] == 0 and plan(128, 200, 200, 64)["ol_parts"] == 1      # one workgroup per column
    # a Lambertian surface couples all directions at the ground: one workgroup per column, in both kernels
    p = plan(128, 200, 40, 40, surface="lambertian")
    assert (p["transport"], p["parts"], p["order_loop"], p["ol_parts"]) == (T.PLAN_TRANSPORT_SCAN, 1, 1, 1)
    p = plan(256, 200, 8, 8, surface="lambertian")  # ... which N = 256 does not have: ring kernel, no order-loop launch
    assert (p["transport"], p["order_loop"]) == (T.PLAN_TRANSPORT_RING, 0)
    # N <= 64: one lane group, one workgroup per column
    p = plan(32, 50, 1, 1)
    assert (p["transport"], p["parts"], p["order_loop"], p["ol_parts"]) == (T.PLAN_TRANSPORT_SCAN, 1, 1, 1)
    # the reference's shipped size (odd N, N > 256, 100 chunks per sweep): the chunk-parallel kernel's WIDE instantiation, eight
    # workgroups per column, for up to four rounds of them; beyond that the register-streaming kernel + repair pass; a Lambertian
    # surface (no split form) takes that one too; N = 70: the general kernel
    p = plan(501, 800, 1, 1)
    assert (p["transport"], p["parts"], p["repair"], p["order_loop"]) == (T.PLAN_TRANSPORT_SCAN, 8, 0, 0)
    assert (plan(501, 800, 128, 128)["transport"], plan(501, 800, 128, 128)["parts"]) == (T.PLAN_TRANSPORT_SCAN, 8)
    p = plan(501, 800, 400, 150)                    # two groups of 200, 150 live: 8 x 150 workgroups are more than four rounds
    assert (p["transport"], p["repair"]) == (T.PLAN_TRANSPORT_FAST, 1)
    assert plan(501, 800, 1, 1, surface="lambertian")["transport"] == T.PLAN_TRANSPORT_FAST
    assert (plan(300, 96, 2, 2)["transport"], plan(300, 96, 2, 2)["parts"]) == (T.PLAN_TRANSPORT_SCAN, 5)
    # N = 70, 129, 257: the rewritten mu -> 0- directions straddle two waves of a half row, which no wave-independent kernel takes --
    # but part 0 of the split form holds them in one wave: chunk-parallel kernel; the general kernel where there is no split form
    assert (plan(70, 50, 4, 4)["transport"], plan(70, 50, 4, 4)["parts"]) == (T.PLAN_TRANSPORT_SCAN, 2)
    assert (plan(257, 64, 2, 2)["transport"], plan(257, 64, 2, 2)["parts"]) == (T.PLAN_TRANSPORT_SCAN, 5)
    assert plan(70, 50, 4, 4, surface="lambertian")["transport"] == T.PLAN_TRANSPORT_GENERAL
    # columns of more than three zones (two aerosol layers): ring-class kernels, dense contraction, no order-loop launch
    p = plan(128, 200, 40, 20, zones=5)
    assert (p["gemm"], p["transport"], p["order_loop"]) == (G.PLAN_GEMM_DENSE, T.PLAN_TRANSPORT_SCAN, 0)
    # ... at the shipped size and at an odd N as well (the WIDE instantiation with the whole zone table)
    assert (plan(501, 800, 2, 2, zones=7)["transport"], plan(501, 800, 2, 2, zones=7)["parts"]) == (T.PLAN_TRANSPORT_SCAN, 8)
    assert (plan(201, 120, 3, 3, zones=5)["transport"], plan(201, 120, 3, 3, zones=5)["parts"]) == (T.PLAN_TRANSPORT_SCAN, 4)
    # a smaller device: the launch is sized by its CUs
    p = plan(128, 200, 16, 16, cus=64)
    assert (p["order_loop"], p["ol_parts"], p["ol_grid"]) == (1, 2, 64)
    assert plan(128, 200, 40, 40, cus=64)["order_loop"] == 0
    with pytest.raises(ValueError):
        plan(128, 200, 4, 5)
