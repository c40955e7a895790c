"""Pins the CPU oracle (oracle/sos_oracle.py) to fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import sos_oracle as O
from util import assert_close, column_case, g1_case, golden, oracle_column

PIN = 2e-13   # the oracle is a restatement: it must agree with the reference to rounding


@pytest.mark.parametrize("path", golden("g1_*.npz"), ids=lambda p: p.split("/")[-1][3:-4])
def test_single_slab_steps(path):
    d, N, P = g1_case(path)
    tau, mu, tS, mu0, alb = d["tau"], d["mu"], float(d["tauStar"]), float(d["mu0"]), float(d["alb"])
    assert_close(O.I1_NumInt(tau, mu, tS, mu0, d["P0"], alb, N), d["I1"], PIN, "I1")
    In_1, n = d["I1"], 2
    assert str(d["error"]) == ""
    while "In_%d" % n in d:
        assert_close(O.Jn_NumInt(n, In_1, tau, mu, tS, mu0, P, alb, N), d["Jn_%d" % n], PIN, "Jn")
        for lit in (True, False):
            In = O.In_NumInt(n, d["Jn_%d" % n], In_1, tau, mu, tS, mu0, P, alb, N, literal=lit)
            assert_close(In, d["In_%d" % n], PIN, "In literal=%s" % lit)
        In_1 = d["In_%d" % n]
        n += 1


def test_helpers():
    d = np.load(golden("g2_helpers.npz")[0])
    for i in range(int(d["n_asym"])):
        tt, mu, r = d["a%d_par" % i]
        v = O.improved_asymptotic_downward_radiance(d["a%d_J" % i], d["a%d_tau" % i], tt, mu)
        assert v == pytest.approx(r, rel=1e-14, abs=0), i
    assert O.improved_asymptotic_downward_radiance(np.zeros(0), np.zeros(0), 0.1, -2e-3) == float(d["asym_empty"])
    for i in range(int(d["n_lim"])):
        N, idx = (int(x) for x in d["l%d_Nidx" % i])
        row, mud = d["l%d_row" % i], np.linspace(-1, 0, N)
        v = np.array([O.improved_limit_mu_down(row, mud, N, idx, k) for k in range(idx)])
        w = np.array([O.limit_mu_down(row, mud, N, idx, k) for k in range(idx)])
        assert np.array_equal(v, d["l%d_vals" % i]) and np.array_equal(w, d["l%d_lin" % i])
    for N, m1, m2 in d["mu_approx"]:
        assert O.mu_approx_In(O.make_mu(int(N)), int(N)) == (int(m1), int(m2))
    assert (O.MU_THRESHOLD, O.MU_EXTREME_THRESHOLD, O.MU_VERY_SMALL_THRESHOLD) == tuple(d["thresholds"])


@pytest.mark.parametrize("path", golden("g3_*.npz") + golden("g6_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_columns(path):
    d, c = column_case(path)
    col = oracle_column(O, c)
    assert np.array_equal(col.tau, c["tau"]) and (col.idx_up, col.idx_down) == (c["idx_up"], c["idx_down"])
    I1 = O.first_order(col)
    assert_close(I1, d["I_saved"][0], PIN, "I1")
    # the Lambertian fixtures come from a modified reference (H1 bypass): only n >= 2 is pinned, from its I1
    seed = d["I_saved"][0] if c["surface"] == "lambertian" else None
    for lit in (True, False):
        s = O.solve_column(col, literal=lit, I1=seed)
        assert s.n == c["n"]
        assert_close(s.I, d["I"], PIN * 50, "I literal=%s" % lit)
        for k in range(s.n):
            assert_close(s.I_saved[k], d["I_saved"][k], PIN * 50, "order %d" % (k + 1))


@pytest.mark.parametrize("path", golden("g3_spec_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_first_order_yardstick_in_long_double(path):
    """`first_order_extended` (the formulas of spec:113-292 with long-double intermediates on the same inputs: the yardstick for
    the rounding noise of the reference's own arithmetic, see its docstring) against the reference's first order.  The golden
    'mu0node' column has mu0 ON a node (the limit form of spec:126 applies); the others are far from any node."""
    d, c = column_case(path)
    col = oracle_column(O, c)
    ext = O.first_order_extended(col)
    assert ext.dtype == np.float64 and np.finfo(np.longdouble).nmant >= 63
    assert_close(ext, d["I_saved"][0], 1e-10, "I1 in long double vs the reference")
    assert np.max(np.abs(ext - d["I_saved"][0])) <= 2e-13 * np.max(np.abs(ext))


def test_zone_table_reduces_to_the_three_zones():
    """The zone-table generalisation with the reference's three zones is the same arithmetic, bit for bit;
    two aerosol layers run, stay positive and converge (more than one slab is parity unpinned: the reference has one)."""
    d, c = column_case(golden("g3_spec_L60_N64_ray_hg_mu0node.npz")[0])
    col = oracle_column(O, c)
    a = O.solve_column(col, literal=False)
    L = c["L"]
    tab = [O._Zone(0, c["idx_up"] - 1, "atm"), O._Zone(c["idx_up"], c["idx_down"], "mix", c["alb_aer"], c["dtau_aer"]),
           O._Zone(c["idx_down"] + 1, L - 1, "atm")]
    import dataclasses
    b = O.solve_column(dataclasses.replace(col, zone_table=tab), literal=False)
    assert a.n == b.n and np.array_equal(a.I, b.I)
    tau1, rows = O.tau_profile_slabs(c["tauStar_atm"], [(c["z_up"], c["z_down"], c["tauStar_aer"])], c["z0"], L)
    assert np.array_equal(tau1, col.tau) and rows == [(c["idx_up"], c["idx_down"])]
    two = O.make_column_slabs(0.6, 120, [(60, 50, 0.2, 0.9), (25, 17, 0.4, 0.97)], 48, 0.124, 0.3, 1.0, 64,
                              c["P0_atm"], c["P_atm"], c["P0_aer"], c["P_aer"])
    assert len(two.zones) == 5 and [z.kind for z in two.zones] == ["atm", "mix", "atm", "mix", "atm"]
    s2 = O.solve_column(two, literal=False)
    assert 5 < s2.n < 60 and np.all(s2.I[:, 1:64 - 1] >= 0) and np.isfinite(s2.I).all()
    # literal and vectorised transport agree on the table too
    s3 = O.solve_column(two, literal=True)
    assert s3.n == s2.n
    assert_close(s3.I, s2.I, PIN * 50, "two slabs, literal vs vectorised")


def test_c2_digest():
    """C2 shape (L=200, N=128, Rayleigh + HG 0.7): digests only, vectorised oracle."""
    d, c = column_case(golden("g4_spec_C2_*.npz")[0])
    s = O.solve_column(oracle_column(O, c), literal=False)
    N, L = c["N"], c["L"]
    assert s.n == c["n"]
    assert_close(s.I[0, N:], d["toa_up"], 1e-12, "TOA up")
    assert_close(s.I[L - 1, :N], d["sfc_down"], 1e-12, "surface down")
    assert_close(s.I.sum(axis=0), d["col_sum"], 1e-12, "column sums")
    assert_close(s.I_saved.reshape(s.n, -1).sum(axis=1), d["order_sum"], 1e-12, "order sums")


def test_inputs():
    tab = np.load(golden("g5_fwc_table.npz")[0])
    mt, pt = tab["mu_fwc"], tab["phase_func_FWC"]
    for path in golden("g5_phase_N32_*.npz"):
        d = np.load(path)
        N, mu, mu0 = int(d["N"]), d["mu"], float(d["mu0"])
        for nm, (P0, P) in (("ray", O.phase_rayleigh(N, mu, mu0)), ("hg07", O.phase_hg(N, mu, mu0, 0.7)),
                            ("hg03", O.phase_hg(N, mu, mu0, 0.3)), ("iso", O.phase_isotropic(N, mu)),
                            ("fwc", O.phase_table(N, mu, mu0, mt, pt))):
            assert_close(P0, d[nm + "_P0"], 1e-13, nm)
            assert_close(P, d[nm + "_P"], 1e-13, nm)
    # the BASELINE angular resolution: P0 for four mu0, P through digests
    d = np.load(golden("g5_phase_N128.npz")[0])
    N, mu = int(d["N"]), d["mu"]
    rows = [0, 1, N - 2, N - 1, N, N + 1, 2 * N - 2, 2 * N - 1]
    for nm, fn in (("ray", lambda m0: O.phase_rayleigh(N, mu, m0)), ("hg07", lambda m0: O.phase_hg(N, mu, m0, 0.7)),
                   ("fwc", lambda m0: O.phase_table(N, mu, m0, mt, pt))):
        for k, m0 in enumerate(d["mu0"]):
            P0, P = fn(float(m0))
            assert_close(P0, d[nm + "_P0"][k], 1e-13, "%s P0 mu0=%g" % (nm, m0))
            if m0 == 0.5:
                assert_close(P[rows], d[nm + "_P_rows"], 1e-13, nm)
                assert_close(P.sum(axis=0), d[nm + "_P_colsum"], 1e-13, nm)
                assert_close(np.diag(P), d[nm + "_P_diag"], 1e-13, nm)
                assert_close(np.diag(P[:, ::-1]), d[nm + "_P_anti"], 1e-13, nm)
    d = np.load(golden("g5_tau_profile.npz")[0])
    for i in range(int(d["n"])):
        ta, tr, z0, zu, zd, L = d["p%d" % i]
        assert np.array_equal(O.tau_profile(ta, tr, z0, zu, zd, int(L)), d["tau%d" % i])


@pytest.mark.parametrize("path", golden("g7_*.npz"), ids=lambda p: p.split("/")[-1][12:-4])
def test_epilogue(path):
    """Fluxes, diffusivity, heating rate (graphe), TOA net flux, forcing and critical albedo (crit)."""
    d, c = column_case(path)
    I, mu, tau, N, L, mu0, rho = d["I"], c["mu"], c["tau"], c["N"], c["L"], c["mu0"], c["grd_alb"]
    z = np.linspace(c["z0"], 0, L)
    assert_close(O.diffusivity(I, mu), d["diffusivity"], 1e-14, "diffusivity")
    assert_close(O.net_flux(I, mu, tau, N, mu0, rho), d["flux_net_F0"], 1e-13, "net flux")
    fd, fu = O.fluxes(I, mu, tau, N, mu0, rho, beam_norm="graphe")
    assert np.array_equal(fd, d["flux_down_F0"]) and np.array_equal(fu, d["flux_up_F0"])
    fd, fu = O.fluxes(I, mu, tau, N, mu0, rho, beam_norm="crit")
    assert np.array_equal(fd, d["flux_down_4pi"]) and np.array_equal(fu, d["flux_up_4pi"])
    assert np.array_equal(fd, d["crit_flux_down"]) and np.array_equal(fu, d["crit_flux_up"])
    assert np.array_equal(O.heating_rate(I, mu, tau, N, mu0, rho, z, c["idx_up"], c["idx_down"]), d["heating_rate"])
    assert O.toa_net_flux(I, mu, tau, N, mu0, rho) == float(d["crit_net_flux_toa"])
    # the oracle's own column gives the same net flux to rounding, the coded forcing is exactly zero
    col = oracle_column(O, c)
    s = O.solve_column(col, literal=False)
    assert s.n == int(d["crit_n"])
    assert O.toa_net_flux(s.I, mu, tau, N, mu0, rho) == pytest.approx(float(d["crit_net_flux_toa"]), rel=1e-12)
    if L * N <= 2000:
        assert O.radiative_forcing(col) == float(d["crit_delta_F_coded"]) == 0.0
        assert O.critical_albedo(lambda w: 0.0) == float(d["crit_critical_albedo_coded"]) == 0.5
        for w in (0.5, 0.75, 0.875):
            cw = oracle_column(O, dict(c, alb_aer=w))
            f = O.toa_net_flux(O.solve_column(cw, literal=False).I, mu, tau, N, mu0, rho)
            assert f == pytest.approx(float(d["crit_net_flux_toa_alb%g" % w]), rel=1e-12)
    base = oracle_column(O, dict(c, tauStar_aer=0.0))
    sb = O.solve_column(base, literal=False)
    assert sb.n == int(d["n_no_aerosol"])
    assert O.toa_net_flux(sb.I, base.mu, base.tau, N, mu0, rho) == pytest.approx(float(d["crit_net_flux_toa_no_aerosol"]), rel=1e-12)
    assert O.radiative_forcing(col, baseline=base) == pytest.approx(
        float(d["crit_net_flux_toa"]) - float(d["crit_net_flux_toa_no_aerosol"]), rel=1e-10)


def test_fp32_contraction_misses_the_parity_bar():
    """The tolerance study behind the choice of an FP64 contraction (tests/study_mixed_precision.py):
    float operands put the converged field 1e-9..1e-8 away, a float accumulator 1e-7; a hi + lo
    operand split only helps if the products are accumulated in double, which no FP32 matrix
    instruction does."""
    import io
    import study_mixed_precision as S
    res = S.study(L=40, N=32, ncol=2, out=io.StringIO())
    assert res["fp32/fp64"]["err"] > 1e-9 and res["fp32"]["err"] > 1e-8
    assert res["2xfp32"]["err"] < 1e-12 and res["2xfp32/f32"]["err"] > 1e-9


def test_readme_lambertian_first_order_properties():
    """PARITY UNPINNED (the reference has no runnable Lambertian first order).  The restatement of
    README.md:126-171 reduces to the coded first order over a black ground, its ground terms are linear in the albedo,
    and the removable singularity at mu' = mu is continuous (the limit agrees with the integrand next to it)."""
    from sosrt import inputs
    N, L, mu0 = 32, 40, 0.55
    mu = inputs.direction_grid(N)
    P0a, Pa = inputs.phase_function("rayleigh", N, mu, mu0)
    P0r, Pr = inputs.phase_function("hg", N, mu, mu0, 0.7)

    def fo(rho, readme=True):
        c = O.make_column(mu0, 120, 25, 17, L, 0.124, 0.3, rho, 1.0, 0.9, N, P0a, Pa, P0r, Pr, surface="lambertian_readme")
        return O.first_order_lambertian_readme(c) if readme else O.first_order(c)
    base = fo(0.0)
    np.testing.assert_array_equal(base, fo(0.0, readme=False))
    a, b = fo(0.2) - base, fo(0.4) - base
    assert np.isfinite(a).all() and (a >= 0).all() and a.max() > 0
    # ground terms: the reflected-beam integral is linear in rho, the reflected downward first order adds a rho^2 part
    # through the upward boundary only; the downward half is exactly linear
    np.testing.assert_allclose(b[:, :N], 2 * a[:, :N], rtol=1e-12, atol=0)
    assert (b[:, N + 1:] >= 2 * a[:, N + 1:] * (1 - 1e-12)).all()
    # smooth across the quadrature node mu' = mu: the upward field has no spike on the grid
    up = fo(0.3)[L // 2, N + 2:]
    assert np.abs(np.diff(up, 2)).max() < 0.05 * up.max()
