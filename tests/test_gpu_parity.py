"""Parity of the HIP path (through the C ABI) with the reference's golden vectors and with the
CPU oracle.  Needs a real MI355X: run with `-m gpu`.

Tolerance: the north star's 1e-10 relative on radiance fields (util.RTOL), element-wise where the
field carries signal and in units of the field maximum elsewhere.
"""
import numpy as np
import pytest

import sos_oracle as O
from sosrt import I1_In, In_limit, _lib, inputs
from sosrt.main import SOS_Aer, SOS_Aer_batch
from sosrt.solver import Solver
from util import RTOL, assert_close, column_case, g1_case, golden, oracle_column, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["fast", "ring", "general", "scan"])
def transport_mode(request, monkeypatch):
    """The transport kernels: the wave-independent one (+ repair), the same fed through an LDS ring by
    loader waves, and the general LDS-exchange one."""
    monkeypatch.setenv("SOSRT_TRANSPORT", request.param)
    for s in list(I1_In._handles.values()):
        s.close()
    I1_In._handles.clear()
    yield request.param
    for s in list(I1_In._handles.values()):
        s.close()
    I1_In._handles.clear()


# ----------------------------------------------------------------------------------------------
# MFMA contraction: operand / result lane maps, with asymmetric data
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L,N,B", [(50, 32, 1), (37, 100, 3), (200, 128, 2)])
def test_source_function_is_the_dense_product(L, N, B):
    rng = np.random.default_rng(7 * N + L)
    mu = inputs.direction_grid(N)
    P = rng.uniform(0.2, 3.0, (2 * N, 2 * N))            # not symmetric: a transposed tile would show
    X = rng.uniform(0.0, 1.0, (B, L, 2 * N)) * np.linspace(0.5, 2.0, 2 * N)
    alb = rng.uniform(0.5, 1.0, B)
    s = Solver(L, N, max_batch=B)
    s.set_grid(mu)
    s.set_phase(P)
    asym, uses = s.phase_asymmetry()
    assert asym > 0.1 and not uses                       # no flip symmetry: the full 2N x 2N product runs
    s.set_columns_single_slab(np.full(B, 0.5), alb, np.full(B, 0.3))
    J = s.source(X)
    for b in range(B):
        ref = O.Jn_NumInt(2, X[b], np.zeros(L), mu, 0.3, 0.5, P, alb[b], N)
        assert_close(J[b], ref, 1e-13, "Jn column %d" % b)
    s.close()


@pytest.mark.parametrize("L,N,B", [(50, 32, 1), (37, 100, 3), (200, 128, 2), (24, 256, 1), (21, 70, 2), (9, 6, 1)])
def test_source_function_symmetric_form(L, N, B):
    """A flip-symmetric matrix (P[m][k] = P[2N-1-m][2N-1-k], as every phase function of the scattering angle gives) that
    is not symmetric under transposition: the library runs the two N x N products (jn_gemm.hip, SYM) and the result is
    the reference's trapezoid sum; the forced full product gives the same to rounding."""
    rng = np.random.default_rng(11 * N + L)
    mu = inputs.direction_grid(N)
    G = rng.uniform(0.2, 3.0, (2 * N, 2 * N))
    P = G + G[::-1, ::-1]
    assert np.abs(P - P.T).max() > 0.1
    X = rng.uniform(0.0, 1.0, (B, L, 2 * N)) * np.linspace(0.5, 2.0, 2 * N)
    alb = rng.uniform(0.5, 1.0, B)
    s = Solver(L, N, max_batch=B)
    s.set_grid(mu)
    s.set_phase(P)
    asym, uses = s.phase_asymmetry()
    assert asym <= 1e-13 and uses           # (the rounding of the trapezoid weights of the grid)
    s.set_columns_single_slab(np.full(B, 0.5), alb, np.full(B, 0.3))
    J = s.source(X)
    s.set_contraction("f64_full")
    assert not s.phase_asymmetry()[1]
    Jf = s.source(X)
    s.set_contraction("f64")
    assert np.array_equal(s.source(X), J)                # (switching back and forth leaves the folded matrices intact)
    for b in range(B):
        ref = O.Jn_NumInt(2, X[b], np.zeros(L), mu, 0.3, 0.5, P, alb[b], N)
        assert_close(J[b], ref, 1e-13, "Jn column %d (symmetric form)" % b)
        assert_close(Jf[b], ref, 1e-13, "Jn column %d (full product)" % b)
    s.close()


# ----------------------------------------------------------------------------------------------
# step level against the reference's own outputs (G1): every a4a / a4b branch
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden("g1_*.npz"), ids=lambda p: p.split("/")[-1][3:-4])
def test_step_functions_match_reference(path, transport_mode):
    d, N, P = g1_case(path)
    tau, mu, tS, mu0, alb = d["tau"], d["mu"], float(d["tauStar"]), float(d["mu0"]), float(d["alb"])
    assert_close(I1_In.I1_NumInt(tau, mu, tS, mu0, d["P0"], alb, N), d["I1"], RTOL, "I1_NumInt")
    In_1, n = d["I1"], 2
    while "In_%d" % n in d:
        Jn = I1_In.Jn_NumInt(n, In_1, tau, mu, tS, mu0, P, alb, N)
        assert_close(Jn, d["Jn_%d" % n], RTOL, "Jn_NumInt n=%d" % n)
        In = I1_In.In_NumInt(n, d["Jn_%d" % n], In_1, tau, mu, tS, mu0, P, alb, N, 0, 0)
        assert_close(In, d["In_%d" % n], RTOL, "In_NumInt n=%d" % n)
        In_1 = d["In_%d" % n]
        n += 1


def test_helpers_match_reference():
    d = np.load(golden("g2_helpers.npz")[0])
    for i in range(int(d["n_asym"])):
        tt, mu, r = d["a%d_par" % i]
        v = In_limit.improved_asymptotic_downward_radiance(d["a%d_J" % i], d["a%d_tau" % i], tt, mu)
        assert v == pytest.approx(r, rel=1e-12, abs=0), i
    assert In_limit.improved_asymptotic_downward_radiance(np.zeros(0), np.zeros(0), 0.1, -2e-3) == 0.0
    for i in range(int(d["n_lim"])):
        N, idx = (int(x) for x in d["l%d_Nidx" % i])
        row, mud = d["l%d_row" % i], np.linspace(-1, 0, N)
        v = np.array([In_limit.improved_limit_mu_down(row, mud, N, idx, k) for k in range(idx)])
        # 1e-10 of the radiance the values are extrapolated from: these rows carry 10 % noise, which an extrapolation over
        # idx grid steps amplifies (|coefficients| up to 150 at N = 501) -- for the reference's np.polyfit as for the table
        src = np.max(np.abs(row[-(idx + 5):-idx]))
        assert np.max(np.abs(v - d["l%d_vals" % i])) <= RTOL * src, "improved_limit_mu_down N=%d idx=%d" % (N, idx)
        if N <= 256:
            assert_close(v, d["l%d_vals" % i], RTOL, "improved_limit_mu_down N=%d idx=%d" % (N, idx))
        w = np.array([In_limit.limit_mu_down(row, mud, N, idx, k) for k in range(idx)])
        assert np.array_equal(w, d["l%d_lin" % i])
    for N, m1, m2 in d["mu_approx"]:
        assert In_limit.mu_approx_In(inputs.direction_grid(int(N)), int(N)) == (int(m1), int(m2))


def test_search_beyond_wave0_is_repaired(transport_mode):
    """Second differences above 1e-4 over the first ~70 upward angles: the search leaves wave 0 and the
    fast kernel hands the column to the general one (cv.redo); same numbers as the oracle either way."""
    N, L = 128, 24
    mu = inputs.direction_grid(N)
    tau = np.linspace(0, 0.4, L) ** 1.3
    j = np.arange(N)
    Jn = np.zeros((L, 2 * N))
    Jn[:, :N] = 0.03 + 0.01 * np.linspace(0, 1, N)
    Jn[:, N:] = 0.1 * (1.0 + 0.3 * mu[N:])[None, :] * (1 + 0.2 * np.linspace(0, 1, L))[:, None]
    Jn[:, N:] += np.where(j < 70, 0.02 * (-1.0) ** j, 0.0)[None, :]
    ref = O.In_NumInt(2, Jn, None, tau, mu, 0.4, 0.5, None, 1.0, N, literal=False)
    got = I1_In.In_NumInt(2, Jn, None, tau, mu, 0.4, 0.5, np.ones((2 * N, 2 * N)), 1.0, N, 0, 0)
    assert_close(got, ref, RTOL, "In with a long blend")
    # the blend really reaches beyond lane 63: the row is exactly linear in mu up to lane ~70
    d2 = np.abs(np.diff(ref[L // 2, N:], 2))
    assert int(np.argmax(d2 > 1e-12)) + 1 > 64


def test_split_scan_redo_reads_the_other_halfs_rows_fresh(monkeypatch):
    """Whole columns whose upward mu -> 0+ search leaves the first 64 lanes in most orders (the phase matrices oscillate over
    the first 70 upward directions), through the chunk-parallel kernel with TWO workgroups per column: the workgroup that
    arrives second redoes the sweep row by row and corrects the running total with rows the OTHER workgroup -- possibly on
    another XCD -- stored (part 1's rows were plain stores, dirty in its L2 when part 0 arrived last).  Same bits
    as the ring kernel (one workgroup per column), on every repeat, and the oracle's field."""
    N, L = 128, 40
    mu = inputs.direction_grid(N)
    j = np.arange(N)
    c = np.ones(2 * N)
    c[N:] = np.where(j < 70, 1 + 0.3 * (-1.0) ** j, 1.0)
    cols = [(0.6, 0.3, 0.3), (0.6, 0.9, 0.6), (0.6, 0.05, 0.1), (0.35, 0.5, 0.45)]
    P_atm = O.phase_rayleigh(N, mu, 0.5)[1] * c[:, None]
    P_aer = O.phase_hg(N, mu, 0.5, 0.7)[1] * c[:, None]
    m0, ta, rh = (np.array(x) for x in zip(*cols))
    P0a = np.stack([O.phase_rayleigh(N, mu, m)[0] for m in m0])
    P0r = np.stack([O.phase_hg(N, mu, m, 0.7)[0] for m in m0])
    iu, idn = inputs.slab_indices(120, 40, 12, L)
    tau = np.stack([inputs.tau_profile(0.124, t, 120, 40, 12, L) for t in ta])
    out = {}
    for mode, reps in (("ring", 1), ("scan", 6)):
        monkeypatch.setenv("SOSRT_TRANSPORT", mode)        # read when the handle is created
        s = Solver(L, N, max_batch=len(cols), max_orders=200)
        try:
            s.set_grid(mu); s.set_phase(P_atm, P_aer)
            assert not s.phase_asymmetry()[1]              # not flip-symmetric: the full product runs
            s.set_columns(np.full(len(cols), iu), np.full(len(cols), idn), m0, rh, 1.0, 0.95, 0.124 / L, ta / (idn + 1 - iu), 0.124 + ta)
            out[mode] = [s.solve(tau, P0a, P0r) for _ in range(reps)]
        finally:
            s.close()
    ring = out["ring"][0]
    assert np.all(ring.status == _lib.COL_OK)
    for r in out["scan"]:
        assert np.array_equal(r.n, ring.n) and np.array_equal(r.status, ring.status)
        assert np.array_equal(r.I, ring.I)                 # bit for bit, every repeat
    long_blends = 0
    for b, (a, t, g) in enumerate(cols):
        col = O.make_column(a, 120, 40, 12, L, 0.124, t, g, 1.0, 0.95, N, P0a[b], P_atm, P0r[b], P_aer)
        ref = O.solve_column(col, literal=False)
        assert ring.n[b] == ref.n
        assert_close(ring.I[b], ref.I, RTOL, "column %d" % b)
        # the blend really went beyond lane 63 in several orders of this column: rows exactly linear in mu up to lane ~70
        long_blends += sum(int(np.argmax(np.abs(np.diff(In[L // 2, N:], 2)) > 1e-12)) + 1 > 64 for In in ref.I_saved)
    assert long_blends >= 12


def test_index_error_like_the_reference():
    N, L = 16, 6
    mu = inputs.direction_grid(N)
    tau = np.linspace(0, 0.3, L)
    Jn = np.zeros((L, 2 * N))
    Jn[:, N:] = 50.0 * (-1.0) ** np.arange(N)
    with pytest.raises(IndexError):
        O.In_NumInt(2, Jn, None, tau, mu, 0.3, 0.5, None, 1.0, N, literal=False)
    with pytest.raises(IndexError):
        I1_In.In_NumInt(2, Jn, None, tau, mu, 0.3, 0.5, np.ones((2 * N, 2 * N)), 1.0, N, 0, 0)


# ----------------------------------------------------------------------------------------------
# column level against the reference (G3 specular, G6 Lambertian n >= 2, G4 digest at C2 shape)
# ----------------------------------------------------------------------------------------------
def _solve_fixture(c, I1=None, max_orders=64, contraction=None):
    s = Solver(c["L"], c["N"], max_batch=1, max_orders=max_orders)
    s.set_grid(c["mu"])
    s.set_phase(c["P_atm"], c["P_aer"])
    if contraction:
        s.set_contraction(contraction)
    else:                # the reference's phase matrices have the flip symmetry to rounding: the symmetric form is what runs
        assert s.phase_asymmetry()[1]
    s.set_columns([c["idx_up"]], [c["idx_down"]], c["mu0"], c["grd_alb"], c["alb_atm"], c["alb_aer"], c["dtau_atm"],
                  c["dtau_aer"], c["tauStar_atm"] + c["tauStar_aer"], surface=c["surface"])
    r = s.solve(c["tau"][None], c["P0_atm"][None], c["P0_aer"][None], I1=None if I1 is None else I1[None], save_orders=True)
    fd, fu = s.fluxes(c["tau"][None], r.I)
    s.close()
    return r, fd[0], fu[0]


@pytest.mark.parametrize("path", golden("g3_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_specular_column_matches_reference(path, transport_mode):
    d, c = column_case(path)
    r, fd, fu = _solve_fixture(c)
    assert r.status[0] == _lib.COL_OK and r.n[0] == c["n"]
    assert_close(r.I[0], d["I"], RTOL, "I")
    for k in range(c["n"]):
        assert_close(r.I_saved[0, k], d["I_saved"][k], RTOL, "order %d" % (k + 1))
    assert not r.I_saved[0, c["n"]:].any()
    ofd, ofu = O.fluxes(d["I"], c["mu"], c["tau"], c["N"], c["mu0"], c["grd_alb"])
    assert_close(fd, ofd, 1e-12, "flux down")
    assert_close(fu, ofu, 1e-12, "flux up")


@pytest.mark.parametrize("path", golden("g3_*.npz") + golden("g4_spec_C2_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_full_product_contraction_matches_reference_and_the_symmetric_form(path):
    """SOSRT_CONTRACT_F64_FULL (the 2N x 2N product, what runs for matrices without the flip symmetry) against the same
    goldens, and against the default symmetric form: same order counts, fields 1e-12 apart at most."""
    d, c = column_case(path)
    rf, _, _ = _solve_fixture(c, contraction="f64_full")
    rs, _, _ = _solve_fixture(c)
    assert rf.n[0] == c["n"] == rs.n[0]
    if "I" in d:
        assert_close(rf.I[0], d["I"], RTOL, "I (full product)")
        for k in range(c["n"]):
            assert_close(rf.I_saved[0, k], d["I_saved"][k], RTOL, "order %d (full product)" % (k + 1))
    else:
        N, L = c["N"], c["L"]
        assert_close(rf.I[0, 0, N:], d["toa_up"], RTOL, "TOA up (full product)")
        assert_close(rf.I[0].sum(axis=0), d["col_sum"], RTOL, "column sums (full product)")
    assert_close(rs.I[0], rf.I[0], 1e-12, "symmetric form vs full product")


@pytest.mark.parametrize("path", golden("g6_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_lambertian_orders_match_modified_reference(path, transport_mode):
    """n >= 2 with the coded sign of lam:399/401, seeded with the fixture's first order (H1, H2)."""
    d, c = column_case(path)
    r, _, _ = _solve_fixture(c, I1=d["I_saved"][0])
    assert r.n[0] == c["n"]
    for k in range(c["n"]):
        assert_close(r.I_saved[0, k], d["I_saved"][k], RTOL, "order %d" % (k + 1))
    assert_close(r.I[0], d["I"], RTOL, "I")
    assert d["I_saved"][1][-1, c["N"] + 8] < 0        # the reflected upward term is negative, as coded


def test_c2_column_matches_reference_digest():
    d, c = column_case(golden("g4_spec_C2_*.npz")[0])
    r, _, _ = _solve_fixture(c)
    N, L = c["N"], c["L"]
    assert r.n[0] == c["n"]
    assert_close(r.I[0, 0, N:], d["toa_up"], RTOL, "TOA up")
    assert_close(r.I[0, L - 1, :N], d["sfc_down"], RTOL, "surface down")
    assert_close(r.I[0, L // 2], d["I_mid_row"], RTOL, "middle row")
    assert_close(r.I[0].sum(axis=0), d["col_sum"], RTOL, "column sums")
    assert_close(r.I_saved[0, :c["n"]].reshape(c["n"], -1).sum(axis=1), d["order_sum"], RTOL, "order sums")
    assert r.I[0, 0, N:].max() == pytest.approx(0.6983369737935123, rel=1e-11)


def test_SOS_Aer_call_surface():
    r = SOS_Aer(mu0=0.5, nb_layers=50, nb_angles=32, grd_alb=0.15, atm_phase_fun="iso", aer_phase_fun="iso")
    d = np.load(golden("g3_spec_C1_iso.npz")[0])
    assert (r.n, r.idx_up, r.idx_down) == (9, 39, 42)
    assert np.array_equal(r.tau, d["tau"]) and np.array_equal(r.mu, d["mu"])
    assert_close(r.I, d["I"], RTOL, "I")
    assert r.I_saved.shape == (9, 50, 64)
    assert r.I[0, 32:].max() == pytest.approx(0.6862303878773028, rel=1e-12)
    with pytest.raises(TypeError):
        SOS_Aer(nb_layer=3)
    full = SOS_Aer(nb_layers=40, nb_angles=32, grd_alb=0.15)     # the shipped EVA aerosol (own Mie series: parity unpinned)
    assert full.status == 0 and full.n >= 2 and np.isfinite(full.I).all() and (full.I[0, 33:] > 0).all()


# ----------------------------------------------------------------------------------------------
# seeded batches against the oracle (ragged convergence, every surface, every idx bucket)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L,N,surface", [(40, 32, "specular"), (36, 64, "lambertian"), (30, 100, "specular"),
                                         (36, 64, "lambertian_readme")])
def test_seeded_batch_matches_oracle(L, N, surface, transport_mode):
    rng = np.random.default_rng(1000 + N)
    B = 6
    mu0 = rng.uniform(0.2, 1.0, B)
    taer = np.array([0.01, 0.1, 0.5, 1.0, 2.0, 0.05])
    rho = np.array([0.0, 0.15, 0.4, 0.8, 0.3, 0.6])
    walb = np.array([1.0, 0.97, 0.9, 0.95, 0.85, 1.0])
    mu = inputs.direction_grid(N)
    P_atm = inputs.phase_function("rayleigh", N, mu, 0.5)[1]
    P_aer = inputs.phase_function("hg", N, mu, 0.5, 0.7)[1]
    r = SOS_Aer_batch(mu0, taer, rho, tauStar_atm=0.124, alb_aer=walb, nb_layers=L, nb_angles=N, z_up=40, z_down=12,
                      P_atm=P_atm, P_aer=P_aer, surface=surface, save_orders=False, max_orders=200, raise_on_error=False)
    for b in range(B):
        P0a = inputs.phase_function("rayleigh", N, mu, mu0[b])[0]
        P0r = inputs.phase_function("hg", N, mu, mu0[b], 0.7)[0]
        col = O.make_column(mu0[b], 120, 40, 12, L, 0.124, taer[b], rho[b], 1.0, walb[b], N, P0a, P_atm, P0r, P_aer,
                            surface=surface)
        try:
            ref = O.solve_column(col, literal=False)
        except IndexError:
            # the reference's unbounded upward search ran off the grid for this column: same outcome required
            assert r.status[b] == _lib.COL_INDEXERROR, b
            continue
        assert r.status[b] == _lib.COL_OK, b
        assert r.n[b] == ref.n, (b, r.n[b], ref.n)
        # 'lambertian' as the reference codes it reflects NEGATIVE radiance: successive orders alternate in
        # sign and their sum cancels, which amplifies the rounding of either implementation relative to |I|
        assert_close(r.I[b], ref.I, 1e-9 if surface == "lambertian" else RTOL, "column %d" % b)


# ----------------------------------------------------------------------------------------------
# BASELINE sizes (C4 shape: L=200, N=128, a sweep of columns): properties + sampled oracle check
# ----------------------------------------------------------------------------------------------
def test_c4_sweep_properties_and_sampled_oracle():
    L, N = 200, 128
    mu0 = np.linspace(0.2, 1.0, 4)
    taer = np.geomspace(0.01, 1.0, 4)
    rho = np.linspace(0.0, 0.8, 4)
    M0, TA, RH = (x.ravel() for x in np.meshgrid(mu0, taer, rho, indexing="ij"))
    mu = inputs.direction_grid(N)
    P_atm = inputs.phase_function("rayleigh", N, mu, 0.5)[1]
    P_aer = inputs.phase_function("hg", N, mu, 0.5, 0.7)[1]
    kw = dict(tauStar_atm=0.124, alb_aer=0.97, nb_layers=L, nb_angles=N, P_atm=P_atm, P_aer=P_aer, max_orders=128)
    r = SOS_Aer_batch(M0, TA, RH, save_orders=True, **kw)
    B = len(M0)
    assert (r.status == 0).all() and (r.n >= 2).all()
    # the field is the sum of its orders
    for b in range(0, B, 7):
        assert_close(r.I_saved[b, :r.n[b]].sum(axis=0), r.I[b], 1e-13, "sum of orders")
    # batch invariance: the same columns in reverse order give bit-identical fields
    r2 = SOS_Aer_batch(M0[::-1], TA[::-1], RH[::-1], **kw)
    assert np.array_equal(r2.I[::-1], r.I) and np.array_equal(r2.n[::-1], r.n)
    # more aerosol or a brighter surface never needs fewer orders
    n3 = r.n.reshape(4, 4, 4)
    assert (np.diff(n3, axis=2) >= 0).all() and (np.diff(n3, axis=1) >= 0).all()
    # sampled columns against the (vectorised) oracle
    for b in (0, 21, 42, 63):
        P0a = inputs.phase_function("rayleigh", N, mu, M0[b])[0]
        P0r = inputs.phase_function("hg", N, mu, M0[b], 0.7)[0]
        col = O.make_column(M0[b], 120, 25, 17, L, 0.124, TA[b], RH[b], 1.0, 0.97, N, P0a, P_atm, P0r, P_aer)
        ref = O.solve_column(col, literal=False)
        assert r.n[b] == ref.n
        assert_close(r.I[b], ref.I, RTOL, "column %d" % b)


def test_argument_errors_on_device():
    s = Solver(20, 16, max_batch=2)
    s.set_grid(inputs.direction_grid(16))
    with pytest.raises(ValueError):
        s.set_columns([0], [5], 0.5, 0.1, 1.0, 1.0, 0.01, 0.01, 0.2)   # idx_up must be >= 1
    with pytest.raises(ValueError):
        s.set_columns([3], [19], 0.5, 0.1, 1.0, 1.0, 0.01, 0.01, 0.2)  # idx_down must be <= L-2
    s.set_columns_single_slab(0.5, 1.0, 0.2)
    with pytest.raises(_lib.SosrtError):
        s.source(np.zeros((1, 20, 32)))                                # phase matrix not set
    with pytest.raises(ValueError):
        s.first_order(np.zeros((1, 19)), np.zeros((1, 32)))            # wrong tau length
    s.close()


def test_mixed_slab_geometries_and_shared_profiles_in_one_batch(transport_mode):
    """Columns of one batch with different aerosol slabs (per-column idx_up / idx_down: the live-column
    tiling of the source function maps plain and slab rows per column) and with repeated optical-depth
    profiles (columns that share an attenuation table) -- each against its own oracle column."""
    L, N = 40, 32
    mu = inputs.direction_grid(N)
    P0_r, P_atm = inputs.phase_function("rayleigh", N, mu, 0.5)
    P_aer = inputs.phase_function("hg", N, mu, 0.5, 0.7)[1]
    #        mu0   z_up z_down  taer  rho
    cols = [(0.50, 40, 12, 0.30, 0.15),
            (0.80, 40, 12, 0.30, 0.40),     # same profile as column 0
            (0.35, 60, 30, 0.30, 0.15),
            (0.60, 25, 17, 0.80, 0.60),
            (0.60, 25, 17, 0.05, 0.00),
            (0.95, 90, 6, 1.50, 0.30),
            (0.50, 40, 12, 0.30, 0.70),     # same profile as column 0
            (0.25, 25, 17, 0.80, 0.90)]     # same profile as column 3
    B = len(cols)
    t_atm = 0.124
    iu = np.empty(B, dtype=np.int32); idn = np.empty(B, dtype=np.int32)
    tau = np.empty((B, L)); P0a = np.empty((B, 2 * N)); P0r = np.empty((B, 2 * N))
    for b, (m0, zu, zd, ta, rho) in enumerate(cols):
        iu[b], idn[b] = inputs.slab_indices(120, zu, zd, L)
        tau[b] = inputs.tau_profile(t_atm, ta, 120, zu, zd, L)
        P0a[b] = inputs.phase_function("rayleigh", N, mu, m0)[0]
        P0r[b] = inputs.phase_function("hg", N, mu, m0, 0.7)[0]
    assert len({(int(a), int(c)) for a, c in zip(iu, idn)}) >= 4
    m0 = np.array([c[0] for c in cols]); ta = np.array([c[3] for c in cols]); rho = np.array([c[4] for c in cols])
    s = Solver(L, N, max_batch=B, max_orders=200)
    s.set_grid(mu); s.set_phase(P_atm, P_aer)
    s.set_columns(iu, idn, m0, rho, 1.0, 0.95, t_atm / L, ta / (idn + 1 - iu), t_atm + ta, surface="specular")
    r = s.solve(tau, P0a, P0r, tol=1e-4)
    s.close()
    for b, (mu0, zu, zd, taer, grd) in enumerate(cols):
        col = O.make_column(mu0, 120, zu, zd, L, t_atm, taer, grd, 1.0, 0.95, N, P0a[b], P_atm, P0r[b], P_aer)
        try:
            ref = O.solve_column(col, literal=False)
        except IndexError:
            assert r.status[b] == _lib.COL_INDEXERROR, b
            continue
        assert r.status[b] == _lib.COL_OK, b
        assert r.n[b] == ref.n, (b, r.n[b], ref.n)
        assert_close(r.I[b], ref.I, RTOL, "column %d" % b)



def test_first_order_just_outside_the_limit_window():
    """spec:113-292 divides a difference of exponentials by (mu0 +- mu) and takes the limit form only where |mu0 - |mu|| < 1e-4.
    A sun 1.6e-4 from a node of the direction grid is the worst case left: the rounding of the reference's two exponentials is
    amplified 5000 times there (a few 1e-13 absolute; 1e-10 and more relative to the element), so the reference's own float64
    result is that far from the exact value of its formula, and no float64 kernel that does not replay its roundings agrees
    with it more closely.  What can be asked: the kernel is as close to the formula evaluated in long double
    (oracle.first_order_extended) as the reference's arithmetic is -- and at the usual bar against the reference everywhere else
    (found by tools/fuzz_parity.py, seed 101 case 24: 1.5e-10 against the oracle in every transport mode)."""
    N, L = 128, 73
    mu = inputs.direction_grid(N)
    for gap, plain in ((1.6e-4, False), (3e-2, True)):
        mu0 = float(-mu[26]) - gap
        P0a, Pa = inputs.phase_function("rayleigh", N, mu, mu0)
        P0r, Pr = inputs.phase_function("hg", N, mu, mu0, 0.7)
        col = O.make_column(mu0, 120, 70.8, 24.0, L, 0.124, 1.0, 0.39, 1.0, 0.95, N, P0a, Pa, P0r, Pr)
        s = Solver(L, N, max_batch=1, max_orders=4)
        s.set_grid(mu); s.set_phase(Pa, Pr)
        s.set_columns([col.idx_up], [col.idx_down], mu0, 0.39, 1.0, 0.95, col.dtau_atm, col.dtau_aer, col.tauStar_tot)
        I1 = s.first_order(col.tau[None], P0a[None], P0r[None])[0]
        s.close()
        ref, ext = O.first_order(col), O.first_order_extended(col)
        noise = rel_err(ref, ext)                      # the reference's arithmetic against its own formula
        mine = rel_err(I1, ext)
        if plain:
            assert rel_err(I1, ref) <= RTOL
        else:
            assert noise > 5e-11                       # the case is what it claims to be (8e-11 here; 2e-10 in the fuzz case)
            assert mine <= max(2 * noise, RTOL), (mine, noise)
            assert np.max(np.abs(I1 - ext)) <= 1e-11 * np.max(np.abs(ext))      # absolute: a few 1e-13 on a scale of 0.4


@pytest.mark.parametrize("L,N,zs,taer", [(40, 256, (26, 24), 4.0), (40, 256, (60, 24), 60.0), (24, 501, (70, 50), 8.0)])
def test_first_order_of_an_optically_thick_layer_at_a_zone_edge_stays_finite(L, N, zs, taer):
    """k_first_order evaluates one exponential per element and gets the row's other attenuations as that one times a constant of
    the zone and the direction, e^{+dtau/|mu|} of the layer at the zone's edge -- which overflows where that layer is optically
    thick for the direction (dtau/|mu| > 709: a one-layer slab of tau 4 at N = 256 has 1020 at mu = 1/255) while the exponential
    underflows: the product was NaN where the direct value is at most 1.  Such directions take the direct
    exponential; the field stays finite and at the usual bar against the oracle, and so does the whole column."""
    mu = inputs.direction_grid(N)
    mu0 = 0.6
    P0a, Pa = inputs.phase_function("rayleigh", N, mu, mu0)
    P0r, Pr = inputs.phase_function("hg", N, mu, mu0, 0.7)
    col = O.make_column(mu0, 120, zs[0], zs[1], L, 0.124, taer, 0.3, 1.0, 0.9, N, P0a, Pa, P0r, Pr)
    edge = (col.tau[col.idx_up] - col.tau[col.idx_up - 1]) / mu[N + 1]
    assert edge > 745, edge                                   # e^{edge} is inf, e^{-edge} is 0
    s = Solver(L, N, max_batch=1, max_orders=4)
    s.set_grid(mu); s.set_phase(Pa, Pr)
    s.set_columns([col.idx_up], [col.idx_down], mu0, 0.3, 1.0, 0.9, col.dtau_atm, col.dtau_aer, col.tauStar_tot)
    I1 = s.first_order(col.tau[None], P0a[None], P0r[None])[0]
    s.close()
    assert np.isfinite(I1).all()
    assert_close(I1, O.first_order(col), RTOL, "first order, thick layer at the zone edge")


# ----------------------------------------------------------------------------------------------
# the README's Lambertian first order (non-default option; PARITY UNPINNED -- SURVEY H1: the reference has no runnable
# code for it).  Device against the oracle's restatement of README.md:126-171, and two properties.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L,N,rho,taer", [(50, 32, 0.15, 0.3), (200, 128, 0.3, 0.12), (64, 48, 0.05, 0.6)])
def test_readme_lambertian_first_order(L, N, rho, taer):
    mu = inputs.direction_grid(N)
    mu0 = 0.6
    P0a, Pa = inputs.phase_function("rayleigh", N, mu, mu0)
    P0r, Pr = inputs.phase_function("hg", N, mu, mu0, 0.7)
    col = O.make_column(mu0, 120, 25, 17, L, 0.124, taer, rho, 1.0, 0.9, N, P0a, Pa, P0r, Pr, surface="lambertian_readme")
    col.first_order = "readme"
    ref1 = O.first_order_lambertian_readme(col)
    kw = dict(mu0=mu0, tauStar_aer=taer, grd_alb=rho, tauStar_atm=0.124, alb_atm=1.0, alb_aer=0.9, z0=120, z_up=25, z_down=17,
              nb_layers=L, nb_angles=N)
    r = SOS_Aer("lambertian_readme", P_atm=Pa, P0_atm=P0a, P_aer=Pr, P0_aer=P0r, first_order="readme", **kw)
    assert_close(r.I_saved[0], ref1, RTOL, "README Lambertian first order")
    ref = O.solve_column(col, literal=False)
    assert r.n == ref.n
    assert_close(r.I, ref.I, RTOL, "column on top of the README first order")
    # more light comes back from a Lambertian ground than the coded first order (specular beam) accounts for at the top
    coded = SOS_Aer("lambertian_readme", P_atm=Pa, P0_atm=P0a, P_aer=Pr, P0_aer=P0r, **kw)
    assert (r.I_saved[0] >= 0).all() and np.isfinite(r.I).all()
    assert r.I_saved[0][0, N + 1:].sum() != coded.I_saved[0][0, N + 1:].sum()
    # a black ground: both first orders are the direct-beam term alone
    kw0 = dict(kw, grd_alb=0.0)
    a = SOS_Aer("lambertian_readme", P_atm=Pa, P0_atm=P0a, P_aer=Pr, P0_aer=P0r, first_order="readme", **kw0)
    b = SOS_Aer("lambertian_readme", P_atm=Pa, P0_atm=P0a, P_aer=Pr, P0_aer=P0r, **kw0)
    # (two kernels, two ways to the same closed forms: the coded one evaluates one exponential per element and derives the other
    # attenuations from it, the README one evaluates each; where e^{-tau/mu0} - e^{-tau_b/mu0} e^{(tau - tau_b)/mu} cancels, a few ulp
    # of the exponentials show as 6e-13 of the difference -- both are checked against the reference / the oracle at 1e-10)
    assert_close(a.I_saved[0], b.I_saved[0], 1e-11, "rho = 0: README and coded first orders agree")
    assert a.n == b.n
