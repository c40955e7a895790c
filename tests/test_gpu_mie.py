"""The aerosol of the scenarios BASELINE names -- EVA (volcanic sulphate) and wildfire: log-normal Mie ensembles,
SOS_Aer_phase_func.py:398-753, README.md:95-111 -- through the device path: the ensemble's phase function goes to the
GPU as a table on the scattering cosine (sosrt/mie.py builds it; `miepython` is absent, so the table itself is PARITY
UNPINNED), the azimuth averages P0(mu, mu0) and P(mu, mu') are evaluated by the HIP kernels, and the solve runs on
those matrices.  What IS pinned here: device builders against the host / oracle evaluation of the same table at 1e-12,
and the HIP solve against the oracle on identical (P0, P) inputs at 1e-10 with equal order counts, at the BASELINE
shapes C2 (L=200, N=128), C3 (N=256) and a 64-column subset of C5 (L=400, N=256, wildfire slab 15/14 km).
Needs an MI355X."""
import numpy as np
import pytest

import sos_oracle as O
from sosrt import _lib, inputs
from sosrt.main import SOS_Aer, SOS_Aer_batch, get_solver
from sosrt.solver import Solver
from util import RTOL, assert_close, rel_err

pytestmark = pytest.mark.gpu

_trapz = getattr(np, "trapezoid", None) or np.trapz

# README.md:95-111
EVA = dict(z=(25, 17), taer=0.120, alb_aer=0.97, rho=0.15, mu0=0.5, tatm=0.124)
WILDFIRE = dict(z=(15, 14), taer=0.0075, alb_aer=0.97, rho=0.15, mu0=0.5, tatm=0.124)


def _oracle_inputs(name, N, mu0s):
    """(P_atm, P_aer, {mu0: (P0_atm, P0_aer)}) evaluated by the oracle's NumPy builders from the scenario's table."""
    mu = O.make_mu(N)
    mt, pt = inputs.scenario_table(name)
    P_atm = O.phase_rayleigh(N, mu, 0.5)[1]
    P_aer = O.phase_table(N, mu, 0.5, mt, pt)[1]
    p0 = {float(m): (O.phase_rayleigh(N, mu, float(m))[0], O.phase_table(N, mu, float(m), mt, pt)[0]) for m in mu0s}
    return P_atm, P_aer, p0


@pytest.mark.parametrize("name", ["eva", "wildfire", "mie"])
@pytest.mark.parametrize("N", [32, 128])
def test_device_builders_match_host_on_mie_tables(name, N):
    """k_phase_p0 / k_phase_matrix on the 6001-point table of a Mie ensemble (or of one sphere) against the host builder
    and the oracle's: 1e-12, plus the normalisations the reference imposes (phase:103,131)."""
    mu = inputs.direction_grid(N)
    kw = dict(r=0.4, lambda0=0.55, indx=1.44 + 0.0j) if name == "mie" else {}
    mu0s = np.array([0.2, 0.5, 0.83, 1.0])
    P0d, Pd = inputs.phase_function_device(name, N, mu, mu0s, **kw)
    mt, pt = inputs._scalar_phase(name, **kw)[1][1]
    assert mt.shape == (6001,) and np.all(pt > 0)
    for i, m0 in enumerate(mu0s):
        P0h, Ph = inputs.phase_function(name, N, mu, float(m0), **kw)
        P0o, Po = O.phase_table(N, mu, float(m0), mt, pt)
        assert_close(P0d[i], P0h, 1e-12, "%s P0 mu0=%g device vs host" % (name, m0))
        assert_close(P0d[i], P0o, 1e-12, "%s P0 mu0=%g device vs oracle" % (name, m0))
        assert abs(_trapz(P0d[i], mu) - 2) < 1e-12
    assert_close(Pd, Ph, 1e-12, "%s P device vs host" % name)
    assert_close(Pd, Po, 1e-12, "%s P device vs oracle" % name)
    assert np.max(np.abs(_trapz(Pd, mu, axis=0) - 4)) < 1e-11


def _solve_vs_oracle(name, sc, L, N, cols, check, what):
    """`cols` = [(mu0, tauStar_aer, grd_alb)]; the HIP solve with DEVICE-built inputs, and -- for the columns `check` -- the
    HIP solve on the oracle's inputs against the oracle's solve of the same inputs."""
    m0, ta, rh = (np.array(x, dtype=np.float64) for x in zip(*cols))
    P_atm, P_aer, p0 = _oracle_inputs(name, N, np.unique(m0[check]))
    kw = dict(tauStar_atm=sc["tatm"], alb_aer=sc["alb_aer"], nb_layers=L, nb_angles=N, z_up=sc["z"][0], z_down=sc["z"][1],
              max_orders=200)
    # (a) the product path: names in, everything built on the device
    r = SOS_Aer_batch(m0, ta, rh, atm_phase_fun="rayleigh", aer_phase_fun=name, **kw)
    assert np.all(r.status == _lib.COL_OK)
    s = get_solver(L, N, len(cols), 200)
    asym, uses = s.phase_asymmetry()
    print("\n[%s] %s L=%d N=%d: matrix asymmetry %.2e, symmetric contraction %s, orders %s" % (
        what, name, L, N, asym, uses, sorted(set(r.n.tolist()))))
    # the folded matrices of a function of the scattering angle are flip-symmetric to the rounding of the builders
    assert asym <= 1e-12 and uses, "the full product must run for this matrix: asymmetry %.3e" % asym
    # (b) identical inputs: the oracle's matrices and P0 rows through the HIP path vs the oracle
    P0a = np.stack([p0[float(m)][0] for m in m0[check]])
    P0r = np.stack([p0[float(m)][1] for m in m0[check]])
    h = SOS_Aer_batch(m0[check], ta[check], rh[check], P_atm=P_atm, P_aer=P_aer, P0_atm=P0a, P0_aer=P0r, **kw)
    worst = 0.0
    for i, b in enumerate(check):
        col = O.make_column(m0[b], 120, sc["z"][0], sc["z"][1], L, sc["tatm"], ta[b], rh[b], 1.0, sc["alb_aer"], N,
                            P0a[i], P_atm, P0r[i], P_aer)
        ref = O.solve_column(col, literal=False)
        assert h.status[i] == _lib.COL_OK and h.n[i] == ref.n, (b, h.status[i], h.n[i], ref.n)
        worst = max(worst, assert_close(h.I[i], ref.I, RTOL, "%s column %d (identical inputs)" % (what, b)))
        # and the product path (device-built inputs, 1e-12 from the oracle's) lands on the same field
        assert r.n[b] == ref.n
        assert_close(r.I[b], ref.I, RTOL, "%s column %d (device-built inputs)" % (what, b))
    print("[%s] max rel err vs oracle on identical inputs: %.2e" % (what, worst))
    return r


def test_c2_eva_column():
    """BASELINE configs[1]: EVA scenario, N_tau = 200, N_mu = 128, one column."""
    _solve_vs_oracle("eva", EVA, 200, 128, [(EVA["mu0"], EVA["taer"], EVA["rho"])], [0], "C2")


def test_c3_eva_column_n256():
    """BASELINE configs[2]: EVA scenario, N_tau = 200, N_mu = 256, one column."""
    _solve_vs_oracle("eva", EVA, 200, 256, [(EVA["mu0"], EVA["taer"], EVA["rho"])], [0], "C3")


def test_c4_eva_sweep_subset():
    """27 columns of the C4 sweep axes (mu0 x tau*_aer x grd_alb) with the EVA aerosol instead of the HG stand-in."""
    cols = [(a, t, g) for a in (0.2, 0.6, 1.0) for t in (0.01, 0.12, 1.0) for g in (0.0, 0.15, 0.8)]
    r = _solve_vs_oracle("eva", EVA, 200, 128, cols, [0, 13, 26], "C4 subset")
    n = r.n.reshape(3, 3, 3)
    assert (np.diff(n, axis=1) >= 0).all() and (np.diff(n, axis=2) >= 0).all()      # more aerosol / brighter ground: more orders


def test_c5_wildfire_subset_l400_n256():
    """64 columns of BASELINE configs[4]: wildfire scenario (slab 15/14 km, m = 1.7 + 0.03j, sigma = 1.5, r_m = 0.065 um),
    N_tau = 400, N_mu = 256, specular surface; 4 x 4 x 4 around the scenario's values."""
    cols = [(a, t, g) for a in np.linspace(0.2, 1.0, 4) for t in WILDFIRE["taer"] * np.array([0.25, 1.0, 4.0, 16.0])
            for g in (0.0, 0.15, 0.45, 0.8)]
    b_scn = 16 * 1 + 4 * 1 + 1                     # nearest to the README's column among the 64
    _solve_vs_oracle("wildfire", WILDFIRE, 400, 256, cols, [0, b_scn, 63], "C5 subset")


def test_rows_whose_search_leaves_the_first_wave_are_finished_one_by_one(monkeypatch):
    """Three columns of the EVA headline sweep (mu0 = 0.2, bright ground) have, in the second order, a handful of rows whose
    upward mu -> 0+ search (spec:403-406) runs past lane 61 -- not the first row of a zone.  Round 2 redid the whole sweep row
    by row for them (620 us on the 512-column launch); now such rows are flagged and finished one by one after the sweep.
    Ring kernel, chunk-parallel kernel with one and with two workgroups per column: the same bits, and the oracle's field."""
    L, N = 200, 128
    tg, rg = np.geomspace(0.01, 1.0, 8), np.linspace(0.0, 0.8, 8)          # the axes of the headline sweep (bench.build_sweep)
    cols = [(0.2, tg[3], rg[7]), (0.2, tg[4], rg[6]), (0.2, tg[4], rg[7]), (0.5, 0.12, 0.15)]       # its columns 31, 38, 39
    m0, ta, rh = (np.array(x, dtype=np.float64) for x in zip(*cols))
    P_atm, P_aer, p0 = _oracle_inputs("eva", N, np.unique(m0))
    P0a = np.stack([p0[float(m)][0] for m in m0]); P0r = np.stack([p0[float(m)][1] for m in m0])
    kw = dict(tauStar_atm=0.124, alb_aer=0.97, nb_layers=L, nb_angles=N, z_up=25, z_down=17, max_orders=200,
              P_atm=P_atm, P_aer=P_aer, P0_atm=P0a, P0_aer=P0r)
    out = {}
    for tag, env in (("ring", dict(SOSRT_TRANSPORT="ring")), ("scan2", dict(SOSRT_TRANSPORT="scan")),
                     ("scan1", dict(SOSRT_TRANSPORT="scan", SOSRT_SCAN_SPLIT="0")), ("general", dict(SOSRT_TRANSPORT="general"))):
        for k in ("SOSRT_TRANSPORT", "SOSRT_SCAN_SPLIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)                     # (part of the solver cache's key / read when the handle is created)
        from sosrt import main as M
        for s_ in list(M._solvers.values()):
            s_.close()
        M._solvers.clear()
        out[tag] = SOS_Aer_batch(m0, ta, rh, **kw)
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()
    ring = out["ring"]
    for tag in ("scan1", "scan2"):
        assert np.array_equal(out[tag].n, ring.n) and np.array_equal(out[tag].I, ring.I), tag
    iu, idn = inputs.slab_indices(120, 25, 17, L)
    flagged = 0
    for b, (a, t, g) in enumerate(cols):
        col = O.make_column(a, 120, 25, 17, L, 0.124, t, g, 1.0, 0.97, N, P0a[b], P_atm, P0r[b], P_aer)
        ref = O.solve_column(col, literal=False)
        for tag in ("ring", "general"):
            assert out[tag].n[b] == ref.n, (tag, b)
            assert_close(out[tag].I[b], ref.I, RTOL, "%s column %d" % (tag, b))
        # rows of the second order that are exactly linear in mu beyond lane 61: the long searches; none of them opens a zone
        reach = np.argmax(np.abs(np.diff(ref.I_saved[1][:, N:], 2, axis=1)) > 1e-12, axis=1) + 1
        rows = np.nonzero(reach > 61)[0]
        assert iu not in rows and (idn + 1) not in rows      # (the first rows of the zones, going up)
        flagged += len(rows)
    assert flagged >= 6


@pytest.mark.parametrize("N,surface", [(128, "specular"), (128, "lambertian_readme"), (256, "specular"), (256, "lambertian_readme")])
def test_long_searches_with_saved_orders_every_surface_and_split(N, surface):
    """The same kind of column (EVA aerosol, mu0 = 0.2, bright ground: rows of the second order whose search passes lane 61)
    with the per-order history stored (I_saved, spec:304-305,458), over a specular and a Lambertian surface, at N = 128 and
    N = 256: two or four workgroups per column (specular), one (Lambertian, N = 128) or the ring kernel (Lambertian, N = 256).
    Every order against the oracle on identical inputs."""
    L = 200
    tg, rg = np.geomspace(0.01, 1.0, 8), np.linspace(0.0, 0.8, 8)
    mu0, taer, rho = 0.2, float(tg[4]), float(rg[7])
    P_atm, P_aer, p0 = _oracle_inputs("eva", N, [mu0])
    r = SOS_Aer(surface, mu0=mu0, nb_layers=L, nb_angles=N, grd_alb=rho, tauStar_atm=0.124, tauStar_aer=taer, alb_aer=0.97,
                P_atm=P_atm, P0_atm=p0[mu0][0], P_aer=P_aer, P0_aer=p0[mu0][1], max_orders=200)
    col = O.make_column(mu0, 120, 25, 17, L, 0.124, taer, rho, 1.0, 0.97, N, p0[mu0][0], P_atm, p0[mu0][1], P_aer, surface=surface)
    ref = O.solve_column(col, literal=False)
    assert r.n == ref.n and r.I_saved.shape[0] == ref.n
    assert_close(r.I, ref.I, RTOL, "I")
    for k in range(ref.n):
        assert_close(r.I_saved[k], ref.I_saved[k], RTOL, "order %d" % (k + 1))
    reach = np.argmax(np.abs(np.diff(ref.I_saved[1][:, N:], 2, axis=1)) > 1e-12, axis=1) + 1
    print("\n[N=%d %s] orders %d, longest search of the second order: lane %d" % (N, surface, ref.n, reach.max()))


def test_symmetric_and_full_contraction_agree_on_a_forward_peaked_matrix():
    """What the flip-symmetric form drops is the antisymmetric part of the folded matrices (sosrt.h, sosrt_set_contraction).
    The EVA matrix is the most forward-peaked input of the suite (p(1)/p(-1) = 117, max P = 59): the two forms of the
    product must still agree far inside the parity bar, on one application and on a whole column."""
    L, N = 200, 128
    P_atm, P_aer, p0 = _oracle_inputs("eva", N, [0.5])
    mu = O.make_mu(N)
    tau = inputs.tau_profile(0.124, 0.12, 120, 25, 17, L)[None]
    iu, idn = inputs.slab_indices(120, 25, 17, L)
    out = {}
    for mode in ("f64", "f64_full"):
        s = Solver(L, N, max_batch=1, max_orders=200)
        try:
            s.set_grid(mu); s.set_phase(P_atm, P_aer); s.set_contraction(mode)
            s.set_columns([iu], [idn], 0.5, 0.15, 1.0, 0.97, 0.124 / L, 0.12 / (idn + 1 - iu), 0.244)
            asym, uses = s.phase_asymmetry()
            assert uses == (mode == "f64")
            I1 = s.first_order(tau, p0[0.5][0][None], p0[0.5][1][None])
            out[mode] = (s.source(I1), s.solve(tau, p0[0.5][0][None], p0[0.5][1][None]))
        finally:
            s.close()
    assert asym <= 1e-13
    assert_close(out["f64"][0], out["f64_full"][0], 1e-13, "Jn: symmetric vs full product")
    assert out["f64"][1].n[0] == out["f64_full"][1].n[0]
    assert_close(out["f64"][1].I, out["f64_full"][1].I, 1e-12, "column: symmetric vs full product")


def test_sos_aer_defaults_name_the_eva_aerosol():
    """`SOS_Aer()` with the reference's literal names (spec:79-96: rayleigh + 'eva') at a small shape: phase functions built on
    the device, result equal to the oracle's on the oracle's evaluation of the same table."""
    L, N = 50, 32
    r = SOS_Aer(nb_layers=L, nb_angles=N, grd_alb=0.15, tauStar_atm=0.124)
    P_atm, P_aer, p0 = _oracle_inputs("eva", N, [0.5])
    col = O.make_column(0.5, 120, 25, 17, L, 0.124, 0.120, 0.15, 1.0, 1.0, N, p0[0.5][0], P_atm, p0[0.5][1], P_aer)
    ref = O.solve_column(col, literal=False)
    assert r.n == ref.n and r.I_saved.shape[0] == ref.n
    assert_close(r.I, ref.I, RTOL, "SOS_Aer() with the EVA aerosol")
