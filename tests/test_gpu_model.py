"""The restructured algorithm the kernels implement (GEMM-form Jn, tau recurrences, linear-map
extrapolation, blended restarts; tests/gpu_model.py) against the reference's goldens.  CPU only.
Proves that the restructuring itself stays inside the 1e-10 budget, independently of any kernel."""
import numpy as np
import pytest

import gpu_model as M
import sos_oracle as O
from util import assert_close, column_case, g1_case, golden

ALG = 5e-13


@pytest.mark.parametrize("path", golden("g1_*.npz"), ids=lambda p: p.split("/")[-1][3:-4])
def test_single_slab(path):
    d, N, P = g1_case(path)
    tau, mu, tS, alb = d["tau"], d["mu"], float(d["tauStar"]), float(d["alb"])
    L = len(tau)
    W = M.fold_weights(P, mu)
    In_1, n = d["I1"], 2
    while "In_%d" % n in d:
        Jn = M.source_model(In_1, W, W, np.full(L, alb / 4), np.zeros(L))
        assert_close(Jn, d["Jn_%d" % n], ALG, "Jn")
        # the flip-symmetric form of the contraction (two N x N products): same bar
        assert M.asymmetry(W) <= 1e-12
        assert_close(M.source_model(In_1, W, W, np.full(L, alb / 4), np.zeros(L), symmetric=True), d["Jn_%d" % n], ALG, "Jn (symmetric form)")
        In, st = M.transport_model(d["Jn_%d" % n], tau, mu, N, [(0, L - 1)], [M.a4b_count(tS, N)], None, 0.0)
        assert st == 0
        assert_close(In, d["In_%d" % n], ALG, "In")
        In, st = M.transport_model(d["Jn_%d" % n], tau, mu, N, [(0, L - 1)], [M.a4b_count(tS, N)], None, 0.0, chunk_local=True)
        assert st == 0
        assert_close(In, d["In_%d" % n], ALG, "In (chunk-local recurrences)")
        In_1 = d["In_%d" % n]
        n += 1


@pytest.mark.parametrize("symmetric,chunk_local", [(False, False), (True, False), (True, True)], ids=["full", "symmetric", "symmetric+chunk-local"])
@pytest.mark.parametrize("path", golden("g3_*.npz") + golden("g6_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_three_zone(path, symmetric, chunk_local):
    d, c = column_case(path)
    N, L, mu, tau = c["N"], c["L"], c["mu"], c["tau"]
    iu, idn = c["idx_up"], c["idx_down"]
    fa = c["dtau_atm"] / (c["dtau_atm"] + c["dtau_aer"])
    fr = c["dtau_aer"] / (c["dtau_atm"] + c["dtau_aer"])
    ca = np.full(L, c["alb_atm"] / 4)
    cr = np.zeros(L)
    ca[iu:idn + 1] *= fa
    cr[iu:idn + 1] = c["alb_aer"] / 4 * fr
    Wa, Wr = M.fold_weights(c["P_atm"], mu), M.fold_weights(c["P_aer"], mu)
    if symmetric:     # every phase matrix of the reference has the symmetry to rounding
        assert max(M.asymmetry(Wa), M.asymmetry(Wr)) <= 1e-12
    zones = [(0, iu - 1), (iu, idn), (idn + 1, L - 1)]
    nfix = [M.a4b_count(tau[iu - 1], N), M.a4b_count(tau[idn], N), M.a4b_count(tau[idn], N)]
    Isv = d["I_saved"]
    In_1, I, In, n = Isv[0], Isv[0].copy(), np.ones_like(Isv[0]), 1
    tol = 5e-11 if c["surface"] == "lambertian" else ALG   # sign-alternating terms (H2) amplify rounding
    while O.convergence_ratio(In, I, N) >= 1e-4:
        n += 1
        Jn = M.source_model(In_1, Wa, Wr, ca, cr, symmetric=symmetric)
        # chunk_local: the recurrences in the chunk-local form of transport_scan.hip (a value carried per chunk of 8 rows)
        In, st = M.transport_model(Jn, tau, mu, N, zones, nfix, c["surface"], c["grd_alb"], chunk_local=chunk_local)
        assert st == 0
        assert_close(In, Isv[n - 1], tol, "order %d" % n)
        In_1 = In
        I = I + In
    assert n == c["n"]
    assert_close(I, d["I"], tol, "I")


def test_symmetric_form_is_exact_on_a_symmetric_matrix_and_bounded_by_the_asymmetry():
    """The two N x N products equal the full product for a flip-symmetric matrix (to rounding); for a matrix with an
    asymmetric residual R the difference is the dropped term In_1 @ R -- the bound sosrt.h states."""
    rng = np.random.default_rng(5)
    N, L = 24, 9
    D = 2 * N
    G = rng.random((D, D))
    Ws = 0.5 * (G + G[::-1, ::-1])
    x = rng.random((L, D))
    assert M.asymmetry(Ws) == 0.0
    assert_close(M.source_symmetric(x, Ws), x @ Ws, 1e-14, "symmetric matrix")
    R = 1e-9 * (G - G[::-1, ::-1])
    W = Ws + R
    diff = np.abs(M.source_symmetric(x, W) - x @ W)
    bound = M.asymmetry(W) * np.max(np.abs(W)) * np.sum(np.abs(x), axis=1, keepdims=True)
    assert np.all(diff <= bound * (1 + 1e-6) + 1e-15)
    assert diff.max() > 1e-11          # (the test would be vacuous if the residual did nothing)


def test_index_error_is_modelled():
    """An upward row whose second differences never fall under 1e-4 makes the reference raise."""
    N, L = 16, 6
    mu = O.make_mu(N)
    tau = np.linspace(0, 0.3, L)
    Jn = np.zeros((L, 2 * N))
    Jn[:, N:] = 50.0 * (-1.0) ** np.arange(N)
    with pytest.raises(IndexError):
        O.In_NumInt(2, Jn, None, tau, mu, 0.3, 0.5, None, 1.0, N, literal=False)
    _, st = M.transport_model(Jn, tau, mu, N, [(0, L - 1)], [0], None, 0.0)
    assert st == 1
