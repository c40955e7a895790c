"""Shared helpers of the test-suite (golden loading, error norms)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerance of the north star: <= 1e-10 relative on radiance fields.  Element-wise relative error is
# taken on the elements that carry signal (|ref| > 1e-9 max|ref|); the rest are bounded in absolute
# terms by the same 1e-10 of the field maximum.
RTOL = 1e-10


def golden(pattern):
    files = sorted(glob.glob(os.path.join(GOLDEN, pattern)))
    assert files, "no golden fixture matches %s" % pattern
    return files


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b))
    if scale == 0:
        return float(np.max(np.abs(a)))
    sig = np.abs(b) > 1e-9 * scale
    e_rel = np.max(np.abs(a - b)[sig] / np.abs(b)[sig]) if sig.any() else 0.0
    e_abs = np.max(np.abs(a - b)) / scale
    return float(max(e_rel, e_abs))


def assert_close(a, b, tol=RTOL, what=""):
    assert np.shape(a) == np.shape(b), "%s shape %s vs %s" % (what, np.shape(a), np.shape(b))
    assert not np.any(np.isnan(a)), "%s has NaN" % what
    e = rel_err(a, b)
    assert e <= tol, "%s: relative error %.3e > %.1e" % (what, e, tol)
    return e


def g1_case(path):
    d = np.load(path)
    N = int(d["N"])
    P = d["P"] if "P" in d else float(d["P_const"]) * np.ones((2 * N, 2 * N))
    return d, N, P


def column_case(path):
    """Inputs of a G3/G4/G6 fixture as plain python values."""
    d = np.load(path)
    N = int(d["nb_angles"])
    D = 2 * N
    out = dict(N=N, L=int(d["nb_layers"]), mu=d["mu"], tau=d["tau"], idx_up=int(d["idx_up"]), idx_down=int(d["idx_down"]),
               mu0=float(d["mu0"]), grd_alb=float(d["grd_alb"]), alb_atm=float(d["alb_atm"]), alb_aer=float(d["alb_aer"]),
               dtau_atm=float(d["dtau_atm"]), dtau_aer=float(d["dtau_aer"]),
               tauStar_atm=float(d["tauStar_atm"]), tauStar_aer=float(d["tauStar_aer"]),
               z0=float(d["z0"]), z_up=float(d["z_up"]), z_down=float(d["z_down"]),
               surface=str(d["surface"]), n=int(d["n"]), P0_atm=d["P0_atm"], P0_aer=d["P0_aer"])
    for mol in ("atm", "aer"):
        if "P_%s" % mol in d:
            out["P_%s" % mol] = d["P_%s" % mol]
        elif "P_%s_const" % mol in d:
            out["P_%s" % mol] = float(d["P_%s_const" % mol]) * np.ones((D, D))
    return d, out


def oracle_column(O, c):
    return O.make_column(c["mu0"], c["z0"], c["z_up"], c["z_down"], c["L"], c["tauStar_atm"], c["tauStar_aer"],
                         c["grd_alb"], c["alb_atm"], c["alb_aer"], c["N"], c["P0_atm"], c["P_atm"], c["P0_aer"],
                         c["P_aer"], surface=c["surface"])
