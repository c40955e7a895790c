"""Drop-in for the reference's step functions (SOS_Aer_I1_In.py:13,62,77): same names,
same positional signatures, same array layout [layer][direction], results from the HIP
kernels.  A handle is cached per (nb_layers, nb_angles); the direction grid and the
phase matrix are re-uploaded only when they change.
"""
import numpy as np

from . import _lib
from .In_limit import (improved_asymptotic_downward_radiance, improved_limit_mu_down, limit_mu_down,  # noqa: F401
                       mu_approx_In)
from .global_va import MU_THRESHOLD, MU_EXTREME_THRESHOLD, MU_VERY_SMALL_THRESHOLD  # noqa: F401
from .solver import Solver

_handles = {}


def _handle(tau, mu, nb_angles) -> Solver:
    L = len(tau)
    mu = np.asarray(mu, dtype=np.float64)
    if mu.shape != (2 * nb_angles,):
        raise ValueError("mu has shape %s, expected (%d,)" % (mu.shape, 2 * nb_angles))
    s = _handles.get((L, nb_angles))
    if s is None:
        s = _handles[(L, nb_angles)] = Solver(L, nb_angles, max_batch=1, max_orders=1)
    if not s.same_grid(mu):
        s.set_grid(mu)
    return s


def I1_NumInt(tau, mu, tauStar, mu0, P0, alb, nb_angles):
    """First order of scattering for a single slab over a black surface (SOS_Aer_I1_In.py:13-58)."""
    s = _handle(tau, mu, nb_angles)
    s.set_columns_single_slab(mu0, alb, tauStar)
    return s.first_order(np.asarray(tau, dtype=np.float64)[None], np.asarray(P0, dtype=np.float64)[None])[0]


def Jn_NumInt(n, In_1, tau, mu, tauStar, mu0, P, alb, nb_angles):
    """Source function of order n (SOS_Aer_I1_In.py:62-74); n, tau, tauStar, mu0 are unused, as in the reference."""
    s = _handle(tau, mu, nb_angles)
    if not s.same_phase(P):
        s.set_phase(P)
    s.set_columns_single_slab(mu0, alb, tauStar)
    return s.source(np.asarray(In_1, dtype=np.float64)[None])[0]


def In_NumInt(n, Jn, In_1, tau, mu, tauStar, mu0, P, alb, nb_angles, µ_1=None, µ_2=None):
    """Radiance of order n (SOS_Aer_I1_In.py:77-130).  Raises IndexError where the reference's
    unbounded upward search (I1_In:103) runs off the grid."""
    s = _handle(tau, mu, nb_angles)
    s.set_columns_single_slab(mu0, alb, tauStar)
    In, st = s.transport(np.asarray(tau, dtype=np.float64)[None], np.asarray(Jn, dtype=np.float64)[None])
    if st[0] == _lib.COL_INDEXERROR:
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (2 * nb_angles, 2 * nb_angles))
    return In[0]
