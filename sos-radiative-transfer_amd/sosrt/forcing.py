"""Radiative forcing and critical single-scattering albedo of the aerosol layer: the reference's only
multi-column caller (SOS_Aer_critical_albedo.py:20-410), on top of the batched solve.

`toa_net_flux` is crit:377-382.  `radiative_forcing` is crit:384-389 with one defined change: the
reference's recursion for the aerosol-free baseline passes the *same* optical-depth grid and mixing
fractions, so its forcing is identically zero; here the baseline is the same column with
the aerosol removed (tauStar_aer = 0: no aerosol optical depth in the grid, f_aer = 0).
`critical_albedo` is the bisection of crit:394-410, run for many aerosol optical depths at once: every
bisection step is one batched solve.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .inputs import direction_grid, phase_function, slab_indices, tau_profile
from .main import get_solver


def _solve_fluxes(mu0, tauStar_atm, tauStar_aer, grd_alb, alb_atm, alb_aer, geom, phases, max_orders, device):
    """TOA net flux (crit:382) of B columns; arrays of length B for tauStar_aer and alb_aer."""
    z0, z_up, z_down, L, N = geom
    B = len(tauStar_aer)
    P0_atm, P_atm, P0_aer, P_aer = phases
    iu, idn = slab_indices(z0, z_up, z_down, L)
    tau = np.stack([tau_profile(tauStar_atm, t, z0, z_up, z_down, L) for t in tauStar_aer])
    s = get_solver(L, N, B, max_orders, device)
    mu = direction_grid(N)
    if not s.same_grid(mu):
        s.set_grid(mu)
    if not s.same_phase(P_atm, P_aer):
        s.set_phase(P_atm, P_aer)
    s.set_columns(np.full(B, iu), np.full(B, idn), mu0, grd_alb, alb_atm, alb_aer, tauStar_atm / L,
                  np.asarray(tauStar_aer) / (idn + 1 - iu), tauStar_atm + np.asarray(tauStar_aer))
    r = s.solve(tau, np.tile(P0_atm, (B, 1)), np.tile(P0_aer, (B, 1)))
    if np.any(r.status == _lib.COL_INDEXERROR):       # what the reference raises (spec:404)
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (2 * N, 2 * N))
    if np.any(r.status != 0):
        raise RuntimeError("columns did not converge within %d orders: status %s" % (max_orders, r.status))
    fd, fu = s.fluxes(tau, r.I, beam_norm="crit")
    return -fd[:, 0] - fu[:, 0]


def toa_net_flux(mu0, tauStar_atm, tauStar_aer, grd_alb, alb_atm, alb_aer, *, z0=120, z_up=25, z_down=17, nb_layers=200,
                 nb_angles=128, atm_phase_fun="rayleigh", g_atm=0.0, aer_phase_fun="hg", g_aer=0.7, phases=None,
                 max_orders=256, device=0):
    """Net flux at the top of the atmosphere, -F_down[0] - F_up[0] (crit:377-382), for arrays of
    (tauStar_aer, alb_aer)."""
    ta, wa = np.broadcast_arrays(np.atleast_1d(np.asarray(tauStar_aer, dtype=float)), np.atleast_1d(np.asarray(alb_aer, dtype=float)))
    if phases is None:
        mu = direction_grid(nb_angles)
        P0a, Pa = phase_function(atm_phase_fun, nb_angles, mu, mu0, g_atm)
        P0r, Pr = phase_function(aer_phase_fun, nb_angles, mu, mu0, g_aer)
        phases = (P0a, Pa, P0r, Pr)
    return _solve_fluxes(mu0, tauStar_atm, ta, grd_alb, alb_atm, wa, (z0, z_up, z_down, nb_layers, nb_angles), phases, max_orders, device)


def radiative_forcing(mu0, tauStar_atm, tauStar_aer, grd_alb, alb_atm, alb_aer, **kw):
    """TOA net flux with the aerosol layer minus the same column without it (crit:384-389, baseline fixed)."""
    ta = np.atleast_1d(np.asarray(tauStar_aer, dtype=float))
    with_aer = toa_net_flux(mu0, tauStar_atm, ta, grd_alb, alb_atm, alb_aer, **kw)
    base = toa_net_flux(mu0, tauStar_atm, [0.0], grd_alb, alb_atm, [1.0], **kw)[0]
    return with_aer - base


def critical_albedo(mu0, tauStar_atm, tauStar_aer, grd_alb, alb_atm, *, width=0.1, tol=1e-3, **kw):
    """Aerosol single-scattering albedo at which the forcing changes sign, by the bisection of crit:394-410
    (stop when the bracket is narrower than `width` or |forcing| < `tol`), for an array of tauStar_aer."""
    ta = np.atleast_1d(np.asarray(tauStar_aer, dtype=float))
    lo, hi = np.zeros_like(ta), np.ones_like(ta)
    result = np.full_like(ta, np.nan)
    base = toa_net_flux(mu0, tauStar_atm, [0.0], grd_alb, alb_atm, [1.0], **kw)[0]
    live = np.ones(len(ta), dtype=bool)
    while live.any() and np.any((hi - lo)[live] > width):
        test = (hi + lo) / 2
        f = np.zeros_like(ta)
        f[live] = toa_net_flux(mu0, tauStar_atm, ta[live], grd_alb, alb_atm, test[live], **kw) - base
        done = live & (np.abs(f) < tol)
        result[done] = test[done]
        live &= ~done
        up = live & (f > 0)
        lo[up] = test[up]
        hi[live & ~up] = test[live & ~up]
        narrow = live & ((hi - lo) <= width)
        result[narrow] = ((hi + lo) / 2)[narrow]
        live &= ~narrow
    result[np.isnan(result)] = ((hi + lo) / 2)[np.isnan(result)]
    return result
