"""Radiative forcing and critical single-scattering albedo of the aerosol layer: the reference's only
multi-column caller (SOS_Aer_critical_albedo.py:20-410), on top of the batched solve.

Two layers:

* drop-in: `SOS_Aer_radiative_forcing(...)` and `SOS_Aer_critical_albedo(...)` with the reference's positional
  signatures (crit:20, crit:394).  `baseline="coded"` reproduces the file as shipped: its recursion for the
  aerosol-free term passes the *same* optical-depth grid, mixing fractions and phase arrays (crit:388), so it
  recomputes the same column, the forcing is exactly 0.0 and the bisection stops at its first probe, 0.5
  (pinned by tests/golden/g7_*).  The default `baseline="no_aerosol"` is the defined fix: the baseline is the
  same column with the aerosol removed (its own grid tau_atm only, f_aer = 0).
* batched: `toa_net_flux`, `radiative_forcing`, `critical_albedo` over arrays of (tauStar_aer, alb_aer); every
  bisection step is one batched solve.

The radiance field never leaves the device: the solve keeps it resident and the flux epilogue
(csrc/epilogue.hip, crit:377-382) returns B numbers.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .inputs import direction_grid, phase_function, slab_indices, tau_profile
from .main import get_solver


def _net_flux_columns(tau, idx_up, idx_down, mu, mu0, grd_alb, alb_atm, alb_aer, dtau_atm, dtau_aer, tauStar_tot, P_atm, P_aer,
                      P0_atm, P0_aer, max_orders=256, device=0):
    """TOA net flux (crit:382) of B columns on their grids tau [B, L]; per-column arrays or scalars otherwise."""
    tau = np.atleast_2d(np.asarray(tau, dtype=np.float64))
    B, L = tau.shape
    N = len(mu) // 2
    s = get_solver(L, N, B, max_orders, device)
    if not s.same_grid(mu):
        s.set_grid(mu)
    if not s.same_phase(P_atm, P_aer):
        s.set_phase(P_atm, P_aer)
    s.set_columns(np.full(B, idx_up), np.full(B, idx_down), mu0, grd_alb, alb_atm, alb_aer, dtau_atm, dtau_aer, tauStar_tot)
    P0a = np.ascontiguousarray(np.broadcast_to(np.asarray(P0_atm, dtype=np.float64), (B, 2 * N)))
    P0r = np.ascontiguousarray(np.broadcast_to(np.asarray(P0_aer, dtype=np.float64), (B, 2 * N)))
    r = s.solve(tau, P0a, P0r, fetch_field=False)
    if np.any(r.status == _lib.COL_INDEXERROR):       # what the reference raises (spec:404)
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (2 * N, 2 * N))
    if np.any(r.status != 0):
        raise RuntimeError("columns did not converge within %d orders: status %s" % (max_orders, r.status))
    return s.epilogue(want=("net_toa",))["net_toa"]


# ---------------------------------------------------------------------------------------------
# drop-in surface (crit:20, crit:394)
# ---------------------------------------------------------------------------------------------
def SOS_Aer_radiative_forcing(tauStar_aer, dtau_aer, tauStar_atm, dtau_atm, P_aer, P0_aer, alb_aer, P_atm, P0_atm, alb_atm,
                              grd_alb, F0, mu, mu0, nb_angles, tau, nb_layers, idx_up, idx_down, *, baseline="no_aerosol",
                              tauStar_tot=None, max_orders=256, device=0):
    """crit:20-389.  `tauStar_aer == 0` returns the net flux at the top of the atmosphere of the column described by
    the other arguments (crit:384-385); otherwise the forcing (crit:387-389).  `tauStar_tot` stands for the module
    global the reference's function reads (crit:39); default tauStar_atm + tauStar_aer.  `F0` must be pi/mu0
    (crit:469), as everywhere in the reference."""
    if baseline not in ("coded", "no_aerosol"):
        raise ValueError("baseline must be 'coded' or 'no_aerosol'")
    if abs(F0 - np.pi / mu0) > 1e-12 * F0:
        raise ValueError("F0 must be pi/mu0 (crit:469)")
    tot = tauStar_atm + tauStar_aer if tauStar_tot is None else tauStar_tot
    args = dict(P_atm=P_atm, P_aer=P_aer, P0_atm=P0_atm, P0_aer=P0_aer, max_orders=max_orders, device=device)
    f = float(_net_flux_columns(tau, idx_up, idx_down, mu, mu0, grd_alb, alb_atm, alb_aer, dtau_atm, dtau_aer, tot, **args)[0])
    if tauStar_aer == 0:
        return f
    if baseline == "coded":
        return f - f                                  # crit:388 solves the same column again: exactly 0.0
    tau0 = np.arange(0, nb_layers) * tauStar_atm / (nb_layers - 1)          # taup:21 with no aerosol
    f0 = float(_net_flux_columns(tau0, idx_up, idx_down, mu, mu0, grd_alb, alb_atm, alb_aer, dtau_atm, 0.0, tauStar_atm, **args)[0])
    return f - f0


def SOS_Aer_critical_albedo(tauStar_aer, dtau_aer, tauStar_atm, dtau_atm, P_aer, P0_aer, P_atm, P0_atm, alb_atm, grd_alb, F0, mu,
                            mu0, nb_angles, tau, nb_layers, idx_up, idx_down, *, baseline="no_aerosol", **kw):
    """crit:394-410: bisection on the aerosol single-scattering albedo until the bracket is narrower than 0.1 or
    |forcing| < 0.001."""
    alb_max, alb_min = 1, 0
    while (alb_max - alb_min) > 0.1:
        test = (alb_max + alb_min) / 2
        f = SOS_Aer_radiative_forcing(tauStar_aer, dtau_aer, tauStar_atm, dtau_atm, P_aer, P0_aer, test, P_atm, P0_atm, alb_atm,
                                      grd_alb, F0, mu, mu0, nb_angles, tau, nb_layers, idx_up, idx_down, baseline=baseline, **kw)
        if np.abs(f) < 0.001:
            return test
        if f > 0:
            alb_min = test
        else:
            alb_max = test
    return (alb_max + alb_min) / 2


# ---------------------------------------------------------------------------------------------
# batched over (tauStar_aer, alb_aer)
# ---------------------------------------------------------------------------------------------
def _solve_fluxes(mu0, tauStar_atm, tauStar_aer, grd_alb, alb_atm, alb_aer, geom, phases, max_orders, device):
    """TOA net flux (crit:382) of B columns; arrays of length B for tauStar_aer and alb_aer."""
    z0, z_up, z_down, L, N = geom
    P0_atm, P_atm, P0_aer, P_aer = phases
    iu, idn = slab_indices(z0, z_up, z_down, L)
    tau = np.stack([tau_profile(tauStar_atm, t, z0, z_up, z_down, L) for t in tauStar_aer])
    ta = np.asarray(tauStar_aer, dtype=np.float64)
    return _net_flux_columns(tau, iu, idn, direction_grid(N), mu0, grd_alb, alb_atm, alb_aer, tauStar_atm / L,
                             ta / (idn + 1 - iu), tauStar_atm + ta, P_atm, P_aer, P0_atm, P0_aer, max_orders, device)


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


def radiative_forcing(mu0, tauStar_atm, tauStar_aer, grd_alb, alb_atm, alb_aer, *, baseline="no_aerosol", **kw):
    """TOA net flux with the aerosol layer minus the baseline (crit:384-389): the same column without the aerosol
    ('no_aerosol', default) or, as the reference is coded, the same column again ('coded': identically zero)."""
    ta = np.atleast_1d(np.asarray(tauStar_aer, dtype=float))
    with_aer = toa_net_flux(mu0, tauStar_atm, ta, grd_alb, alb_atm, alb_aer, **kw)
    if baseline == "coded":
        return with_aer - with_aer
    base = toa_net_flux(mu0, tauStar_atm, [0.0], grd_alb, alb_atm, [1.0], **kw)[0]
    return with_aer - base


def critical_albedo(mu0, tauStar_atm, tauStar_aer, grd_alb, alb_atm, *, width=0.1, tol=1e-3, baseline="no_aerosol", **kw):
    """Aerosol single-scattering albedo at which the forcing changes sign, by the bisection of crit:394-410
    (stop when the bracket is narrower than `width` or |forcing| < `tol`), for an array of tauStar_aer."""
    ta = np.atleast_1d(np.asarray(tauStar_aer, dtype=float))
    lo, hi = np.zeros_like(ta), np.ones_like(ta)
    result = np.full_like(ta, np.nan)
    base = None if baseline == "coded" else toa_net_flux(mu0, tauStar_atm, [0.0], grd_alb, alb_atm, [1.0], **kw)[0]
    live = np.ones(len(ta), dtype=bool)
    while live.any() and np.any((hi - lo)[live] > width):
        test = (hi + lo) / 2
        f = np.zeros_like(ta)
        fl = toa_net_flux(mu0, tauStar_atm, ta[live], grd_alb, alb_atm, test[live], **kw)
        f[live] = fl - (fl if base is None else base)
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
