"""Column-level drivers: the call surface of SOS_Aer_main_specular.py / SOS_Aer_main_lambertian.py.

`SOS_Aer(**overrides)` takes the reference's local names as keywords (spec:23-96); called
with no arguments it uses the literals the reference ships.  Unlike the reference (which
returns None and plots) it returns the radiance fields.  `SOS_Aer_batch` solves many
independent columns that share (nb_layers, nb_angles, phase matrices) in one launch sequence.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .inputs import direction_grid, phase_function, slab_indices, tau_profile
from .solver import Solver

# the literals of SOS_Aer_main_specular.py:23-96
DEFAULTS = dict(
    mu0=0.5, z0=120, z_up=25, z_down=17, nb_layers=800, tauStar_atm=0.104, tauStar_aer=0.120,
    grd_alb=1, alb_atm=1.0, alb_aer=1.0, nb_angles=501,
    atm_phase_fun="rayleigh", g_atm=0.5, r_atm=0, lambda0_atm=0, indx_atm=0, N0_atm=None, r_m_atm=None, sig_atm=None,
    aer_phase_fun="eva", g_aer=0.5, r_aer=0, lambda0_aer=0.550, indx_aer=1.44 + 0.0j, N0_aer=501187, r_m_aer=0.506,
    sig_aer=1.2,
)


@dataclass
class ColumnResult:
    I: np.ndarray            # [L, 2N] total radiance (spec:303,456)
    I_saved: np.ndarray      # [n, L, 2N] per-order fields (spec:304-305,458)
    n: int                   # final order (spec:307-310)
    tau: np.ndarray
    mu: np.ndarray
    idx_up: int
    idx_down: int
    status: int


@dataclass
class BatchResult:
    I: np.ndarray            # [B, L, 2N]
    n: np.ndarray            # [B]
    status: np.ndarray       # [B]
    tau: np.ndarray          # [B, L]
    mu: np.ndarray
    idx_up: int
    idx_down: int
    I_saved: Optional[np.ndarray] = None


_solvers = {}


def get_solver(nb_layers, nb_angles, batch, max_orders, device=0) -> Solver:
    import os
    key = (nb_layers, nb_angles, device, os.environ.get("SOSRT_TRANSPORT", ""))
    s = _solvers.get(key)
    if s is None or s.max_batch < batch or s.max_orders < max_orders:
        if s is not None:
            s.close()
        s = _solvers[key] = Solver(nb_layers, nb_angles, max_batch=batch, max_orders=max_orders, device=device)
    return s


def _raise_status(status, nb_angles):
    st = np.atleast_1d(status)
    if np.any(st == _lib.COL_INDEXERROR):
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (2 * nb_angles, 2 * nb_angles))
    if np.any(st == _lib.COL_INTERNAL):
        raise RuntimeError("sosrt: internal error in the transport kernel (SOSRT_COL_INTERNAL)")


def SOS_Aer_batch(mu0, tauStar_aer, grd_alb, *, tauStar_atm=0.124, alb_atm=1.0, alb_aer=1.0, z0=120, z_up=25, z_down=17,
                  nb_layers=200, nb_angles=128, atm_phase_fun="rayleigh", g_atm=0.0, aer_phase_fun="hg", g_aer=0.7,
                  P_atm=None, P_aer=None, P0_atm=None, P0_aer=None, surface="specular", tol=1e-4, max_orders=256,
                  save_orders=False, device=0, devices=None, raise_on_error=True, first_order="coded") -> BatchResult:
    """Solve B independent columns (arrays mu0, tauStar_aer, grd_alb broadcast to a common length;
    tauStar_atm, alb_atm, alb_aer may be arrays too).  `first_order='readme'`: the README's Lambertian first order
    (Solver.set_first_order; parity unpinned, single device).  `devices=[0, 1, ...]` shards the columns over several
    GPUs of the node, one worker process each, and gathers the fields (sosrt.dist.solve_on_devices; per-order
    fields are not gathered)."""
    if devices is not None and len(devices) > 1:
        if save_orders:
            raise ValueError("save_orders is not available with devices=[...]")
        if first_order != "coded":
            raise ValueError("first_order='readme' is not available with devices=[...]")
        from .dist import solve_on_devices
        r = solve_on_devices(devices, mu0, tauStar_aer, grd_alb, tauStar_atm=tauStar_atm, alb_atm=alb_atm, alb_aer=alb_aer,
                             z0=z0, z_up=z_up, z_down=z_down, nb_layers=nb_layers, nb_angles=nb_angles,
                             atm_phase_fun=atm_phase_fun, g_atm=g_atm, aer_phase_fun=aer_phase_fun, g_aer=g_aer, P_atm=P_atm,
                             P_aer=P_aer, P0_atm=P0_atm, P0_aer=P0_aer, surface=surface, tol=tol, max_orders=max_orders)
        if raise_on_error:
            _raise_status(r.status, int(nb_angles))
        return r
    if devices is not None and len(devices) == 1:
        device = int(devices[0])
    mu0, tauStar_aer, grd_alb, tauStar_atm, alb_atm, alb_aer = np.broadcast_arrays(
        *[np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in (mu0, tauStar_aer, grd_alb, tauStar_atm, alb_atm, alb_aer)])
    B = mu0.shape[0]
    L, N = int(nb_layers), int(nb_angles)
    if z_down > z_up:
        z_down, z_up = z_up, z_down
    mu = direction_grid(N)
    iu, idn = slab_indices(z0, z_up, z_down, L)
    tau = np.stack([tau_profile(tauStar_atm[b], tauStar_aer[b], z0, z_up, z_down, L) for b in range(B)])
    if P_atm is None:
        P_atm = phase_function(atm_phase_fun, N, mu, 0.5, g_atm)[1]
    if P_aer is None:
        P_aer = phase_function(aer_phase_fun, N, mu, 0.5, g_aer)[1]
    if P0_atm is None or P0_aer is None:
        cache = {}
        P0a = np.empty((B, 2 * N))
        P0r = np.empty((B, 2 * N))
        for b in range(B):
            k = float(mu0[b])
            if k not in cache:
                cache[k] = (phase_function(atm_phase_fun, N, mu, k, g_atm)[0] if P0_atm is None else None,
                            phase_function(aer_phase_fun, N, mu, k, g_aer)[0] if P0_aer is None else None)
            P0a[b] = cache[k][0] if P0_atm is None else np.broadcast_to(P0_atm, (B, 2 * N))[b]
            P0r[b] = cache[k][1] if P0_aer is None else np.broadcast_to(P0_aer, (B, 2 * N))[b]
    else:
        P0a = np.ascontiguousarray(np.broadcast_to(P0_atm, (B, 2 * N)))
        P0r = np.ascontiguousarray(np.broadcast_to(P0_aer, (B, 2 * N)))
    s = get_solver(L, N, B, max_orders, device)
    if not s.same_grid(mu):
        s.set_grid(mu)
    if not s.same_phase(P_atm, P_aer):
        s.set_phase(P_atm, P_aer)
    s.set_columns(np.full(B, iu), np.full(B, idn), mu0, grd_alb, alb_atm, alb_aer,
                  tauStar_atm / L, tauStar_aer / (idn + 1 - iu), tauStar_atm + tauStar_aer, surface=surface)
    s.set_first_order(first_order)
    try:
        r = s.solve(tau, P0a, P0r, tol=tol, save_orders=save_orders)
    finally:
        s.set_first_order("coded")                           # (the solver is cached)
    if raise_on_error:
        _raise_status(r.status, N)
    return BatchResult(I=r.I, n=r.n, status=r.status, tau=tau, mu=mu, idx_up=iu, idx_down=idn, I_saved=r.I_saved)


def SOS_Aer_layers(mu0, grd_alb, slabs, *, tauStar_atm=0.124, alb_atm=1.0, z0=120, nb_layers=200, nb_angles=128,
                   atm_phase_fun="rayleigh", g_atm=0.0, aer_phase_fun="hg", g_aer=0.7, P_atm=None, P_aer=None, surface="specular",
                   tol=1e-4, max_orders=256, device=0, raise_on_error=True) -> BatchResult:
    """Columns with SEVERAL aerosol layers (the reference has one): `slabs` = [(z_up, z_down, tauStar_aer,
    alb_aer), ...] from the top down, shared by the B columns of the arrays `mu0`, `grd_alb`.  Every formula of the path is
    evaluated per zone as the reference writes it for its three zones; one layer gives `SOS_Aer_batch`'s result bit for
    bit.  All layers share the aerosol phase function.  `idx_up` / `idx_down` of the result are those of the first layer."""
    from .inputs import tau_profile_slabs
    mu0, grd_alb = np.broadcast_arrays(np.atleast_1d(np.asarray(mu0, dtype=np.float64)), np.atleast_1d(np.asarray(grd_alb, dtype=np.float64)))
    B, L, N = mu0.shape[0], int(nb_layers), int(nb_angles)
    mu = direction_grid(N)
    tau, r0, mix, dta = tau_profile_slabs(tauStar_atm, [s[:3] for s in slabs], z0, L)
    zwr = np.zeros(len(r0))
    zwr[1::2] = [s[3] for s in slabs]
    if P_atm is None:
        P_atm = phase_function(atm_phase_fun, N, mu, 0.5, g_atm)[1]
    if P_aer is None:
        P_aer = phase_function(aer_phase_fun, N, mu, 0.5, g_aer)[1]
    s = get_solver(L, N, B, max_orders, device)
    if not s.same_grid(mu):
        s.set_grid(mu)
    if not s.same_phase(P_atm, P_aer):
        s.set_phase(P_atm, P_aer)
    # P0(mu, mu0) per column on the device for the analytic phase functions, on the host otherwise
    def p0(name, g):
        if name in ("iso", "rayleigh", "hg", "fwc"):
            if name == "fwc":
                from .inputs import fwc_table
                s.set_phase_table(*fwc_table())
            return s.phase_p0({"fwc": "table"}.get(name, name), mu0, g)
        cache = {float(m): phase_function(name, N, mu, float(m), g)[0] for m in np.unique(mu0)}
        return np.stack([cache[float(m)] for m in mu0])
    P0a, P0r = p0(atm_phase_fun, g_atm), p0(aer_phase_fun, g_aer)
    s.set_columns_zones(np.tile(r0, (B, 1)), mix, mu0, grd_alb, alb_atm, tauStar_atm / L, zwr, dta,
                        tauStar_atm + sum(x[2] for x in slabs), surface=surface)
    r = s.solve(np.tile(tau, (B, 1)), P0a, P0r, tol=tol)
    if raise_on_error:
        _raise_status(r.status, N)
    return BatchResult(I=r.I, n=r.n, status=r.status, tau=np.tile(tau, (B, 1)), mu=mu, idx_up=int(r0[1]), idx_down=int(r0[2]) - 1)


def SOS_Aer(surface="specular", tol=1e-4, max_orders=256, P_atm=None, P0_atm=None, P_aer=None, P0_aer=None, device=0,
            first_order="coded", **overrides) -> ColumnResult:
    """One column with the reference's parameter names (spec:19-96).  `surface` selects the file of
    the reference that would be run ('specular' | 'lambertian'); P*/P0* accept pre-built phase
    arrays."""
    unknown = set(overrides) - set(DEFAULTS)
    if unknown:
        raise TypeError("unknown parameter(s): %s" % ", ".join(sorted(unknown)))
    p = dict(DEFAULTS, **overrides)
    N, L = int(p["nb_angles"]), int(p["nb_layers"])
    mu = direction_grid(N)
    if P_atm is None or P0_atm is None:
        P0_atm, P_atm = phase_function(p["atm_phase_fun"], N, mu, p["mu0"], p["g_atm"], p["r_atm"], p["lambda0_atm"],
                                       p["indx_atm"], p["r_m_atm"], p["sig_atm"])
    if P_aer is None or P0_aer is None:
        P0_aer, P_aer = phase_function(p["aer_phase_fun"], N, mu, p["mu0"], p["g_aer"], p["r_aer"], p["lambda0_aer"],
                                       p["indx_aer"], p["r_m_aer"], p["sig_aer"])
    r = SOS_Aer_batch(p["mu0"], p["tauStar_aer"], p["grd_alb"], tauStar_atm=p["tauStar_atm"], alb_atm=p["alb_atm"],
                      alb_aer=p["alb_aer"], z0=p["z0"], z_up=p["z_up"], z_down=p["z_down"], nb_layers=L, nb_angles=N,
                      P_atm=P_atm, P_aer=P_aer, P0_atm=np.asarray(P0_atm)[None], P0_aer=np.asarray(P0_aer)[None],
                      surface=surface, tol=tol, max_orders=max_orders, save_orders=True, device=device, first_order=first_order)
    n = int(r.n[0])
    return ColumnResult(I=r.I[0], I_saved=r.I_saved[0, :n].copy(), n=n, tau=r.tau[0], mu=r.mu, idx_up=r.idx_up,
                        idx_down=r.idx_down, status=int(r.status[0]))
