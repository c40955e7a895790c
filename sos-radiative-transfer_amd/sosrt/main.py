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
from .inputs import direction_grid, slab_indices, tau_profile
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
    if s.order_budget != max_orders:                 # (a cached handle made for a larger budget serves this call with its own)
        s.set_order_budget(max_orders)
    return s


def _raise_status(status, nb_angles):
    st = np.atleast_1d(status)
    if np.any(st == _lib.COL_INDEXERROR):
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (2 * nb_angles, 2 * nb_angles))
    if np.any(st == _lib.COL_INTERNAL):
        raise RuntimeError("sosrt: internal error in the transport kernel (SOSRT_COL_INTERNAL)")


def device_phase(s: Solver, name, mu0, g=0.0, mie=None, matrix=True):
    """(P0 rows [len(mu0), 2N], P [2N, 2N] or None) of a named phase function, azimuth-averaged by the HIP kernels of solver
    `s` (phase:79-133 and its siblings; the grid must be set).  `mie` = dict(r=, lambda0=, indx=, r_m=, sig=) for 'mie' /
    'eva' / 'wildfire', whose phase function goes to the device as a table on the scattering cosine."""
    from .inputs import _scalar_phase
    kind, tab = ("iso", None) if name == "iso" else _scalar_phase(name, g, **(mie or {}))[1]
    if tab is not None:
        s.set_phase_table(*tab)
    P0 = s.phase_p0(kind, mu0, g)
    return P0, (s.phase_matrix(kind, g) if matrix else None)


def SOS_Aer_batch(mu0, tauStar_aer, grd_alb, *, tauStar_atm=0.124, alb_atm=1.0, alb_aer=1.0, z0=120, z_up=25, z_down=17,
                  nb_layers=200, nb_angles=128, atm_phase_fun="rayleigh", g_atm=0.0, aer_phase_fun="hg", g_aer=0.7,
                  mie_atm=None, mie_aer=None,
                  P_atm=None, P_aer=None, P0_atm=None, P0_aer=None, surface="specular", tol=1e-4, max_orders=256,
                  save_orders=False, device=0, devices=None, raise_on_error=True, first_order="coded") -> BatchResult:
    """Solve B independent columns (arrays mu0, tauStar_aer, grd_alb broadcast to a common length;
    tauStar_atm, alb_atm, alb_aer may be arrays too).  Phase functions that are not handed in as arrays are built on the
    device (`device_phase`): any name of `inputs.phase_function`, `mie_atm` / `mie_aer` = dict(r=, lambda0=, indx=, r_m=,
    sig=) for the Mie-derived ones ('eva' and 'wildfire' default to the README's scenarios).
    `first_order='readme'`: the README's Lambertian first order
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
                             atm_phase_fun=atm_phase_fun, g_atm=g_atm, aer_phase_fun=aer_phase_fun, g_aer=g_aer,
                             mie_atm=mie_atm, mie_aer=mie_aer, P_atm=P_atm,
                             P_aer=P_aer, P0_atm=P0_atm, P0_aer=P0_aer, surface=surface, tol=tol, max_orders=max_orders)
        if raise_on_error:
            _raise_status(r.status, int(nb_angles))
        return r
    if devices is not None and len(devices) == 1:
        device = int(devices[0])
    s, tau, P0a, P0r, mu, iu, idn, N = _prepare_batch(mu0, tauStar_aer, grd_alb, tauStar_atm, alb_atm, alb_aer, z0, z_up, z_down,
                                                      nb_layers, nb_angles, atm_phase_fun, g_atm, aer_phase_fun, g_aer, mie_atm,
                                                      mie_aer, P_atm, P_aer, P0_atm, P0_aer, surface, max_orders, device)
    s.set_first_order(first_order)
    try:
        r = s.solve(tau, P0a, P0r, tol=tol, save_orders=save_orders)
    finally:
        s.set_first_order("coded")                           # (the solver is cached)
    if raise_on_error:
        _raise_status(r.status, N)
    return BatchResult(I=r.I, n=r.n, status=r.status, tau=tau, mu=mu, idx_up=iu, idx_down=idn, I_saved=r.I_saved)


def _prepare_batch(mu0, tauStar_aer, grd_alb, tauStar_atm, alb_atm, alb_aer, z0, z_up, z_down, nb_layers, nb_angles,
                   atm_phase_fun, g_atm, aer_phase_fun, g_aer, mie_atm, mie_aer, P_atm, P_aer, P0_atm, P0_aer, surface,
                   max_orders, device, p0_on_host=True):
    """Everything before the order loop (spec:23-96): optical-depth grids, direction grid, phase matrices folded into the
    handle, per-column scalars.  Returns the solver and the per-column inputs."""
    mu0, tauStar_aer, grd_alb, tauStar_atm, alb_atm, alb_aer = np.broadcast_arrays(
        *[np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in (mu0, tauStar_aer, grd_alb, tauStar_atm, alb_atm, alb_aer)])
    B = mu0.shape[0]
    L, N = int(nb_layers), int(nb_angles)
    if z_down > z_up:
        z_down, z_up = z_up, z_down
    mu = direction_grid(N)
    iu, idn = slab_indices(z0, z_up, z_down, L)
    tau = np.stack([tau_profile(tauStar_atm[b], tauStar_aer[b], z0, z_up, z_down, L) for b in range(B)])
    s = get_solver(L, N, B, max_orders, device)
    if not s.same_grid(mu):
        s.set_grid(mu)
    # the inputs of the path that are not handed in: P0(mu, mu0[b]) per column and P(mu, mu') on the device
    # (phase:79-133; the Mie-derived functions as tables on the scattering cosine)
    P0a = P0r = None
    if P_atm is None or P0_atm is None:
        P0a, Pm = device_phase(s, atm_phase_fun, mu0, g_atm, mie_atm, matrix=P_atm is None)
        P_atm = Pm if P_atm is None else P_atm
    if P_aer is None or P0_aer is None:
        P0r, Pm = device_phase(s, aer_phase_fun, mu0, g_aer, mie_aer, matrix=P_aer is None)
        P_aer = Pm if P_aer is None else P_aer
    if P0_atm is not None:
        P0a = np.ascontiguousarray(np.broadcast_to(P0_atm, (B, 2 * N)))
    if P0_aer is not None:
        P0r = np.ascontiguousarray(np.broadcast_to(P0_aer, (B, 2 * N)))
    if not s.same_phase(P_atm, P_aer):
        s.set_phase(P_atm, P_aer)
    s.set_columns(np.full(B, iu), np.full(B, idn), mu0, grd_alb, alb_atm, alb_aer,
                  tauStar_atm / L, tauStar_aer / (idn + 1 - iu), tauStar_atm + tauStar_aer, surface=surface)
    return s, tau, P0a, P0r, mu, iu, idn, N


def solve_batch_device(mu0, tauStar_aer, grd_alb, *, tauStar_atm=0.124, alb_atm=1.0, alb_aer=1.0, z0=120, z_up=25, z_down=17,
                       nb_layers=200, nb_angles=128, atm_phase_fun="rayleigh", g_atm=0.0, aer_phase_fun="hg", g_aer=0.7,
                       mie_atm=None, mie_aer=None, P_atm=None, P_aer=None, P0_atm=None, P0_aer=None, surface="specular",
                       tol=1e-4, max_orders=256, device=0):
    """`SOS_Aer_batch` with the RESULT LEFT ON THE DEVICE: returns ({"I": [B, L, 2N] float64, "n": [B] int32, "status": [B]
    int32, "tau": [B, L]} as torch tensors on cuda:`device`, (mu, idx_up, idx_down)).  The solve is enqueued on torch's
    current stream for that device, so the tensors are ordered like any torch result.  What `dist.solve_sharded` gathers."""
    import torch
    s, tau, P0a, P0r, mu, iu, idn, N = _prepare_batch(mu0, tauStar_aer, grd_alb, tauStar_atm, alb_atm, alb_aer, z0, z_up, z_down,
                                                      nb_layers, nb_angles, atm_phase_fun, g_atm, aer_phase_fun, g_aer, mie_atm,
                                                      mie_aer, P_atm, P_aer, P0_atm, P0_aer, surface, max_orders, device)
    dev = torch.device("cuda", device)
    B, L = tau.shape
    with torch.cuda.device(dev):
        s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        try:
            d_tau = torch.from_numpy(tau).to(dev)
            d_P0a = torch.from_numpy(np.ascontiguousarray(P0a)).to(dev)
            d_P0r = torch.from_numpy(np.ascontiguousarray(P0r)).to(dev)
            d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev)
            d_n = torch.zeros(B, dtype=torch.int32, device=dev)
            d_st = torch.zeros(B, dtype=torch.int32, device=dev)
            s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), tol=tol,
                           d_n_orders=d_n.data_ptr(), d_status=d_st.data_ptr())
        finally:
            # the solver is cached and goes back to its own stream: drain this one first, so that the handle's internal
            # buffers are not reused under the last launches (the order loop has already waited for all but the last two)
            s.synchronize()
            s.set_stream(None)
    return {"I": d_I, "n": d_n, "status": d_st, "tau": d_tau}, (mu, iu, idn)


def SOS_Aer_layers(mu0, grd_alb, slabs, *, tauStar_atm=0.124, alb_atm=1.0, z0=120, nb_layers=200, nb_angles=128,
                   atm_phase_fun="rayleigh", g_atm=0.0, aer_phase_fun="hg", g_aer=0.7, mie_atm=None, mie_aer=None, P_atm=None,
                   P_aer=None, surface="specular", tol=1e-4, max_orders=256, device=0, raise_on_error=True) -> BatchResult:
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
    s = get_solver(L, N, B, max_orders, device)
    if not s.same_grid(mu):
        s.set_grid(mu)
    # P0(mu, mu0) per column, and the matrices that were not handed in, on the device
    P0a, Pm = device_phase(s, atm_phase_fun, mu0, g_atm, mie_atm, matrix=P_atm is None)
    P_atm = Pm if P_atm is None else P_atm
    P0r, Pm = device_phase(s, aer_phase_fun, mu0, g_aer, mie_aer, matrix=P_aer is None)
    P_aer = Pm if P_aer is None else P_aer
    if not s.same_phase(P_atm, P_aer):
        s.set_phase(P_atm, P_aer)
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
    mie = {k: dict(r=p["r_" + k], lambda0=p["lambda0_" + k], indx=p["indx_" + k], r_m=p["r_m_" + k], sig=p["sig_" + k])
           for k in ("atm", "aer")}
    r = SOS_Aer_batch(p["mu0"], p["tauStar_aer"], p["grd_alb"], tauStar_atm=p["tauStar_atm"], alb_atm=p["alb_atm"],
                      alb_aer=p["alb_aer"], z0=p["z0"], z_up=p["z_up"], z_down=p["z_down"], nb_layers=L, nb_angles=N,
                      atm_phase_fun=p["atm_phase_fun"], g_atm=p["g_atm"], aer_phase_fun=p["aer_phase_fun"], g_aer=p["g_aer"],
                      mie_atm=mie["atm"], mie_aer=mie["aer"], P_atm=P_atm, P_aer=P_aer,
                      P0_atm=None if P0_atm is None else np.asarray(P0_atm)[None],
                      P0_aer=None if P0_aer is None else np.asarray(P0_aer)[None],
                      surface=surface, tol=tol, max_orders=max_orders, save_orders=True, device=device, first_order=first_order)
    n = int(r.n[0])
    return ColumnResult(I=r.I[0], I_saved=r.I_saved[0, :n].copy(), n=n, tau=r.tau[0], mu=r.mu, idx_up=r.idx_up,
                        idx_down=r.idx_down, status=int(r.status[0]))
