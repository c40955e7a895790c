"""Inputs of the hot path: optical-depth grid, direction grid, phase functions.

Builders for the arrays the reference's drivers construct before the order loop
(SOS_Aer_tau_profile.py, SOS_Aer_phase_func.py).  'iso', 'rayleigh', 'hg' and 'fwc' are
pinned to the reference's outputs at N = 32 and N = 128 (tests/golden/g5_*).  The
Mie-derived functions ('mie', 'eva', 'wildfire') follow the reference's recipe on top of
this package's own Mie series (sosrt/mie.py) because `miepython` is not available offline:
their parity is unpinned.

`phase_function(...)` is the host (NumPy) builder; `phase_function_device(...)` evaluates
the same azimuth averages with the HIP kernels of csrc/epilogue.hip -- P0(mu, mu0) for a
whole array of mu0 at once (a mu0 sweep needs a fresh P0 per column) and P(mu, mu').
"""
import os

import numpy as np

from . import mie as _mie

_FWC_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "fwc_table.npz")


def fwc_table():
    """(mu_fwc, phase_func_FWC): the tabulated fair-weather-cumulus phase function of SOS_Aer_fwc_data.py:3,173
    (1001 points on cos(Theta) = -1..1); data file written by tests/golden/make_golden.py."""
    d = np.load(_FWC_TABLE)
    return d["mu_fwc"], d["phase_func_FWC"]


def interpolate_table(mu_tab, p_tab, c):
    """SOS_Aer_phase_func.py:198-236 (interpolate_fwc_phase), vectorised."""
    c = np.clip(c, -1, 1)
    idx = np.searchsorted(mu_tab, c)
    lo = np.clip(idx - 1, 0, len(mu_tab) - 1)
    hi = np.clip(idx, 0, len(mu_tab) - 1)
    w = (c - mu_tab[lo]) / np.where(hi == lo, 1.0, mu_tab[hi] - mu_tab[lo])
    v = p_tab[lo] + w * (p_tab[hi] - p_tab[lo])
    return np.where(idx == 0, p_tab[0], np.where(idx >= len(mu_tab), p_tab[-1], v))

_trapz = getattr(np, "trapezoid", None) or np.trapz


def direction_grid(nb_angles):
    """[-1..0] then [0..1], mu=0 twice (SOS_Aer_main_specular.py:59-61)."""
    return np.concatenate((np.linspace(-1, 0, nb_angles), np.linspace(0, 1, nb_angles)))


def slab_indices(z0, z_up, z_down, nb_layers):
    """idx_up, idx_down of SOS_Aer_main_specular.py:30,39-40."""
    if z_down > z_up:
        z_down, z_up = z_up, z_down
    z = np.linspace(z0, 0, nb_layers)
    return int(np.argmin(np.abs(z - z_up))), int(np.argmin(np.abs(z - z_down)))


def tau_profile(tauStar_atm, tauStar_aer, z0, z_up, z_down, nb_layers):
    """Cumulative optical depth from the top (SOS_Aer_tau_profile.py:15-27), without the plot."""
    iu, idn = slab_indices(z0, z_up, z_down, nb_layers)
    tau = np.arange(0, nb_layers) * tauStar_atm / (nb_layers - 1)
    step = tauStar_aer / (idn + 1 - iu)
    k = np.arange(nb_layers)
    tau = tau + np.where(k < iu, 0.0, np.where(k <= idn, (k + 1 - iu) * step, tauStar_aer))
    return tau


def tau_profile_slabs(tauStar_atm, slabs, z0, nb_layers):
    """SOS_Aer_tau_profile.py:15-27 with several aerosol layers: `slabs` = [(z_up, z_down, tauStar_aer), ...]
    from the top down, each a linear ramp of its aerosol optical depth over its rows on top of the uniform molecular
    profile.  Returns (tau [L], zone_r0, zone_mix, zone_dtau_aer) -- the zone table `Solver.set_columns_zones` takes:
    clear and aerosol zones alternating.  One slab gives `tau_profile` and the reference's three zones."""
    L = int(nb_layers)
    z = np.linspace(z0, 0, L)
    tau = np.arange(0, L) * tauStar_atm / (L - 1)
    r0, mix, dta, prev = [0], [0], [0.0], 0
    for z_up, z_down, t_aer in slabs:
        if z_down > z_up:
            z_down, z_up = z_up, z_down
        iu, idn = int(np.argmin(np.abs(z - z_up))), int(np.argmin(np.abs(z - z_down)))
        if not (iu > prev and idn >= iu and idn <= L - 2):
            raise ValueError("aerosol layers must be given from the top down, separated by clear rows and above the last row")
        step = t_aer / (idn + 1 - iu)
        k = np.arange(L)
        tau = tau + np.where(k < iu, 0.0, np.where(k <= idn, (k + 1 - iu) * step, t_aer))
        r0 += [iu, idn + 1]
        mix += [1, 0]
        dta += [step, 0.0]
        prev = idn + 1
    return tau, np.array(r0, dtype=np.int32), np.array(mix, dtype=np.int32), np.array(dta)


def _ring_average(p, mu_a, mu_b, nb_phi=25):
    phi = np.linspace(0, np.pi, nb_phi)
    cc = mu_a[:, None] * mu_b[None, :]
    ss = np.sqrt(1 - mu_b * mu_b)[None, :] * np.sqrt(1 - mu_a * mu_a)[:, None]
    c = np.cos(0 - phi)
    return _trapz(p(-(cc[..., None] + ss[..., None] * c)) + p(-(cc[..., None] - ss[..., None] * c)), phi, axis=-1)


def _azimuth_averaged(p, mu, mu0):
    """(P0, P) of SOS_Aer_phase_func.py:79-133 for a scalar phase function p(cos Theta)."""
    P0 = _ring_average(p, mu, np.array([float(mu0)]))[:, 0] / (4 * np.pi)
    P0 = P0 / _trapz(P0, mu) * 2
    S = _ring_average(p, mu, mu) / (2 * np.pi)
    return P0, 4 * S / _trapz(S, mu, axis=0)[None, :]


_bulk_cache = {}


def _bulk(name, **kw):
    """(mu_diff [6001], p [6001]): the ensemble phase function of a scenario as a table on the scattering cosine.  The
    reference interpolates the table of every radius and integrates over the radii afterwards (phase:484-489, 738-744);
    the interpolation is linear in the ordinates, so interpolating the radius-integrated table gives the same number."""
    key = (name,) + tuple(sorted(kw.items()))
    if key not in _bulk_cache:
        _bulk_cache[key] = _mie.log_normal_bulk_phase(**kw)
    return _bulk_cache[key]


def scenario_table(name, lambda0=None, indx=None, r_m=None, sig=None, indx_convention="absorbing"):
    """The (mu_diff, p) table of 'eva' | 'wildfire' (README.md:95-111 unless overridden): what `phase_function` evaluates on
    the host and `phase_function_device` hands to the device table kernel (SOSRT_PHASE_TABLE).  Parity unpinned (own Mie
    series, sosrt/mie.py).  `indx_convention`: how the sign of Im(indx) is read -- 'absorbing' (default: any imaginary part
    absorbs), 'n+ik', 'n-ik' (miepython's documented sign); sosrt/mie.py."""
    kw = dict(_mie.SCENARIOS[name])
    for k, v in (("wl", lambda0), ("m", indx), ("r_m", r_m), ("sig", sig)):
        if v:
            kw[k] = v
    kw["m"] = _mie.refractive_index(kw["m"], indx_convention)          # (the cache key holds the index as the series takes it)
    return _bulk(name, convention="n+ik", **kw)


def _scalar_phase(name, g=0.0, r=None, lambda0=None, indx=None, r_m=None, sig=None, table=None, indx_convention="absorbing"):
    """p(cos Theta) as a NumPy callable, and -- for the device builders -- its (kind, table) form."""
    if name == "rayleigh":
        return (lambda c: (3 / 4) * (1 + c * c)), ("rayleigh", None)
    if name == "hg":
        return (lambda c: (1 - g * g) / ((1 + g * g - 2 * g * c) ** 1.5)), ("hg", None)
    if name == "fwc" or name == "table":
        mt, pt = fwc_table() if table is None else (np.asarray(table[0], dtype=np.float64), np.asarray(table[1], dtype=np.float64))
        return (lambda c: interpolate_table(mt, pt, c)), ("table", (mt, pt))
    if name in ("eva", "wildfire"):
        mt, pt = scenario_table(name, lambda0, indx, r_m, sig, indx_convention)
        return (lambda c: interpolate_table(mt, pt, c)), ("table", (mt, pt))
    if name == "mie":
        if not (r and lambda0 and indx):
            raise ValueError("'mie' needs r, lambda0 and indx")
        x = 2 * np.pi * r / lambda0
        mt = np.linspace(-1, 1, 6001)                       # phase:684-694 (compute_P's grid of scattering cosines)
        pt = _mie.i_unpolarized(complex(indx), x, mt, indx_convention)
        return (lambda c: interpolate_table(mt, pt, c)), ("table", (mt, pt))
    raise ValueError("unknown phase function %r" % (name,))


def phase_function(name, nb_angles, mu, mu0, g=0.0, r=None, lambda0=None, indx=None, r_m=None, sig=None, table=None,
                   indx_convention="absorbing"):
    """'iso' | 'rayleigh' | 'hg' | 'fwc' (SOS_Aer_phase_func.py:68,79,141,238; 'table' = 'fwc' with a caller's
    (mu_tab, p_tab)); 'mie' (one sphere of radius r, :299); 'eva' | 'wildfire' (log-normal ensemble, :398;
    parameters default to the README's scenarios).  Host (NumPy) evaluation."""
    if name == "iso":
        return np.ones(2 * nb_angles), 2 * np.ones((2 * nb_angles, 2 * nb_angles))
    fn, _ = _scalar_phase(name, g, r, lambda0, indx, r_m, sig, table, indx_convention)
    return _azimuth_averaged(fn, mu, mu0)


def phase_function_device(name, nb_angles, mu, mu0, g=0.0, r=None, lambda0=None, indx=None, r_m=None, sig=None, table=None,
                          matrix=True, device=0, solver=None, indx_convention="absorbing"):
    """The same on the GPU: `mu0` may be an array (one P0 row per column).  Returns (P0 [len(mu0), 2N] or [2N] for a
    scalar mu0, P [2N, 2N] or None when matrix=False).  Every name `phase_function` takes: 'iso' | 'rayleigh' | 'hg' |
    'fwc' | 'table' | 'mie' | 'eva' | 'wildfire' -- the Mie-derived ones as tables on the scattering cosine (the Mie series
    and the size integration run once on the host, sosrt/mie.py; the O(D^2 * 50) azimuth averages on the device)."""
    from .solver import Solver
    scalar = np.ndim(mu0) == 0
    m = np.atleast_1d(np.asarray(mu0, dtype=np.float64))
    kind, tab = ("iso", None) if name == "iso" else _scalar_phase(name, g, r, lambda0, indx, r_m, sig, table, indx_convention)[1]
    s = solver or Solver(2, nb_angles, max_batch=max(1, min(m.size, 4096)), max_orders=1, device=device)
    try:
        if not s.same_grid(mu):
            s.set_grid(mu)
        if tab is not None:
            s.set_phase_table(*tab)
        P0 = s.phase_p0(kind, m, g)
        P = s.phase_matrix(kind, g) if matrix else None
    finally:
        if solver is None:
            s.close()
    return (P0[0] if scalar else P0), P
