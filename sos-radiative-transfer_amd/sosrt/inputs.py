"""Inputs of the hot path: optical-depth grid, direction grid, phase functions.

These are host-side (NumPy) builders for the arrays the reference's drivers
construct before the order loop (SOS_Aer_tau_profile.py, SOS_Aer_phase_func.py);
they are not on the timed path.  'iso', 'rayleigh' and 'hg' are pinned to the
reference's outputs (tests/golden/g5_*).  The Mie-derived functions ('mie',
'eva', 'wildfire') follow the reference's recipe on top of this package's own
Mie series (sosrt/mie.py) because `miepython` is not available offline: their
parity is unpinned.
"""
import numpy as np

from . import mie as _mie

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
    key = (name,) + tuple(sorted(kw.items()))
    if key not in _bulk_cache:
        _bulk_cache[key] = _mie.tabulated_phase(*_mie.log_normal_bulk_phase(**kw))
    return _bulk_cache[key]


def phase_function(name, nb_angles, mu, mu0, g=0.0, r=None, lambda0=None, indx=None, r_m=None, sig=None):
    """'iso' | 'rayleigh' | 'hg' (SOS_Aer_phase_func.py:68,79,141); 'mie' (one sphere of radius r, :299);
    'eva' | 'wildfire' (log-normal ensemble, :398; parameters default to the README's scenarios)."""
    if name == "iso":
        return np.ones(2 * nb_angles), 2 * np.ones((2 * nb_angles, 2 * nb_angles))
    if name == "rayleigh":
        return _azimuth_averaged(lambda c: (3 / 4) * (1 + c * c), mu, mu0)
    if name == "hg":
        return _azimuth_averaged(lambda c: (1 - g * g) / ((1 + g * g - 2 * g * c) ** 1.5), mu, mu0)
    if name in ("eva", "wildfire"):
        kw = dict(_mie.SCENARIOS[name])
        for k, v in (("wl", lambda0), ("m", indx), ("r_m", r_m), ("sig", sig)):
            if v:
                kw[k] = v
        return _azimuth_averaged(_bulk(name, **kw), mu, mu0)
    if name == "mie":
        if not (r and lambda0 and indx):
            raise ValueError("'mie' needs r, lambda0 and indx")
        x = 2 * np.pi * r / lambda0
        mu_d = np.linspace(-1, 1, 6001)
        return _azimuth_averaged(_mie.tabulated_phase(mu_d, _mie.i_unpolarized(complex(indx), x, mu_d)), mu, mu0)
    raise ValueError("unknown phase function %r" % (name,))
