"""Lorenz-Mie scattering by homogeneous spheres and the log-normal ensemble phase function that the
reference builds for its EVA (volcanic sulphate) and wildfire scenarios
(SOS_Aer_phase_func.py:398-681: `log_normal_mie`, `compute_P`, `interpolate_phase`).

The reference gets the single-sphere quantities from the third-party `miepython`
(`efficiencies`, `i_unpolarized`), which is not available offline and whose version it does not pin.
This module computes them from the published Mie series (Bohren & Huffman 1983, ch. 4; downward
recurrence of the logarithmic derivative as in Wiscombe 1980) and then follows the reference's recipe
for the ensemble.  Nothing here can be checked against the reference's own outputs: **parity
unpinned**.  What is pinned is the series itself: tests/test_host.py asserts the single-sphere
efficiencies against the published values of Wiscombe (1979, NCAR/TN-140+STR: m = 1.5, 0.75, 1.5 - i,
10 - 10i at x = 1 ... 1000) and Bohren & Huffman's appendix-A case to six digits, plus physical
identities (Rayleigh limit, optical theorem, normalisation, extinction paradox).

Sign of the imaginary part of the refractive index.  The series below is written for a time factor
exp(-i w t), in which an ABSORBING sphere has m = n + ik with k > 0 (Bohren & Huffman); `miepython`
documents the other sign, m = n - ik.  The reference writes `1.7 + 0.03j` for its absorbing wildfire
aerosol (README.md:109) and hands it to `miepython` unchanged (phase:334-335,419) -- what that means
under the reference's own, unpinned dependency cannot be known here.  This module therefore takes the
physical reading by default: any non-zero imaginary part is absorption (`convention="absorbing"`,
m -> n + i|k|), so `1.7 + 0.03j` and `1.7 - 0.03j` give the same, absorbing, sphere; `"n+ik"` and
`"n-ik"` take the sign literally under the named convention (a gain medium if it is the wrong one),
for callers who want exactly that.  The choice is an argument of every public function here and of
`inputs.phase_function(..., indx_convention=)`.

Inputs of the hot path only -- host NumPy, not on the timed path.
"""
from __future__ import annotations

import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz


CONVENTIONS = ("absorbing", "n+ik", "n-ik")


def refractive_index(indx, convention: str = "absorbing") -> complex:
    """The index as the series below wants it (m = n + ik, k > 0 absorbs) from a caller's value under `convention`:
    'absorbing' (default) any imaginary part is absorption; 'n+ik' taken as is; 'n-ik' (miepython's documented sign)
    conjugated."""
    m = complex(indx)
    if convention == "absorbing":
        return complex(m.real, abs(m.imag))
    if convention == "n+ik":
        return m
    if convention == "n-ik":
        return m.conjugate()
    raise ValueError("convention must be one of %s" % (CONVENTIONS,))


def mie_coefficients(m: complex, x: float):
    """a_n, b_n for n = 1..nmax (Bohren & Huffman eq. 4.88), nmax = x + 4 x^(1/3) + 2.  m = n + ik, k > 0 absorbs."""
    nmax = int(np.round(x + 4.0 * x ** (1.0 / 3.0) + 2.0))
    mx = m * x
    # logarithmic derivative D_n(mx) by downward recurrence
    nmx = int(max(nmax, abs(mx)) + 16)
    D = np.zeros(nmx + 1, dtype=complex)
    for n in range(nmx, 0, -1):
        D[n - 1] = n / mx - 1.0 / (D[n] + n / mx)
    # Riccati-Bessel functions by upward recurrence
    psi0, psi1 = np.cos(x), np.sin(x)
    chi0, chi1 = -np.sin(x), np.cos(x)
    xi1 = complex(psi1, -chi1)
    a = np.zeros(nmax, dtype=complex)
    b = np.zeros(nmax, dtype=complex)
    for n in range(1, nmax + 1):
        psi = (2.0 * n - 1.0) / x * psi1 - psi0
        chi = (2.0 * n - 1.0) / x * chi1 - chi0
        xi = complex(psi, -chi)
        da = D[n] / m + n / x
        db = D[n] * m + n / x
        a[n - 1] = (da * psi - psi1) / (da * xi - xi1)
        b[n - 1] = (db * psi - psi1) / (db * xi - xi1)
        psi0, psi1 = psi1, psi
        chi0, chi1 = chi1, chi
        xi1 = complex(psi1, -chi1)
    return a, b


def efficiencies(m: complex, x: float, convention: str = "absorbing"):
    """Q_ext, Q_sca, Q_back, g for one sphere (`convention`: sign of Im m, see the module docstring)."""
    m = refractive_index(m, convention)
    a, b = mie_coefficients(m, x)
    n = np.arange(1, len(a) + 1)
    qext = 2.0 / x ** 2 * np.sum((2 * n + 1) * (a + b).real)
    qsca = 2.0 / x ** 2 * np.sum((2 * n + 1) * (np.abs(a) ** 2 + np.abs(b) ** 2))
    qback = np.abs(np.sum((2 * n + 1) * (-1.0) ** n * (a - b))) ** 2 / x ** 2
    g = 4.0 / (qsca * x ** 2) * (np.sum(n[:-1] * (n[:-1] + 2.0) / (n[:-1] + 1.0) * (a[:-1] * np.conj(a[1:]) + b[:-1] * np.conj(b[1:])).real)
                                 + np.sum((2 * n + 1.0) / (n * (n + 1.0)) * (a * np.conj(b)).real))
    return float(qext), float(qsca), float(qback), float(g)


def amplitudes(m: complex, x: float, mu: np.ndarray, convention: str = "absorbing"):
    """S1(mu), S2(mu), mu = cos(scattering angle) (Bohren & Huffman eq. 4.74)."""
    m = refractive_index(m, convention)
    a, b = mie_coefficients(m, x)
    mu = np.asarray(mu, dtype=np.float64)
    pi0 = np.zeros_like(mu)
    pi1 = np.ones_like(mu)
    S1 = np.zeros(mu.shape, dtype=complex)
    S2 = np.zeros(mu.shape, dtype=complex)
    for n in range(1, len(a) + 1):
        tau = n * mu * pi1 - (n + 1) * pi0
        f = (2.0 * n + 1.0) / (n * (n + 1.0))
        S1 += f * (a[n - 1] * pi1 + b[n - 1] * tau)
        S2 += f * (a[n - 1] * tau + b[n - 1] * pi1)
        pi0, pi1 = pi1, ((2 * n + 1.0) * mu * pi1 - (n + 1.0) * pi0) / n
    return S1, S2


def i_unpolarized(m: complex, x: float, mu: np.ndarray, convention: str = "absorbing"):
    """Unpolarised scattered intensity normalised so that its integral over 4 pi steradians is the
    single-scattering albedo Q_sca/Q_ext (the default 'albedo' normalisation of miepython 2.x)."""
    m = refractive_index(m, convention)
    S1, S2 = amplitudes(m, x, mu, "n+ik")
    qext, _, _, _ = efficiencies(m, x, "n+ik")
    return (np.abs(S1) ** 2 + np.abs(S2) ** 2) / 2.0 / (np.pi * x ** 2 * qext)


def log_normal_bulk_phase(wl, m, r_m, sig, nb_radius=100, r_min=0.01, r_max=10.0, nb_mu=6001, convention="absorbing"):
    """Ensemble phase function on a grid of scattering cosines, following phase:403-422,684-694:
    radii linspace(0.01, 10) micrometres, n(r) = exp(-(ln r - ln r_m)^2 / (2 ln^2 sig)) / r,
    weight n(r) Q_sca(r), trapezoid over r.  Returns (mu_diff[nb_mu], p[nb_mu]) un-normalised (every
    consumer normalises)."""
    m = refractive_index(m, convention)
    radii = np.linspace(r_min, r_max, nb_radius)
    n_r = (1.0 / radii) * np.exp(-((np.log(radii) - np.log(r_m)) ** 2) / (2 * np.log(sig) ** 2))
    xs = 2 * np.pi * radii / wl
    mu_d = np.linspace(-1, 1, nb_mu)
    P = np.empty((nb_radius, nb_mu))
    qsca = np.empty(nb_radius)
    for i, x in enumerate(xs):
        qsca[i] = efficiencies(m, x, "n+ik")[1]
        P[i] = i_unpolarized(m, x, mu_d, "n+ik")
    w = n_r * qsca
    return mu_d, _trapz(w[:, None] * P, radii, axis=0)


def tabulated_phase(mu_d, p):
    """p(cos Theta) by linear interpolation in the table, arguments clipped to [-1, 1] (phase:696-711)."""
    def f(c):
        return np.interp(np.clip(c, -1.0, 1.0), mu_d, p)
    return f


# scenario parameters of the reference README (README.md:95-111) and of SOS_Aer_main_specular.py:88-94
SCENARIOS = {
    "eva": dict(wl=0.550, m=1.44 + 0.0j, r_m=0.506, sig=1.2),
    "wildfire": dict(wl=0.550, m=1.7 + 0.03j, r_m=0.065, sig=1.5),
}
