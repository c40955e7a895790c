"""CPU oracle for the SOS radiative-transfer hot path  --  TEST INFRASTRUCTURE.

This file is a from-scratch NumPy restatement of the reference algorithm
(Guillaume-SOULIER/SOS-Radiative-Transfer, snapshot 2025-09-05).  It exists so
that the HIP path can be checked against something that runs on the GPU box,
where the reference itself is not present.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it; the product (package `sosrt`) never does.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function
here against fixtures in `tests/golden/` that were produced by running the
reference's own code in the build container (`tests/golden/make_golden.py`).

Every public function cites the reference lines it follows, as
`file:line` relative to the reference root, with the aliases
  I1_In = SOS_Aer_I1_In.py      In_limit = SOS_Aer_In_limit.py
  spec  = SOS_Aer_main_specular.py   lam = SOS_Aer_main_lambertian.py
  gva   = SOS_Aer_global_va.py  taup = SOS_Aer_tau_profile.py
  phase = SOS_Aer_phase_func.py graphe = SOS_Aer_graphe.py
  crit  = SOS_Aer_critical_albedo.py

The three-zone column of the reference mains (above / inside / below the
aerosol slab, spec:104-458) is written here once, driven by a zone table,
instead of three times.  `literal=True` keeps the reference's per-(layer,
angle) scalar trapezoid calls in the downward transport (its dominant cost)
so that timing this oracle is representative of timing the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

# gva:5-7
MU_THRESHOLD = 0.01
MU_EXTREME_THRESHOLD = 1e-8
MU_VERY_SMALL_THRESHOLD = 0.001

_trapz = getattr(np, "trapezoid", None) or np.trapz  # the reference calls np.trapz


# ---------------------------------------------------------------------------
# grids
# ---------------------------------------------------------------------------
def make_mu(N: int) -> np.ndarray:
    """Direction grid of spec:59-61: -1..0 then 0..1, mu=0 stored twice."""
    return np.concatenate((np.linspace(-1, 0, N), np.linspace(0, 1, N)))


def slab_indices(z0, z_up, z_down, L) -> Tuple[int, int]:
    """spec:30,39-40."""
    if z_down > z_up:
        z_down, z_up = z_up, z_down
    z = np.linspace(z0, 0, L)
    return int(np.argmin(np.abs(z - z_up))), int(np.argmin(np.abs(z - z_down)))


def tau_profile(tauStar_atm, tauStar_aer, z0, z_up, z_down, L) -> np.ndarray:
    """Cumulative optical depth grid, taup:15-27 (no plotting)."""
    iu, idn = slab_indices(z0, z_up, z_down, L)
    tau = np.arange(0, L) * tauStar_atm / (L - 1)
    d_aer = tauStar_aer / (idn + 1 - iu)
    for i in range(iu, L):
        tau[i] += (i + 1 - iu) * d_aer if i <= idn else tauStar_aer
    return tau


def tau_profile_slabs(tauStar_atm, slabs, z0, L):
    """taup:15-27 with several aerosol layers: `slabs` = [(z_up, z_down, tauStar_aer), ...] from the top down, each a
    linear ramp of its aerosol optical depth over its rows on top of the uniform molecular profile.  Returns
    (tau, [(idx_up, idx_down), ...]).  One slab gives tau_profile()."""
    z = np.linspace(z0, 0, L)
    tau = np.arange(0, L) * tauStar_atm / (L - 1)
    rows = []
    for z_up, z_down, t_aer in slabs:
        if z_down > z_up:
            z_down, z_up = z_up, z_down
        iu, idn = int(np.argmin(np.abs(z - z_up))), int(np.argmin(np.abs(z - z_down)))
        d_aer = t_aer / (idn + 1 - iu)
        for i in range(iu, L):
            tau[i] += (i + 1 - iu) * d_aer if i <= idn else t_aer
        rows.append((iu, idn))
    return tau, rows


def mu_approx_In(mu, N):
    """In_limit:145-153 (result is unused by the reference's transport)."""
    i = N
    while mu[i] < 0.009:
        i += 1
    m1 = i
    while mu[i] < 0.020:
        i += 1
    return m1, i


# ---------------------------------------------------------------------------
# mu -> 0 helpers
# ---------------------------------------------------------------------------
def improved_asymptotic_downward_radiance(J, tau_s, tau_t, mu):
    """In_limit:70-109: downward radiance for |mu| < MU_THRESHOLD."""
    if len(tau_s) == 0:
        return 0.0
    if abs(mu) < MU_VERY_SMALL_THRESHOLD:          # both Taylor branches are identical (:79-93)
        slope = (J[-1] - J[-2]) / (tau_s[-1] - tau_s[-2]) if len(tau_s) > 1 else 0.0
        return -J[-1] + mu * slope
    keep = np.where(tau_s >= (tau_t - 5 * abs(mu)))[0]
    if len(keep) == 0:
        return -J[-1]
    with np.errstate(all="ignore"):
        f = J[keep] * np.exp((tau_t - tau_s[keep]) / mu)
    if np.any(np.isinf(f)) or np.any(np.isnan(f)):
        return -J[-1]
    return -_trapz(f, tau_s[keep]) / mu


def limit_mu_down(row, mu_down, N, idx, i):
    """In_limit:155-158: straight line through the two angles beyond the rewritten ones."""
    slope = (row[-idx - 2] - row[-idx - 1]) / (mu_down[-idx - 2] - mu_down[-idx - 1])
    return slope * (mu_down[-i - 1] - mu_down[-idx - 1]) + row[-idx - 1]


def improved_limit_mu_down(row, mu_down, N, idx, i):
    """In_limit:113-141: value at mu_down[-i-1] extrapolated from the n=min(5,idx) angles
    just outside the rewritten block."""
    n = min(5, idx)
    if n < 2:
        return limit_mu_down(row, mu_down, N, idx, i)
    x = np.array(mu_down[-(idx + n):-idx], dtype=np.float64)
    y = np.array(row[-(idx + n):-idx], dtype=np.float64)
    xe = float(mu_down[-i - 1])
    if len(x) >= 3:
        return np.polyval(np.polyfit(x, y, min(2, len(x) - 1)), xe)
    return y[0] + (y[-1] - y[0]) / (x[-1] - x[0]) * (xe - x[0])


def a4b_count(tau_ref, N) -> int:
    """Number of downward angles next to mu=0 that are rewritten (I1_In:124-127, spec:342-345)."""
    if tau_ref <= 0.0625:
        c = 0.005
    elif tau_ref <= 1:
        c = 0.02
    elif tau_ref < 4:
        c = 0.04
    else:
        c = 0.06
    return int(c * N)


def _rewrite_small_down(In_row, mu, N, idx):
    for i in range(idx):
        In_row[N - 1 - i] = improved_limit_mu_down(In_row[:N], mu[:N], N, idx, i)


def _blend_small_up(In_row, mu, N):
    """spec:402-409 / I1_In:101-108.  Raises IndexError like the reference when the
    second-difference test never passes."""
    k = N + 1
    while np.abs((In_row[k] - In_row[k + 1]) - (In_row[k + 1] - In_row[k + 2])) > 0.0001:
        k += 1
    k += 1
    for m in range(N + 1, k):
        w = mu[m] / mu[k]
        In_row[m] = (1 - w) * In_row[N] + w * In_row[k]


# ---------------------------------------------------------------------------
# single-slab function API  (I1_In:13,62,77)
# ---------------------------------------------------------------------------
def I1_NumInt(tau, mu, tauStar, mu0, P0, alb, N):
    """I1_In:13-58."""
    L = len(tau)
    I1 = np.zeros((L, 2 * N))
    e0 = np.exp(-tau / mu0)
    eS = np.exp(-tauStar / mu0)
    c = alb / (4 * np.pi)
    md = mu[:N - 1]
    near = np.abs(md + mu0) < 0.0001
    with np.errstate(all="ignore"):
        for t in range(L):
            I1[t, :N - 1] = (mu0 / (mu0 + md)) * c * P0[:N - 1] * (e0[t] - np.exp(tau[t] / md))
            I1[t, N - 1] = c * (mu0 / (mu0 + mu[N - 1])) * P0[N - 1] * e0[t]
            if near.any():
                I1[t, :N - 1][near] = c * P0[:N - 1][near] * e0[t] * tau[t] / mu0
            I1[t, N] = c * (mu0 / (mu0 + mu[N])) * P0[N] * e0[t]
            mp = mu[N + 1:]
            I1[t, N + 1:] = (mu0 / (mu0 + mp)) * c * P0[N + 1:] * (e0[t] - eS * np.exp(-(tauStar - tau[t]) / mp))
    return I1 * np.pi / mu0


def Jn_NumInt(n, In_1, tau, mu, tauStar, mu0, P, alb, N):
    """I1_In:62-74."""
    L = len(tau)
    Jn = np.zeros((L, 2 * N))
    Pf = P[:, ::-1]
    for t in range(L):
        Jn[t, :] = (alb / 4) * _trapz(Pf * In_1[t, :], mu, axis=1)
    return Jn


def In_NumInt(n, Jn, In_1, tau, mu, tauStar, mu0, P, alb, N, mu_1=None, mu_2=None, literal=True):
    """I1_In:77-130."""
    z = [_Zone(0, len(tau) - 1, "atm")]
    return _transport(Jn, tau, mu, N, z, tau_ref=[tauStar], thick_test=tauStar / mu[N + 1],
                      surface=None, grd_alb=0.0, literal=literal)


# ---------------------------------------------------------------------------
# three-zone column  (spec:104-458, lam:399/401)
# ---------------------------------------------------------------------------
@dataclass
class _Zone:
    r0: int
    r1: int
    kind: str  # 'atm' | 'mix'
    alb_aer: float = 0.0       # an aerosol zone's own single-scattering albedo and optical-depth step (zone tables)
    dtau_aer: float = 0.0


@dataclass
class Column:
    """One independent SOS problem (the locals of spec:23-96 that the hot path reads)."""
    tau: np.ndarray
    mu: np.ndarray
    N: int
    idx_up: int
    idx_down: int
    mu0: float
    grd_alb: float
    alb_atm: float
    alb_aer: float
    dtau_atm: float
    dtau_aer: float
    tauStar_tot: float
    P0_atm: np.ndarray
    P_atm: np.ndarray
    P0_aer: np.ndarray
    P_aer: np.ndarray
    surface: str = "specular"      # 'specular' | 'lambertian' (coded sign of lam:399/401) | 'lambertian_readme'
    # Generalisation beyond the reference: an explicit zone table, clear and aerosol zones alternating,
    # each aerosol zone with its own albedo and optical-depth step.  None = the reference's three zones.  The reference
    # has one slab only, so more than one is PARITY UNPINNED by construction; with the three zones of the reference the
    # table path is the same arithmetic (checked bit for bit in tests/test_oracle_golden.py).
    zone_table: Optional[List[_Zone]] = None
    first_order: str = "coded"     # 'coded' (spec:104-292, what both mains compute) | 'readme' (README.md:126-171, Lambertian; unpinned)

    @property
    def zones(self) -> List[_Zone]:
        if self.zone_table is not None:
            return self.zone_table
        L = len(self.tau)
        return [_Zone(0, self.idx_up - 1, "atm"), _Zone(self.idx_up, self.idx_down, "mix", self.alb_aer, self.dtau_aer),
                _Zone(self.idx_down + 1, L - 1, "atm")]

    @property
    def f_atm(self):
        return self.dtau_atm / (self.dtau_atm + self.dtau_aer)

    @property
    def f_aer(self):
        return self.dtau_aer / (self.dtau_atm + self.dtau_aer)

    def zone_fractions(self, z: _Zone):
        """(f_atm, f_aer) of an aerosol zone, spec:149."""
        return self.dtau_atm / (self.dtau_atm + z.dtau_aer), z.dtau_aer / (self.dtau_atm + z.dtau_aer)

    def tau_ref(self) -> List[float]:
        """tau_ref of every zone (spec:342,361,380): the zone's own last row for the top zone and for a slab, the last row
        of the slab above for a clear zone below one."""
        zs = self.zones
        return [self.tau[z.r1] if (i == 0 or z.kind == "mix") else self.tau[z.r0 - 1] for i, z in enumerate(zs)]


def make_column(mu0, z0, z_up, z_down, nb_layers, tauStar_atm, tauStar_aer, grd_alb, alb_atm, alb_aer,
                nb_angles, P0_atm, P_atm, P0_aer, P_aer, surface="specular") -> Column:
    """spec:23-62 (grid and derived scalars)."""
    if z_down > z_up:
        z_down, z_up = z_up, z_down
    tau = tau_profile(tauStar_atm, tauStar_aer, z0, z_up, z_down, nb_layers)
    iu, idn = slab_indices(z0, z_up, z_down, nb_layers)
    return Column(tau=tau, mu=make_mu(nb_angles), N=nb_angles, idx_up=iu, idx_down=idn, mu0=mu0,
                  grd_alb=grd_alb, alb_atm=alb_atm, alb_aer=alb_aer,
                  dtau_atm=tauStar_atm / nb_layers, dtau_aer=tauStar_aer / (idn + 1 - iu),
                  tauStar_tot=tauStar_atm + tauStar_aer,
                  P0_atm=P0_atm, P_atm=P_atm, P0_aer=P0_aer, P_aer=P_aer, surface=surface)


def make_column_slabs(mu0, z0, slabs, nb_layers, tauStar_atm, grd_alb, alb_atm, nb_angles, P0_atm, P_atm, P0_aer, P_aer,
                      surface="specular") -> Column:
    """A column with several aerosol layers: `slabs` = [(z_up, z_down, tauStar_aer, alb_aer), ...] from the top down."""
    tau, rows = tau_profile_slabs(tauStar_atm, [s[:3] for s in slabs], z0, nb_layers)
    zt, prev = [], 0
    for (iu, idn), sl in zip(rows, slabs):
        assert iu > prev and idn >= iu, "slabs must be separated by clear rows"
        zt.append(_Zone(prev, iu - 1, "atm"))
        zt.append(_Zone(iu, idn, "mix", sl[3], sl[2] / (idn + 1 - iu)))
        prev = idn + 1
    assert prev <= nb_layers - 1
    zt.append(_Zone(prev, nb_layers - 1, "atm"))
    t_aer = sum(s[2] for s in slabs)
    return Column(tau=tau, mu=make_mu(nb_angles), N=nb_angles, idx_up=rows[0][0], idx_down=rows[0][1], mu0=mu0, grd_alb=grd_alb,
                  alb_atm=alb_atm, alb_aer=slabs[0][3], dtau_atm=tauStar_atm / nb_layers, dtau_aer=zt[1].dtau_aer,
                  tauStar_tot=tauStar_atm + t_aer, P0_atm=P0_atm, P_atm=P_atm, P0_aer=P0_aer, P_aer=P_aer, surface=surface,
                  zone_table=zt)


def first_order(c: Column) -> np.ndarray:
    """Three-zone first order with the specularly reflected beam, spec:104-292."""
    tau, mu, N, mu0 = c.tau, c.mu, c.N, c.mu0
    L = len(tau)
    F0 = np.pi / mu0
    T = c.tauStar_tot
    R = F0 * c.grd_alb * np.exp(-T / mu0)
    q_atm = c.alb_atm * c.P0_atm / (4 * np.pi)

    def q_of(z):
        if z.kind != "mix":
            return q_atm
        fa, fr = c.zone_fractions(z)
        return (c.alb_atm * c.P0_atm * fa + z.alb_aer * c.P0_aer * fr) / (4 * np.pi)
    mirror = 2 * N - 1 - np.arange(2 * N)
    I1 = np.zeros((L, 2 * N))
    zones = c.zones
    with np.errstate(all="ignore"):
        # downward, top zone first
        md = mu[:N - 1]
        near = np.abs(md + mu0) < 0.0001
        for zi, z in enumerate(zones):
            q = q_of(z)
            qd, qdm = q[:N - 1], q[mirror[:N - 1]]
            if zi == 0:
                t_bd = t_bs = 0.0
            else:
                t_bd, t_bs = tau[z.r0 - 1], tau[z.r0]
            for t in range(z.r0, z.r1 + 1):
                before = 0.0 if zi == 0 else I1[z.r0 - 1, :N - 1] * np.exp((tau[t] - t_bd) / md)
                direct = (mu0 / (mu0 + md)) * qd * F0 * (np.exp(-tau[t] / mu0) - np.exp(-t_bd / mu0) * np.exp((tau[t] - t_bd) / md))
                direct_near = qd * F0 * np.exp(-tau[t] / mu0) * (tau[t] - t_bd) / mu0
                surf = (mu0 / (mu0 - md)) * qdm * R * (np.exp(-(T - tau[t]) / mu0) - np.exp(-(T - t_bs) / mu0) * np.exp((tau[t] - t_bs) / md))
                I1[t, :N - 1] = before + np.where(near, direct_near, direct) + surf
                I1[t, N - 1] = (mu0 / (mu0 + mu[N - 1])) * q[N - 1] * F0 * np.exp(-tau[t] / mu0) \
                    + (mu0 / (mu0 - mu[N - 1])) * q[N] * R * np.exp(-(T - tau[t]) / mu0)
        # upward, bottom zone first
        mp = mu[N + 1:]
        near = np.abs(mp - mu0) < 0.0001
        for zi in range(len(zones) - 1, -1, -1):
            z = zones[zi]
            q = q_of(z)
            qu, qum = q[N + 1:], q[mirror[N + 1:]]
            bottom = zi == len(zones) - 1
            if bottom:
                t_bu, t_su = tau[L - 1], T
                B = c.grd_alb * I1[L - 1, mirror[N + 1:]]
            else:
                t_bu, t_su = tau[z.r1 + 1], tau[z.r1]
            for t in range(z.r0, z.r1 + 1):
                if not bottom:
                    B = I1[z.r1 + 1, N + 1:]
                before = B * np.exp(-(t_bu - tau[t]) / mp)
                direct = (mu0 / (mu0 + mp)) * qu * F0 * (np.exp(-tau[t] / mu0) - np.exp(-t_bu / mu0) * np.exp(-(t_bu - tau[t]) / mp))
                surf = (mu0 / (mu0 - mp)) * qum * R * (np.exp(-(T - tau[t]) / mu0) - np.exp(-(T - t_su) / mu0) * np.exp(-(t_su - tau[t]) / mp))
                surf_near = qum * R * np.exp(-(T - tau[t]) / mu0) * (t_su - tau[t]) / mu0
                I1[t, N + 1:] = before + direct + np.where(near, surf_near, surf)
                I1[t, N] = (mu0 / (mu0 + mu[N])) * q[N] * F0 * np.exp(-tau[t] / mu0) \
                    + (mu0 / (mu0 - mu[N])) * q[N - 1] * R * np.exp(-(T - tau[t]) / mu0)
    return I1


def first_order_extended(c: Column) -> np.ndarray:
    """`first_order` with every intermediate in numpy's long double (x86-64: 64-bit mantissa) on the same float64 inputs,
    rounded to float64 once at the end.  Not a second oracle: a yardstick for the ROUNDING NOISE of the reference's own
    arithmetic.  spec:113-292 divides a difference of exponentials by (mu0 + mu) and switches to the limit form only within
    |mu0 - |mu|| < 1e-4 (spec:126,...): a direction that lies just outside that window (say 1.6e-4 from mu0) carries the
    rounding of the two exponentials amplified by mu0 / 1.6e-4 = 5000 -- a few 1e-13 absolute, up to 2e-10 relative to a
    small element -- in the reference itself.  A float64 implementation that does not replay the reference's exact sequence
    of roundings (numpy's exp included) cannot agree with it more closely than that there; tools/fuzz_parity.py and
    tests/test_gpu_parity.py compare both against this evaluation when a column has such a direction."""
    import copy
    ld = np.longdouble
    c2 = copy.copy(c)
    for k in ("tau", "mu", "P0_atm", "P0_aer"):
        setattr(c2, k, np.asarray(getattr(c, k)).astype(ld))
    for k in ("mu0", "grd_alb", "alb_atm", "alb_aer", "dtau_atm", "dtau_aer", "tauStar_tot"):
        setattr(c2, k, ld(getattr(c, k)))
    return np.asarray(first_order(c2), dtype=np.float64)


def first_order_lambertian_readme(c: Column) -> np.ndarray:
    """First order over a Lambertian surface as the reference's README writes it (README.md:126-171).  PARITY UNPINNED: the
    reference ships no runnable code for it -- `lam:274-276` crashes and the other first-order blocks of that
    file are the specular-beam formulas.  Non-default option (`first_order='readme'`).

    Per zone and direction: attenuated boundary row + single scattering of the direct beam (the reference's own
    `scatt_direct`, spec:113-292) + single scattering of the direct beam reflected isotropically by the ground,

        int_0^1  mu'/(mu'-mu)  omega P(mu,-mu')/(4 pi)  2 rho F0 e^{-T/mu0}  ( e^{-(T-tau)/mu'} - e^{-(T-tau_b)/mu'} e^{-|tau-tau_b|/|mu|} ) dmu'

    by the trapezoid rule on the upward half of the direction grid (the grid the reference integrates on everywhere),
    with the README's constant as written.  The integrand's removable singularity at mu' = mu (upward directions lie on
    the quadrature nodes) is replaced by its limit  omega P/(4 pi) 2 rho F0 e^{-T/mu0} e^{-(T-tau)/mu} (tau_b-tau)/mu.
    `tau_b` is the level the zone is entered at (the row above going down, the row below -- the surface row for the bottom
    zone -- going up), the same level for the beam and the surface term.  Upward boundary at the ground: isotropic
    2 rho int_0^1 I1_down(T, -mu') mu' dmu' (README.md:151, taken positive as in README.md:215)."""
    tau, mu, N, mu0 = c.tau, c.mu, c.N, c.mu0
    L = len(tau)
    F0 = np.pi / mu0
    T = c.tauStar_tot
    R2 = 2 * c.grd_alb * F0 * np.exp(-T / mu0)
    mirror = 2 * N - 1 - np.arange(2 * N)
    mup = mu[N:]                                   # quadrature nodes mu' = 0 .. 1
    wq = np.zeros(N)                               # trapezoid weights on them
    wq[:-1] += np.diff(mup) / 2
    wq[1:] += np.diff(mup) / 2

    def q_of(z):
        if z.kind != "mix":
            return c.alb_atm * c.P0_atm / (4 * np.pi), c.alb_atm * c.P_atm / (4 * np.pi)
        fa, fr = c.zone_fractions(z)
        return ((c.alb_atm * c.P0_atm * fa + z.alb_aer * c.P0_aer * fr) / (4 * np.pi),
                (c.alb_atm * c.P_atm * fa + z.alb_aer * c.P_aer * fr) / (4 * np.pi))
    I1 = np.zeros((L, 2 * N))
    zones = c.zones
    with np.errstate(all="ignore"):
        md = mu[:N - 1]
        near_d = np.abs(md + mu0) < 0.0001
        for zi, z in enumerate(zones):
            q, Q = q_of(z)
            Qk = Q[:, mirror[N:]]                  # radiation travelling along +mu'_k scattered into direction m (Jn's convention)
            t_b = 0.0 if zi == 0 else tau[z.r0 - 1]
            for t in range(z.r0, z.r1 + 1):
                att = np.exp((tau[t] - t_b) / md)
                before = 0.0 if zi == 0 else I1[z.r0 - 1, :N - 1] * att
                direct = (mu0 / (mu0 + md)) * q[:N - 1] * F0 * (np.exp(-tau[t] / mu0) - np.exp(-t_b / mu0) * att)
                direct_near = q[:N - 1] * F0 * np.exp(-tau[t] / mu0) * (tau[t] - t_b) / mu0
                e1 = np.where(mup > 0, np.exp(-(T - tau[t]) / np.where(mup > 0, mup, 1.0)), 0.0)
                e2 = np.where(mup > 0, np.exp(-(T - t_b) / np.where(mup > 0, mup, 1.0)), 0.0)
                ker = mup[None, :] / (mup[None, :] - md[:, None])                       # mu' - mu > 0 for downward mu
                lam = (wq[None, :] * ker * Qk[:N - 1] * (e1[None, :] - e2[None, :] * att[:, None])).sum(axis=1) * R2
                I1[t, :N - 1] = before + np.where(near_d, direct_near, direct) + lam
                # mu = 0-: the second exponential vanishes, mu'/(mu'-mu) = 1
                I1[t, N - 1] = q[N - 1] * F0 * np.exp(-tau[t] / mu0) + (wq * Qk[N - 1] * e1).sum() * R2
        mp = mu[N + 1:]
        for zi in range(len(zones) - 1, -1, -1):
            z = zones[zi]
            q, Q = q_of(z)
            Qk = Q[:, mirror[N:]]
            bottom = zi == len(zones) - 1
            if bottom:
                t_b = tau[L - 1]
                w_dn = np.zeros(N)
                w_dn[:-1] += np.diff(mu[:N]) / 2
                w_dn[1:] += np.diff(mu[:N]) / 2
                B = 2 * c.grd_alb * (w_dn * I1[L - 1, :N] * (-mu[:N])).sum() * np.ones(N - 1)
            else:
                t_b = tau[z.r1 + 1]
            for t in range(z.r0, z.r1 + 1):
                if not bottom:
                    B = I1[z.r1 + 1, N + 1:]
                att = np.exp(-(t_b - tau[t]) / mp)
                direct = (mu0 / (mu0 + mp)) * q[N + 1:] * F0 * (np.exp(-tau[t] / mu0) - np.exp(-t_b / mu0) * att)
                e1 = np.where(mup > 0, np.exp(-(T - tau[t]) / np.where(mup > 0, mup, 1.0)), 0.0)
                e2 = np.where(mup > 0, np.exp(-(T - t_b) / np.where(mup > 0, mup, 1.0)), 0.0)
                diff = mup[None, :] - mp[:, None]
                same = np.abs(diff) < 0.0001                                            # the node mu' = mu itself
                ker = mup[None, :] / np.where(same, 1.0, diff)
                term = ker * (e1[None, :] - e2[None, :] * att[:, None])
                lim = (np.exp(-(T - tau[t]) / mp) * (t_b - tau[t]) / mp)[:, None]
                lam = (wq[None, :] * Qk[N + 1:] * np.where(same, lim, term)).sum(axis=1) * R2
                I1[t, N + 1:] = B * att + direct + lam
                I1[t, N] = q[N] * F0 * np.exp(-tau[t] / mu0) + (wq * Qk[N] * e1).sum() * R2
    return I1


def source_function(c: Column, In_1: np.ndarray) -> np.ndarray:
    """Three-zone Jn, spec:314-323."""
    L = len(c.tau)
    Jn = np.zeros((L, 2 * c.N))
    Pa, Pr = c.P_atm[:, ::-1], c.P_aer[:, ::-1]
    for z in c.zones:
        for t in range(z.r0, z.r1 + 1):
            if z.kind == "mix":
                fa, fr = c.zone_fractions(z)
                Jn[t, :] = (c.alb_atm / 4) * _trapz(Pa * In_1[t, :], c.mu, axis=1) * fa \
                    + (z.alb_aer / 4) * _trapz(Pr * In_1[t, :], c.mu, axis=1) * fr
            else:
                Jn[t, :] = (c.alb_atm / 4) * _trapz(Pa * In_1[t, :], c.mu, axis=1)
    return Jn


def _transport(Jn, tau, mu, N, zones: Sequence[_Zone], tau_ref: Sequence[float], thick_test: float,
               surface: Optional[str], grd_alb: float, literal: bool) -> np.ndarray:
    """Downward then upward transport of one order.
    downward: spec:326-385 / I1_In:110-129;  upward: spec:388-449 / I1_In:86-108;
    Lambertian surface: lam:399/401."""
    L = len(tau)
    In = np.zeros((L, 2 * N))
    std = [m for m in range(N - 1) if not abs(mu[m]) < MU_THRESHOLD]
    small = [m for m in range(N - 1) if abs(mu[m]) < MU_THRESHOLD]
    std_a = np.array(std, dtype=int)
    with np.errstate(all="ignore"):
        for zi, z in enumerate(zones):
            b = z.r0 - 1 if zi > 0 else 0          # first row of the quadrature
            for t in range(z.r0, z.r1 + 1):
                for m in small:
                    In[t, m] = improved_asymptotic_downward_radiance(Jn[z.r0:t + 1, m], tau[z.r0:t + 1], tau[t], mu[m])
                if literal:
                    for m in std:
                        f = Jn[b:t + 1, m] * np.exp((tau[t] - tau[b:t + 1]) / mu[m])
                        v = -_trapz(f, tau[b:t + 1]) / mu[m]
                        if zi > 0:
                            v = In[b, m] * np.exp((tau[t] - tau[b]) / mu[m]) + v
                        In[t, m] = v
                elif len(std):
                    f = Jn[b:t + 1][:, std_a] * np.exp((tau[t] - tau[b:t + 1])[:, None] / mu[std_a])
                    v = -_trapz(f, tau[b:t + 1], axis=0) / mu[std_a]
                    if zi > 0:
                        v = In[b, std_a] * np.exp((tau[t] - tau[b]) / mu[std_a]) + v
                    In[t, std_a] = v
                _rewrite_small_down(In[t], mu, N, a4b_count(tau_ref[zi], N))
        mp = mu[N + 1:]
        rev = np.arange(N - 2, -1, -1)
        for zi in range(len(zones) - 1, -1, -1):
            z = zones[zi]
            bottom = zi == len(zones) - 1
            e = z.r1
            for t in range(z.r0, z.r1 + 1):
                if bottom:
                    if surface == "specular":
                        B = grd_alb * In[L - 1, rev]
                    elif surface == "lambertian":
                        B = -2 * grd_alb * _trapz(In[L - 1, rev] * mu[rev], mu[rev])
                    elif surface == "lambertian_readme":        # README.md:215: the same integral over ascending mu
                        B = 2 * grd_alb * _trapz(In[L - 1, rev] * mu[rev], mu[rev])
                    else:
                        B = 0.0
                    tb = tau[L - 1]
                else:
                    B = In[z.r1 + 1, N + 1:]
                    tb = tau[z.r1 + 1]
                ex = np.exp(-(tau[t:e + 1] - tau[t])[:, None] / mp)
                if thick_test >= 50:
                    q = _trapz(Jn[t:e + 1, N + 1:] * (ex / mp), tau[t:e + 1], axis=0)
                else:
                    q = _trapz(Jn[t:e + 1, N + 1:] * ex, tau[t:e + 1], axis=0) / mp
                if surface is None:
                    In[t, N + 1:] = q
                else:
                    In[t, N + 1:] = B * np.exp(-(tb - tau[t]) / mp) + q
                In[t, N] = Jn[t, N]
                _blend_small_up(In[t], mu, N)
    return In


def transport(c: Column, Jn: np.ndarray, literal: bool = True) -> np.ndarray:
    """One order of three-zone transport for a column (spec:326-449)."""
    tau = c.tau
    L = len(tau)
    tref = c.tau_ref()                                             # spec:342,361,380
    return _transport(Jn, tau, c.mu, c.N, c.zones, tref, thick_test=tau[L - 1] / c.mu[c.N + 1],
                      surface=c.surface, grd_alb=c.grd_alb, literal=literal)


def _py_max(a):
    """Python's builtin max over a sequence (first element wins ties / NaN poisons only if first)."""
    r = a[0]
    for x in a[1:]:
        if x > r:
            r = x
    return r


def convergence_ratio(In, I, N):
    """The loop test of spec:309."""
    L = In.shape[0]
    with np.errstate(all="ignore"):
        a = _py_max(list(In[0, N:] / I[0, N:]))
        b = _py_max(list(In[L - 1, :N] / I[L - 1, :N]))
    return b if b > a else a


@dataclass
class Solution:
    I: np.ndarray
    I_saved: np.ndarray
    n: int
    ratios: List[float] = field(default_factory=list)


def solve_column(c: Column, tol: float = 1e-4, max_orders: int = 10000, literal: bool = True,
                 I1: Optional[np.ndarray] = None) -> Solution:
    """The order loop of spec:301-458."""
    if I1 is None:
        I1 = first_order_lambertian_readme(c) if c.first_order == "readme" else first_order(c)
    In_1 = I1
    I = I1.copy()
    saved = [I1]
    In = np.ones_like(I1)
    n = 1
    ratios = []
    while True:
        r = convergence_ratio(In, I, c.N)
        ratios.append(r)
        if not (r >= tol) or n >= max_orders:
            break
        n += 1
        Jn = source_function(c, In_1)
        In = transport(c, Jn, literal=literal)
        In_1 = In
        I = I + In
        saved.append(In)
    return Solution(I=I, I_saved=np.stack(saved), n=n, ratios=ratios)


def solve_single_slab(tau, mu, tauStar, mu0, P0, P, alb, N, tol=1e-4, max_orders=10000, literal=True) -> Solution:
    """The caller-side loop the I1_In functions expect (same test as spec:309)."""
    I1 = I1_NumInt(tau, mu, tauStar, mu0, P0, alb, N)
    In_1, I, saved, In, n = I1, I1.copy(), [I1], np.ones_like(I1), 1
    while convergence_ratio(In, I, N) >= tol and n < max_orders:
        n += 1
        Jn = Jn_NumInt(n, In_1, tau, mu, tauStar, mu0, P, alb, N)
        In = In_NumInt(n, Jn, In_1, tau, mu, tauStar, mu0, P, alb, N, literal=literal)
        In_1 = In
        I = I + In
        saved.append(In)
    return Solution(I=I, I_saved=np.stack(saved), n=n)


# ---------------------------------------------------------------------------
# fluxes (graphe:157-158, crit:377-382)
# ---------------------------------------------------------------------------
def fluxes(I, mu, tau, N, mu0, grd_alb, beam_norm="crit"):
    """Downward / upward flux per level.  beam_norm='crit' uses F0/(4 pi) for the direct beam
    (crit:380-381, graphe:78-79), 'graphe' uses F0 (graphe:157-158)."""
    F0 = np.pi / mu0
    fb = F0 / (4 * np.pi) if beam_norm == "crit" else F0
    L = len(tau)
    fd = np.array([_trapz(I[i, :N] * mu[:N], mu[:N]) - fb * np.exp(-tau[i] / mu0) for i in range(L)])
    fu = np.array([_trapz(I[i, N:] * mu[N:], mu[N:]) + fb * grd_alb * np.exp(-(2 * tau[L - 1] - tau[i]) / mu0) for i in range(L)])
    return fd, fu


def diffusivity(I, mu):
    """graphe:10."""
    return np.array([-_trapz(I[i] * mu, mu) / _trapz(I[i], mu) for i in range(I.shape[0])])


def net_flux(I, mu, tau, N, mu0, grd_alb):
    """graphe:41: one trapezoid over the whole direction grid, beam terms with F0."""
    F0, L = np.pi / mu0, len(tau)
    return np.array([_trapz(I[i] * mu, mu) - F0 * np.exp(-tau[i] / mu0) + grd_alb * F0 * np.exp(-(2 * tau[L - 1] - tau[i]) / mu0)
                     for i in range(L)])


def heating_rate(I, mu, tau, N, mu0, grd_alb, z_profile, idx_up, idx_down, slabs=None):
    """graphe:70-91: -(1/(rho c_p)) d(flux_down + flux_up)/dz with the F0/(4 pi) beam terms, last level copied
    from the one above, and the two levels at the slab boundaries overwritten by their upper neighbours
    ('erase_pics', graphe:87-91)."""
    rho, c_p = 1.225, 1004
    fd, fu = fluxes(I, mu, tau, N, mu0, grd_alb, beam_norm="crit")
    flux = fd + fu
    L = len(tau)
    hr = np.zeros(L)
    for i in range(L - 1):
        hr[i] = -(1 / (rho * c_p)) * (flux[i + 1] - flux[i]) / (z_profile[i + 1] - z_profile[i])
    hr[-1] = hr[-2]
    for iu, idn in (slabs if slabs is not None else [(idx_up, idx_down)]):      # one pair per aerosol zone
        hr[iu - 1] = hr[iu - 2]
        hr[idn] = hr[idn - 1]
    return hr


def toa_net_flux(I, mu, tau, N, mu0, grd_alb):
    """crit:377-382."""
    fd, fu = fluxes(I, mu, tau, N, mu0, grd_alb, beam_norm="crit")
    return -fd[0] - fu[0]


def radiative_forcing(col: "Column", baseline: Optional["Column"] = None, literal=False):
    """crit:384-389.  As coded (`baseline=None`) the recursion for the aerosol-free term passes the same
    optical-depth grid, mixing fractions and phase arrays, so it recomputes the same column and the
    forcing is exactly 0.0.  With `baseline` (the column without the aerosol on its own grid)
    it is the difference of the two net fluxes."""
    f = toa_net_flux(solve_column(col, literal=literal).I, col.mu, col.tau, col.N, col.mu0, col.grd_alb)
    if baseline is None:
        return f - toa_net_flux(solve_column(col, literal=literal).I, col.mu, col.tau, col.N, col.mu0, col.grd_alb)
    fb = toa_net_flux(solve_column(baseline, literal=literal).I, baseline.mu, baseline.tau, baseline.N, baseline.mu0,
                      baseline.grd_alb)
    return f - fb


def critical_albedo(forcing_of_albedo, width=0.1, tol=0.001):
    """The bisection of crit:394-410 on the aerosol single-scattering albedo; `forcing_of_albedo(w)` plays
    SOS_Aer_radiative_forcing with alb_aer = w."""
    alb_max, alb_min = 1, 0
    while (alb_max - alb_min) > width:
        test = (alb_max + alb_min) / 2
        f = forcing_of_albedo(test)
        if abs(f) < tol:
            return test
        if f > 0:
            alb_min = test
        else:
            alb_max = test
    return (alb_max + alb_min) / 2


# ---------------------------------------------------------------------------
# phase functions (phase:68-195), vectorised
# ---------------------------------------------------------------------------
def _azimuth_average(fn, mu_a, mu_b, nb_phi=25):
    """(1/2pi) * trapz_phi [ p(cos Theta+) + p(cos Theta-) ] for all pairs (phase:116-128)."""
    phi = np.linspace(0, np.pi, nb_phi)
    cc = mu_a[:, None] * mu_b[None, :]
    ss = np.sqrt(1 - mu_b * mu_b)[None, :] * np.sqrt(1 - mu_a * mu_a)[:, None]
    cp = np.cos(0 - phi)
    pos = -(cc[..., None] + ss[..., None] * cp)
    neg = -(cc[..., None] - ss[..., None] * cp)
    return _trapz(fn(pos) + fn(neg), phi, axis=-1)


def _phase_pair(fn, N, mu, mu0):
    P0 = _azimuth_average(fn, mu, np.array([mu0]))[:, 0] / (4 * np.pi)
    P0 = P0 / _trapz(P0, mu) * 2                                   # phase:103
    S = _azimuth_average(fn, mu, mu) / (2 * np.pi)                  # symmetric raw matrix
    P = 4 * S / _trapz(S, mu, axis=0)[None, :]                      # column normalisation, phase:131
    return P0, P


def phase_isotropic(N, mu):
    """phase:68-76."""
    return np.ones(2 * N), 2 * np.ones((2 * N, 2 * N))


def phase_rayleigh(N, mu, mu0):
    """phase:79-133."""
    return _phase_pair(lambda c: (3 / 4) * (1 + c * c), N, mu, mu0)


def phase_hg(N, mu, mu0, g):
    """phase:141-195."""
    return _phase_pair(lambda c: (1 - g * g) / ((1 + g * g - 2 * g * c) ** 1.5), N, mu, mu0)


def interpolate_table(mu_tab, p_tab, c):
    """phase:198-236 (interpolate_fwc_phase), vectorised: clip to [-1, 1], `idx = searchsorted(mu_tab, c)`
    (left), table ends returned as they are, linear interpolation between idx-1 and idx otherwise."""
    c = np.clip(c, -1, 1)
    idx = np.searchsorted(mu_tab, c)
    lo = np.clip(idx - 1, 0, len(mu_tab) - 1)
    hi = np.clip(idx, 0, len(mu_tab) - 1)
    w = (c - mu_tab[lo]) / np.where(hi == lo, 1.0, mu_tab[hi] - mu_tab[lo])
    v = p_tab[lo] + w * (p_tab[hi] - p_tab[lo])
    return np.where(idx == 0, p_tab[0], np.where(idx >= len(mu_tab), p_tab[-1], v))


def phase_p0(kind, N, mu, mu0, g=0.0, table=None):
    """P0(mu, mu0) alone (phase:86-103) for kind 'rayleigh' | 'hg' | 'table' -- the per-column input of a mu0 sweep, without
    the O(D^2) matrix that `phase_rayleigh` & co. build beside it."""
    fn = {"rayleigh": lambda c: (3 / 4) * (1 + c * c),
          "hg": lambda c: (1 - g * g) / ((1 + g * g - 2 * g * c) ** 1.5),
          "table": (lambda c: interpolate_table(table[0], table[1], c)) if table is not None else None}[kind]
    P0 = _azimuth_average(fn, mu, np.array([mu0]))[:, 0] / (4 * np.pi)
    return P0 / _trapz(P0, mu) * 2


def phase_table(N, mu, mu0, mu_tab, p_tab):
    """phase:238-292 (fwc) for any tabulated phase function (fwc:3,173 is the reference's table)."""
    return _phase_pair(lambda c: interpolate_table(mu_tab, p_tab, c), N, mu, mu0)
