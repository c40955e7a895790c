"""mu -> 0 helpers with the reference's names and signatures (SOS_Aer_In_limit.py).

In the reference these are scalar Python functions called once per (layer, angle) by the
drivers.  In this build they are fused into the transport kernel; the functions here exist
for drop-in callers and evaluate the same device code through the C ABI on one row.
"""
import numpy as np

from .global_va import MU_THRESHOLD, MU_EXTREME_THRESHOLD, MU_VERY_SMALL_THRESHOLD  # noqa: F401
from .solver import Solver

_cache = {}


def _solver(nb_angles, mu_down):
    s = _cache.get(nb_angles)
    mu = np.concatenate((np.asarray(mu_down, dtype=np.float64), -np.asarray(mu_down, dtype=np.float64)[::-1]))
    if s is None:
        s = Solver(2, nb_angles, max_batch=4, max_orders=1)
        _cache[nb_angles] = s
    if not s.same_grid(mu):
        s.set_grid(mu)
    return s


def improved_limit_mu_down(In_down, mu_down, nb_angles, idx, i):
    """SOS_Aer_In_limit.py:113-141."""
    s = _solver(nb_angles, mu_down)
    return float(s.limit_mu_down(np.asarray(In_down, dtype=np.float64)[None, :nb_angles], idx)[0, i])


def limit_mu_down(In_down, mu_down, nb_angles, idx, i):
    """SOS_Aer_In_limit.py:155-158 (the legacy straight line; two-term arithmetic, no kernel)."""
    slope = (In_down[-idx - 2] - In_down[-idx - 1]) / (mu_down[-idx - 2] - mu_down[-idx - 1])
    return slope * (mu_down[-i - 1] - mu_down[-idx - 1]) + In_down[-idx - 1]


def improved_asymptotic_downward_radiance(Jn_slice, tau_slice, tau_t, mu):
    """SOS_Aer_In_limit.py:70-109."""
    n = len(tau_slice)
    if n == 0:
        return 0.0
    s = _cache.get("asym")
    if s is None:
        s = _cache["asym"] = Solver(64, 8, max_batch=64, max_orders=1)
    stride = max(n, 1)
    if 2 * stride + 4 > s.max_batch * s.L * s.D:
        s.close()
        s = _cache["asym"] = Solver(64, 8, max_batch=(2 * stride + 4) // (64 * 16) + 1, max_orders=1)
    J = np.asarray(Jn_slice, dtype=np.float64).reshape(1, n)
    t = np.asarray(tau_slice, dtype=np.float64).reshape(1, n)
    return float(s.asymptotic_down(J, t, [n], [tau_t], [mu])[0])


def mu_approx_In(mu, nb_angles):
    """SOS_Aer_In_limit.py:145-153 (pure index search; its result is unused by the transport)."""
    idx = nb_angles
    while mu[idx] < 0.009:
        idx += 1
    mu_1 = idx
    while mu[idx] < 0.020:
        idx += 1
    return mu_1, idx
