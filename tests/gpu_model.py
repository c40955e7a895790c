"""NumPy model of the *restructured* algorithm the HIP kernels implement
(test infrastructure, not product code).

The reference evaluates every radiance as an O(L) trapezoid sum; the kernels
use instead
  * Jn as a dense product  In_1 @ W  with the trapezoid weights folded into W,
  * transport as first-order linear recurrences along tau,
  * the quadratic / linear extrapolation next to mu=0- as a fixed linear map
    C[idx][n_src] of the source angles,
  * the upward mu->0+ blend evaluated per row with restarts from the blended
    boundary rows.
This file states that algorithm in NumPy, one row at a time, so that the
restructuring can be proven against the oracle on the CPU and so that a HIP
failure can be bisected (kernel bug vs. algorithm bug).
"""
import numpy as np

MU_THRESHOLD = 0.01
MU_VERY_SMALL = 0.001


def quad_weights(mu):
    d = np.diff(mu)
    w = np.zeros_like(mu)
    w[:-1] += d / 2
    w[1:] += d / 2
    return w


def fold_weights(P, mu):
    """W[k, m] = w_k * P[m, D-1-k]  so that  trapz(P[:, ::-1] * x, mu, axis=1) == x @ W."""
    w = quad_weights(mu)
    return (w[:, None] * P[:, ::-1].T).copy()


def a4b_count(tau_ref, N):
    c = 0.005 if tau_ref <= 0.0625 else 0.02 if tau_ref <= 1 else 0.04 if tau_ref < 4 else 0.06
    return int(c * N)


def a4b_table(mu, N, idx):
    """(s0, C) with rewritten lane N-1-i = sum_j C[i, j] * row[s0 + j]."""
    if idx == 0:
        return 0, np.zeros((0, 0))
    n = min(5, idx)
    x_all = mu[:N].astype(np.longdouble)
    if n < 2:
        s0, ns = N - idx - 2, 2
        xs = x_all[s0:s0 + 2]
        C = np.zeros((idx, 2), dtype=np.longdouble)
        for i in range(idx):
            xe = x_all[N - 1 - i]
            c3 = (xe - xs[1]) / (xs[0] - xs[1])
            C[i] = [c3, 1 - c3]
        return s0, C.astype(np.float64)
    s0, ns = N - idx - n, n
    xs = x_all[s0:s0 + n]
    C = np.zeros((idx, n), dtype=np.longdouble)
    if n == 2:
        for i in range(idx):
            xe = x_all[N - 1 - i]
            c1 = (xe - xs[0]) / (xs[1] - xs[0])
            C[i] = [1 - c1, c1]
        return s0, C.astype(np.float64)
    # degree-2 least squares, centred and scaled, in extended precision
    xb = xs.mean()
    h = xs[1] - xs[0]
    u = (xs - xb) / h
    V = np.stack([np.ones(n, dtype=np.longdouble), u, u * u], axis=1)
    G = V.T @ V
    Ginv = _inv3(G)
    M = Ginv @ V.T                      # coefficients = M @ y
    for i in range(idx):
        ue = (x_all[N - 1 - i] - xb) / h
        C[i] = np.array([1, ue, ue * ue], dtype=np.longdouble) @ M
    return s0, C.astype(np.float64)


def _inv3(G):
    a, b, c = G[0]; d, e, f = G[1]; g, h, i = G[2]
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    adj = np.array([[e * i - f * h, c * h - b * i, b * f - c * e],
                    [f * g - d * i, a * i - c * g, c * d - a * f],
                    [d * h - e * g, b * g - a * h, a * e - b * d]], dtype=np.longdouble)
    return adj / det


def small_mu_value(J, tau, zs, t, mu):
    """a4a for one (t, lane): zone-local slice [zs, t]."""
    if abs(mu) < MU_VERY_SMALL:
        slope = (J[t] - J[t - 1]) / (tau[t] - tau[t - 1]) if t > zs else 0.0
        return -J[t] + mu * slope
    lim = tau[t] - 5 * abs(mu)
    s = t
    while s - 1 >= zs and tau[s - 1] >= lim:
        s -= 1
    if not (tau[t] >= lim):
        return -J[t]
    f = J[s:t + 1] * np.exp((tau[t] - tau[s:t + 1]) / mu)
    if not np.all(np.isfinite(f)):
        return -J[t]
    acc = 0.0
    for q in range(s, t):
        acc += (tau[q + 1] - tau[q]) * (f[q + 1 - s] + f[q - s]) / 2
    return -acc / mu


TC = 8      # rows per chunk of the transport kernels


def special_chunks(L, zones):
    """Chunks (of TC rows, counted from the start of each sweep) that the chunk-parallel kernel evaluates in the serial
    form: those with a zone boundary and the last one of the sweep.  Returns (set for the downward sweep, set for the upward)."""
    nch = (L + TC - 1) // TC
    rows = [r1 for (_, r1) in zones[:-1]] + [r0 for (r0, _) in zones[1:]]
    return ({nch - 1} | {t // TC for t in rows}), ({nch - 1} | {(L - 1 - t) // TC for t in rows})


def transport_model(Jn, tau, mu, N, zones, nfix, surface, rho, chunk_local=False):
    """zones: list of (r0, r1); nfix: rewritten-angle count per zone;
    surface: None | 'specular' | 'lambertian'.  Returns (In, status).
    chunk_local: the arithmetic of transport_scan.hip -- inside a chunk without a zone boundary the recurrence
    S_t = E_t S_{t-1} + c_t is evaluated as d_u = E_u d_{u-1} + c_u, p_u = E_u p_{u-1} from d = 0, p = 1 and
    S_u = p_u S_in + d_u, with S_in the value carried out of the previous chunk."""
    L = len(tau)
    D = 2 * N
    sp_dn, sp_up = special_chunks(L, zones) if chunk_local else (set(), set())
    In = np.zeros((L, D))
    tabs = [a4b_table(mu, N, k) for k in nfix]
    lane_small = np.abs(mu[:N]) < MU_THRESHOLD
    lane_small[N - 1] = False
    with np.errstate(all="ignore"):
        # ---- downward ----
        Dv = np.zeros(N)                 # recurrence state, lanes 0..N-2
        md = mu[:N].copy()
        md[N - 1] = -1.0                 # placeholder, lane N-1 is not transported
        zone_of = np.zeros(L, dtype=int)
        for zi, (r0, r1) in enumerate(zones):
            zone_of[r0:r1 + 1] = zi
        for t in range(L):
            zi = zone_of[t]
            r0, r1 = zones[zi]
            dl = tau[t] - tau[t - 1] if t > 0 else 0.0
            E = np.exp(dl / md)
            c = -(dl / 2) * (Jn[t - 1, :N] * E + Jn[t, :N]) / md if t > 0 else np.zeros(N)
            if chunk_local and (t // TC) not in sp_dn:
                if t % TC == 0:
                    S_in, d_loc, p_loc = Dv.copy(), np.zeros(N), np.ones(N)
                d_loc = d_loc * E + c
                p_loc = p_loc * E
                Dv = p_loc * S_in + d_loc
            else:
                Dv = Dv * E + c
            row = Dv.copy()
            row[N - 1] = 0.0
            for m in np.nonzero(lane_small)[0]:
                row[m] = small_mu_value(Jn[:, m], tau, r0, t, mu[m])
            s0, C = tabs[zi]
            k = nfix[zi]
            if k:
                src = row[s0:s0 + C.shape[1]]
                for i in range(k):
                    row[N - 1 - i] = C[i] @ src
            In[t, :N] = row
            if t == r1:                  # restart of the next zone from the final row
                Dv = row.copy()
        # ---- surface ----
        sfc = In[L - 1, :N]
        mp = mu[N:].copy()
        if surface == "specular":
            B = rho * sfc[::-1]          # lane j (m=N+j) <- down lane N-1-j
        elif surface == "lambertian":
            rev = np.arange(N - 2, -1, -1)
            y = sfc[rev] * mu[rev]; x = mu[rev]
            S = np.sum(np.diff(x) * (y[1:] + y[:-1]) / 2)
            B = np.full(N, -2 * rho * S)
        else:
            B = np.zeros(N)
        # ---- upward ----
        U = B.copy()
        status = 0
        mp[0] = 1.0
        for t in range(L - 1, -1, -1):
            zi = zone_of[t]
            r0, r1 = zones[zi]
            dl = tau[t + 1] - tau[t] if t < L - 1 else 0.0
            E = np.exp(-dl / mp)
            # first row of a zone above: gap, no source integral
            c = (dl / 2) * (Jn[t, N:] + Jn[t + 1, N:] * E) / mp if (t < L - 1 and t != r1) else np.zeros(N)
            if chunk_local and ((L - 1 - t) // TC) not in sp_up:
                if (L - 1 - t) % TC == 0:
                    S_in, d_loc, p_loc = U.copy(), np.zeros(N), np.ones(N)
                d_loc = d_loc * E + c
                p_loc = p_loc * E
                U = p_loc * S_in + d_loc
            else:
                U = U * E + c
            r = U.copy()
            r[0] = Jn[t, N]
            # blend
            ks = None
            for j in range(1, N - 2):
                if not (abs((r[j] - r[j + 1]) - (r[j + 1] - r[j + 2])) > 1e-4):
                    ks = j
                    break
            if ks is None:
                return In, 1             # IndexError in the reference
            kf = ks + 1
            out = r.copy()
            for j in range(1, kf):
                w = mp[j] / mp[kf]
                out[j] = (1 - w) * r[0] + w * r[kf]
            In[t, N:] = out
            if t == r0:
                U = out.copy()
    return In, status


def source_model(In_1, Wa, Wr, ca, cr, symmetric=False):
    """ca, cr: per-row coefficients.  symmetric: the flip-symmetric form of the kernels (jn_gemm.hip, SYM)."""
    if symmetric:
        return ca[:, None] * source_symmetric(In_1, Wa) + cr[:, None] * source_symmetric(In_1, Wr)
    return ca[:, None] * (In_1 @ Wa) + cr[:, None] * (In_1 @ Wr)


def asymmetry(W):
    """max |W[k, m] - W[D-1-k, D-1-m]| / max |W|: what sosrt_set_phase measures (0 for an exactly flip-symmetric matrix;
    1e-14 for the reference's phase-matrix builders on its direction grid)."""
    return float(np.max(np.abs(W - W[::-1, ::-1])) / np.max(np.abs(W)))


def symfold(W):
    """(S, A), N x N each: S = (P + Q) / 2, A = (P - Q) / 2 with P[k, m] = (W[k, m] + W[k', m']) / 2 and
    Q[k, m] = (W[k', m] + W[k, m']) / 2, k' = D-1-k, m' = D-1-m (k_symfold)."""
    N = W.shape[0] // 2
    F = W[::-1, ::-1]
    p = W[:N, :N] + F[:N, :N]                  # W[k, m] + W[k', m']
    q = W[::-1][:N, :N] + W[:, ::-1][:N, :N]   # W[k', m] + W[k, m']
    return 0.25 * (p + q), 0.25 * (p - q)


def source_symmetric(In_1, W):
    """In_1 @ W for a flip-symmetric W as two N x N products: with a = In_1[:, :N], b = In_1[:, :N-1:-1] (mirrored),
    X = (a + b) @ S, Y = (a - b) @ A:  Jn[:, m] = X + Y, Jn[:, D-1-m] = X - Y."""
    N = W.shape[0] // 2
    S, A = symfold(W)
    a, b = In_1[:, :N], In_1[:, ::-1][:, :N]
    X, Y = (a + b) @ S, (a - b) @ A
    out = np.empty_like(In_1)
    out[:, :N] = X + Y
    out[:, ::-1][:, :N] = X - Y
    return out
