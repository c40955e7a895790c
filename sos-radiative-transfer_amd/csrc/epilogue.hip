// Kernels either side of the order loop, gfx950:
//
//   k_epilogue      what the reference's callers consume from a converged field, computed where the field
//                   lives: downward / upward flux per level (graphe:157-158, crit:380-381), diffusivity
//                   (graphe:10), heating rate incl. the 'erase_pics' fix-up (graphe:74-91; at every aerosol
//                   zone of the column's zone table) and the net flux
//                   at the top of the atmosphere (crit:382).  One workgroup per column, one wavefront per
//                   layer row at a time (a row of 2N doubles is read once, coalesced); the per-level net
//                   flux stays in LDS for the finite differences of the heating rate.
//   k_phase_p0      azimuth-averaged first-order phase function P0(mu, mu0) of every column of a sweep
//                   (phase:86-103, 148-165, 245-262): a mu0 sweep needs a fresh P0 per column.
//   k_phase_matrix  P(mu, mu') with the reference's column normalisation (phase:107-131, 169-193, 266-290).
//
// Phase-function kinds: isotropic (phase:68), Rayleigh (phase:79), Henyey-Greenstein (phase:141) and a
// tabulated function with the reference's linear interpolation (phase:198-236; fwc:3,173 is its table).
#include "kernels.hpp"

#include "../../include/sosrt.h"

namespace sosrt {

namespace {

#define SOSRT_PI 3.14159265358979323846

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the workgroup; every thread gets the result
__device__ double bsum(double x, double* s_red) {
    const int tid = threadIdx.x, nw = blockDim.x >> 6;
    const double v = wsum(x);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int i = 0; i < nw; ++i) r += s_red[i];
    return r;
}

__global__ __launch_bounds__(256) void k_epilogue(Grid g, const double* __restrict__ w_all, int B,
                                                  const double* __restrict__ tau_all, const double* __restrict__ I_all,
                                                  const ColDesc* __restrict__ desc, int beam_norm,
                                                  const double* __restrict__ z_profile, EpilogueOut out) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int L = g.L, N = g.N, D = g.D;
    extern __shared__ double s_flux[];                       // [2 L]: flux_down + flux_up with the F0/(4 pi) beam terms; heating rate
    const ColDesc& d = desc[b];
    const double* tau = tau_all + (size_t)b * L;
    const double mu0 = d.mu0, rho = d.rho;
    const double F0 = SOSRT_PI / mu0;                        // spec:105
    const double f4 = F0 / (4 * SOSRT_PI);
    const double fb = beam_norm ? F0 : f4;
    const double tau_last = tau[L - 1];
    for (int t = wave; t < L; t += nw) {
        const double* I = I_all + ((size_t)b * L + t) * D;
        double sd = 0, su = 0, den = 0;
        for (int k = lane; k < N; k += 64) {
            const double a = I[k], c = I[N + k];
            sd += g.wflux_dn[k] * a;                         // trapz(I[:N] mu[:N], mu[:N])
            su += g.wflux_up[k] * c;
            den += w_all[k] * a + w_all[N + k] * c;          // trapz(I, mu) over the whole grid (graphe:10)
        }
        sd = wsum(sd); su = wsum(su); den = wsum(den);
        if (lane == 0) {
            const double e_dn = exp(-tau[t] / mu0), e_up = exp(-(2 * tau_last - tau[t]) / mu0);
            const size_t o = (size_t)b * L + t;
            if (out.flux_down) out.flux_down[o] = sd - fb * e_dn;
            if (out.flux_up) out.flux_up[o] = su + fb * rho * e_up;
            // the zero-width interval between the two mu = 0 nodes adds nothing: trapz(I mu, mu) = sd + su
            if (out.diffusivity) out.diffusivity[o] = -(sd + su) / den;
            const double fd4 = sd - f4 * e_dn, fu4 = su + f4 * rho * e_up;
            s_flux[t] = fd4 + fu4;
            if (t == 0 && out.net_toa) out.net_toa[b] = -fd4 - fu4;          // crit:382
        }
    }
    __syncthreads();
    if (out.heating_rate && z_profile) {
        const double k = -(1.0 / (1.225 * 1004));            // graphe:71-72: -(1 / (rho c_p))
        double* s_hr = s_flux + L;                           // [L]
        // graphe:83-85: forward difference, the last level copies the one above
        for (int t = tid; t < L; t += blockDim.x) {
            const int s = t == L - 1 ? L - 2 : t;
            s_hr[t] = k * (s_flux[s + 1] - s_flux[s]) / (z_profile[s + 1] - z_profile[s]);
        }
        __syncthreads();
        if (tid == 0 && d.nz >= 3) {
            // graphe:87-91 'erase_pics', per aerosol zone, in the reference's statement order (Python indexing: -1 is the
            // last level): the level above a slab and the slab's last level take the value of their upper neighbour
            for (int z = 1; z < d.nz; ++z) {
                if (!d.mix[z]) continue;
                const int iu = d.r0[z], id = d.r1[z];
                auto at = [&](int i) { return i < 0 ? i + L : i; };
                s_hr[at(iu - 1)] = s_hr[at(iu - 2)];
                s_hr[at(id)] = s_hr[at(id - 1)];
            }
        }
        __syncthreads();
        double* hr = out.heating_rate + (size_t)b * L;
        for (int t = tid; t < L; t += blockDim.x) hr[t] = s_hr[t];
    }
}

// ---------------------------------------------------------------------------------------------
// phase functions
// ---------------------------------------------------------------------------------------------
struct PhaseFn {
    int kind;
    double g;
    const double* tab_mu;
    const double* tab_p;
    int ntab;
    // p(cos Theta)
    __device__ __forceinline__ double operator()(double c) const {
        if (kind == SOSRT_PHASE_RAYLEIGH) return (3.0 / 4) * (1 + c * c);                    // phase:96
        if (kind == SOSRT_PHASE_HG) {                                                         // phase:158
            const double x = 1 + g * g - 2 * g * c;
            return (1 - g * g) / (x * sqrt(x));
        }
        if (kind == SOSRT_PHASE_TABLE) {                                                      // phase:198-236
            c = fmin(fmax(c, -1.0), 1.0);
            int lo = 0, hi = ntab;                                                            // searchsorted, side='left'
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (tab_mu[mid] < c) lo = mid + 1;
                else hi = mid;
            }
            if (lo == 0) return tab_p[0];
            if (lo >= ntab) return tab_p[ntab - 1];
            const double ml = tab_mu[lo - 1], mh = tab_mu[lo], pl = tab_p[lo - 1], ph = tab_p[lo];
            return pl + (c - ml) / (mh - ml) * (ph - pl);
        }
        return 1.0;
    }
};

// trapz over phi = linspace(0, pi, nphi) of p(cos Theta+) + p(cos Theta-), cos Theta+- = -(a b +- sa sb cos phi)
__device__ __forceinline__ double ring(const PhaseFn& p, double cc, double ss, const double* __restrict__ cosphi,
                                       const double* __restrict__ wphi, int nphi) {
    double acc = 0;
    for (int q = 0; q < nphi; ++q) {
        const double x = ss * cosphi[q];
        acc += wphi[q] * (p(-(cc + x)) + p(-(cc - x)));
    }
    return acc;
}

__global__ __launch_bounds__(256) void k_phase_p0(Grid g, const double* __restrict__ w_all, int B, PhaseFn p,
                                                  const double* __restrict__ cosphi, const double* __restrict__ wphi,
                                                  int nphi, const double* __restrict__ mu0_all,
                                                  double* __restrict__ P0_all) {
    const int b = blockIdx.x, tid = threadIdx.x, D = g.D;
    __shared__ double s_red[8];
    const double mu0 = mu0_all[b];
    const double s0 = sqrt(1 - mu0 * mu0);
    double* P0 = P0_all + (size_t)b * D;
    if (p.kind == SOSRT_PHASE_ISO) {                                  // phase:68-76: ones, no normalisation
        for (int m = tid; m < D; m += blockDim.x) P0[m] = 1.0;
        return;
    }
    double part = 0;
    for (int m0 = 0; m0 < D; m0 += blockDim.x) {
        const int m = m0 + tid;
        if (m < D) {
            const double mu = g.mu[m];
            const double v = ring(p, mu * mu0, s0 * sqrt(1 - mu * mu), cosphi, wphi, nphi) / (4 * SOSRT_PI);   // phase:101
            P0[m] = v;
            part += w_all[m] * v;
        }
    }
    const double norm = bsum(part, s_red);                            // trapz(P0, mu), phase:103
    for (int m = tid; m < D; m += blockDim.x) P0[m] = P0[m] / norm * 2;
}

__global__ __launch_bounds__(256) void k_phase_matrix(Grid g, const double* __restrict__ w_all, PhaseFn p,
                                                      const double* __restrict__ cosphi, const double* __restrict__ wphi,
                                                      int nphi, double* __restrict__ P) {
    const int n = blockIdx.x, tid = threadIdx.x, D = g.D;             // column n: incidence direction mu[n]
    __shared__ double s_red[8];
    if (p.kind == SOSRT_PHASE_ISO) {                                  // phase:74: 2 everywhere, no normalisation
        for (int m = tid; m < D; m += blockDim.x) P[(size_t)m * D + n] = 2.0;
        return;
    }
    const double mun = g.mu[n], sn = sqrt(1 - mun * mun);
    double part = 0;
    for (int m0 = 0; m0 < D; m0 += blockDim.x) {
        const int m = m0 + tid;
        if (m < D) {
            const double mu = g.mu[m];
            const double v = ring(p, mu * mun, sn * sqrt(1 - mu * mu), cosphi, wphi, nphi) / (2 * SOSRT_PI);   // phase:128
            P[(size_t)m * D + n] = v;
            part += w_all[m] * v;
        }
    }
    const double norm = bsum(part, s_red);                            // trapz(P[:, n], mu), phase:131
    for (int m = tid; m < D; m += blockDim.x) P[(size_t)m * D + n] = 4 * P[(size_t)m * D + n] / norm;
}

}  // namespace

void launch_epilogue(hipStream_t s, const Grid& g, const double* w, int B, const double* tau, const double* I,
                     const ColDesc* desc, int beam_norm, const double* z_profile, const EpilogueOut& out) {
    hipLaunchKernelGGL(k_epilogue, dim3(B), dim3(256), (size_t)2 * g.L * sizeof(double), s, g, w, B, tau, I, desc, beam_norm,
                       z_profile, out);
}

void launch_phase_p0(hipStream_t s, const Grid& g, const double* w, int B, int kind, double gpar, const double* tab_mu,
                     const double* tab_p, int ntab, const double* cosphi, const double* wphi, int nphi,
                     const double* mu0, double* P0) {
    PhaseFn p{kind, gpar, tab_mu, tab_p, ntab};
    hipLaunchKernelGGL(k_phase_p0, dim3(B), dim3(256), 0, s, g, w, B, p, cosphi, wphi, nphi, mu0, P0);
}

void launch_phase_matrix(hipStream_t s, const Grid& g, const double* w, int kind, double gpar, const double* tab_mu,
                         const double* tab_p, int ntab, const double* cosphi, const double* wphi, int nphi, double* P) {
    PhaseFn p{kind, gpar, tab_mu, tab_p, ntab};
    hipLaunchKernelGGL(k_phase_matrix, dim3(g.D), dim3(256), 0, s, g, w, p, cosphi, wphi, nphi, P);
}

}  // namespace sosrt
