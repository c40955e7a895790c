// HIP kernels of the SOS hot path for gfx950 (CDNA4, wave64).
//
//   k_prepare      per-column zone table + Jn row coefficients            (spec:40-53, 342-345)
//   k_first_order  closed-form first order I1, three-zone or single slab  (spec:104-292, I1_In:13-58)
//   k_jn_gemm      Jn = diag(c) In_1 W   as an FP64-MFMA contraction        (spec:314-323, I1_In:62-74)
//   k_smallmu      windowed / Taylor value of the |mu| < 0.01 lanes        (In_limit:70-109)
//   k_transport    per-order layer recurrences, mu->0 treatments, surface,
//                  accumulation and the convergence test                   (spec:326-458, I1_In:77-130)
//   k_fluxes       flux epilogue                                           (graphe:157-158, crit:377-382)
//
// Layout: every radiance field is [column][layer t][direction m], m fastest, so that one
// wavefront reads 64 consecutive directions of one layer (512 B, coalesced).
#include <cstdlib>

#include "kernels.hpp"

#include "../../include/sosrt.h"

namespace sosrt {


typedef double f64x4 __attribute__((ext_vector_type(4)));

#define SOSRT_PI 3.14159265358979323846

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_fmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));   // fmax drops NaN operands
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Python's builtin max() over x[0..n) as the reference's loop test applies it to a NumPy row
// (spec:309): the first element seeds the running maximum and NaN never wins a comparison, so
// the result is NaN iff x[0] is NaN, else the largest non-NaN element.  `x` is this thread's
// element (thread i holds x[i]), `valid` marks i < n.  Every thread gets the result.
__device__ double block_pymax(double x, bool valid, double* s_red, int first_tid = 0) {
    const int tid = threadIdx.x, nw = blockDim.x >> 6;
    double v = wave_fmax(valid ? x : __builtin_nan(""));
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    if (tid == first_tid) s_red[nw] = x;
    __syncthreads();
    double r = s_red[0];
    for (int i = 1; i < nw; ++i) r = fmax(r, s_red[i]);
    const double first = s_red[nw];
    return (first != first) ? first : r;
}

__device__ double block_sum(double x, double* s_red) {
    const int tid = threadIdx.x, nw = blockDim.x >> 6;
    double v = wave_sum(x);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int i = 0; i < nw; ++i) r += s_red[i];
    return r;
}

// the loop test of spec:309 given max-over-row results a (TOA, upward) and b (surface, downward)
__device__ __forceinline__ double outer_pymax(double a, double b) { return (b > a) ? b : a; }

__device__ __forceinline__ int d_fix_bucket(double tau_ref) {
    if (tau_ref <= 0.0625) return 0;
    if (tau_ref <= 1) return 1;
    if (tau_ref < 4) return 2;
    return 3;
}

// ------------------------------------------------------------------------------------------
// k_prepare: zone table per column
// ------------------------------------------------------------------------------------------
__global__ void k_prepare(Grid g, int B, int geom, int surface, ColScalars sc, const double* __restrict__ tau,
                          ColDesc* __restrict__ desc, double* __restrict__ rowcoef_a,
                          double* __restrict__ rowcoef_r, int* __restrict__ need_small, SolveSetup su) {
    const int b = blockIdx.x;
    if (b >= B) return;
    // The start of a solve (su: all null elsewhere) rides on this launch instead of three more: the counters of the NEXT solve are
    // zeroed (this solve's were zeroed by the previous one, or at sosrt_create), the column's redo flag is cleared, and the
    // second wave hashes the column's optical-depth profile (the tau groups of k_tau_rep).
    if (su.zero_next && b == 0 && threadIdx.x < su.n_zero) su.zero_next[threadIdx.x] = 0;
    if (su.redo && threadIdx.x == 0) su.redo[b] = 0;
    if (su.scan_sync && threadIdx.x < 2) su.scan_sync[2 * b + threadIdx.x] = 0;     // {arrivals, flags}: a solve that ended in an error may have left them set
    if (su.hash && threadIdx.x >= 64) {
        const int lane = threadIdx.x - 64;
        const unsigned long long* t = reinterpret_cast<const unsigned long long*>(tau + (size_t)b * g.L);
        unsigned long long h = 0;
        for (int i = lane; i < g.L; i += 64) {
            unsigned long long x = t[i] + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
            x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
            h += x;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
        if (lane == 0) su.hash[b] = h;
    }
    __shared__ ColDesc d;
    const double* tb = tau + (size_t)b * g.L;
    if (threadIdx.x == 0) {
        const int L = g.L;
        d.surface = surface;
        d.geom = geom;
        d.mu0 = sc.mu0[b];
        d.T = sc.T[b];
        d.wa = sc.alb_atm[b];
        double tref[kMaxZones];
        if (geom == SOSRT_GEOM_THREE_ZONE) {
            // The zone table: the reference's three zones (spec:113-449 writes every formula three times) or any
            // alternation of clear and aerosol zones.  Extrapolation bucket of a zone (I1_In:124, spec:342,361,380): the
            // optical depth at the zone's own last row for the top zone and for a slab, at the last row of the slab
            // above for a clear zone below one.
            const int nz = sc.nz[b];
            d.nz = nz;
            d.rho = sc.rho[b];
            d.tau_bottom = tb[L - 1];
            const double da = sc.dtau_atm[b];
            for (int z = 0; z < kMaxZones; ++z) {
                if (z < nz) {
                    d.r0[z] = sc.zr0[b * kMaxZones + z];
                    d.r1[z] = z + 1 < nz ? sc.zr0[b * kMaxZones + z + 1] - 1 : L - 1;
                    d.mix[z] = sc.zmix[b * kMaxZones + z];
                    const double dr = d.mix[z] ? sc.zdtr[b * kMaxZones + z] : 0.0;
                    d.wr[z] = d.mix[z] ? sc.zwr[b * kMaxZones + z] : 0.0;
                    d.fa[z] = da / (da + dr);
                    d.fr[z] = dr / (da + dr);
                    d.ca[z] = d.mix[z] ? (d.wa / 4) * d.fa[z] : d.wa / 4;            // spec:321,323
                    d.cr[z] = d.mix[z] ? (d.wr[z] / 4) * d.fr[z] : 0.0;
                    tref[z] = (z == 0 || d.mix[z]) ? tb[d.r1[z]] : tb[d.r0[z] - 1];
                } else {
                    d.r0[z] = L; d.r1[z] = L - 1; d.mix[z] = 0; d.wr[z] = 0; d.fa[z] = 1; d.fr[z] = 0; d.ca[z] = 0; d.cr[z] = 0;
                    tref[z] = 0;
                }
            }
        } else {
            d.nz = 1;
            d.r0[0] = 0; d.r1[0] = L - 1; d.mix[0] = 0;
            for (int z = 0; z < kMaxZones; ++z) {
                if (z) { d.r0[z] = L; d.r1[z] = L - 1; d.mix[z] = 0; d.ca[z] = 0; tref[z] = 0; }
                d.cr[z] = 0; d.wr[z] = 0; d.fa[z] = 1; d.fr[z] = 0;
            }
            tref[0] = d.T;                                               // I1_In:124 uses tauStar
            d.rho = 0;
            d.tau_bottom = d.T;                                          // I1_In:54 uses tauStar, not tau[-1]
            d.ca[0] = d.wa / 4;
        }
        for (int z = 0; z < kMaxZones; ++z) {
            const int k = d_fix_bucket(tref[z]);
            d.fixtab[z] = k;
            d.nfix[z] = (z < d.nz) ? g.fix[k].idx : 0;
            // a |mu| < 0.01 lane below the rewritten ones keeps its k_smallmu value: the launch is needed
            if (need_small && z < d.nz && g.nsmall > 0 && g.small_lanes[0] < g.N - d.nfix[z]) atomicOr(need_small, 1);
        }
        desc[b] = d;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < g.L; t += blockDim.x) {
        int z = 0;
        while (z + 1 < d.nz && t > d.r1[z]) ++z;
        rowcoef_a[(size_t)b * g.L + t] = d.ca[z];
        rowcoef_r[(size_t)b * g.L + t] = d.cr[z];
    }
}

void launch_prepare(hipStream_t s, const Grid& g, int B, int geom, int surface, ColScalars sc, const double* tau,
                    ColDesc* desc, double* rowcoef_a, double* rowcoef_r, int* need_small, SolveSetup su) {
    hipLaunchKernelGGL(k_prepare, dim3(B), dim3(128), 0, s, g, B, geom, surface, sc, tau, desc, rowcoef_a, rowcoef_r,
                       need_small, su);
}

// ------------------------------------------------------------------------------------------
// k_first_order
// One workgroup per column, thread j owns the downward direction m = j and then the upward
// direction m = N + j; the layer loop is sequential only because each zone starts from the
// attenuated last row of the previous one ("scatt_before", spec:147,176,240,270).
// ------------------------------------------------------------------------------------------
constexpr double kFactorMax = 1e150;    // largest zone constant the one-exponential factorisation of k_first_order is used with
__global__ void k_first_order(Grid g, const double* __restrict__ tau_all, const double* __restrict__ P0a_all,
                              const double* __restrict__ P0r_all, const ColDesc* __restrict__ desc,
                              double* __restrict__ I1_all, double* __restrict__ I_all, double* __restrict__ saved,
                              size_t saved_col_stride, Conv cv, int do_conv) {
    // blockIdx.y = 0: the downward half of the field, 1: the upward half and the loop test.  The upward
    // half needs the downward radiance at the surface (spec:211); it evaluates that one row itself (the
    // same closed form, one row per zone) instead of waiting for the other workgroup.
    // blockIdx.z = row block: the rows of a zone are independent closed forms once the zone's boundary row is known, and
    // that row is itself a closed form (one row per zone above it), so a workgroup takes the rows t = blockIdx.z mod gridDim.z
    // and evaluates the boundary rows for itself.  A lone column is then spread over 2 x gridDim.z workgroups instead of two
    // (its fp64 exponentials are a latency chain: 94 us for one column with two), and a full batch keeps more waves per SIMD
    // to hide them behind.
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool up_half = blockIdx.y == 1;
    const int blk = blockIdx.z, nblk = gridDim.z;
    const int L = g.L, N = g.N, D = g.D;
    extern __shared__ double sm[];
    double* s_tau = sm;                 // [L]
    double* s_sfc = s_tau + L;          // [blockDim]
    double* s_red = s_sfc + blockDim.x; // [nw + 1]
    __shared__ ColDesc d;                // dynamically indexed: keep it out of scratch
    if (tid == 0) d = desc[b];
    const double* tau = tau_all + (size_t)b * L;
    const double* P0a = P0a_all + (size_t)b * D;
    const double* P0r = P0r_all ? P0r_all + (size_t)b * D : P0a;
    double* I1 = I1_all + (size_t)b * L * D;
    double* Iacc = I_all ? I_all + (size_t)b * L * D : nullptr;
    double* sv = saved ? saved + (size_t)b * saved_col_stride : nullptr;
    for (int t = tid; t < L; t += blockDim.x) s_tau[t] = tau[t];
    __syncthreads();

    const double mu0 = d.mu0, T = d.T, rho = d.rho;
    const double F0 = SOSRT_PI / mu0;                       // spec:105
    const double R = F0 * rho * exp(-T / mu0);              // reflected beam at the surface
    const double c4pi = 1.0 / (4 * SOSRT_PI);
    const bool valid = tid < N;
    double rdn = 0, rup = 0;                                // ratios of the first loop test (In = ones)
    // the two beam attenuations of a layer do not depend on the direction: once per row, in LDS
    double* s_e0 = s_red + (blockDim.x >> 6) + 2;           // [L] exp(-tau/mu0)
    double* s_eT = s_e0 + L;                                // [L] exp(-(T - tau)/mu0)
    for (int t = tid; t < L; t += blockDim.x) {
        s_e0[t] = exp(-s_tau[t] / mu0);
        s_eT[t] = exp(-(T - s_tau[t]) / mu0);
    }
    __syncthreads();
    constexpr int FU = 4;    // rows of a zone are independent: FU of them are evaluated together
    // first row >= r of this block (rows t with t mod nblk == blk)
    auto first_from = [&](int r) { return r + ((blk - r % nblk) + nblk) % nblk; };
    auto last_upto = [&](int r) { return r - ((r % nblk - blk) + nblk) % nblk; };

    // ---- downward, m = tid ----
    {
        const int m = tid;
        const int mm = valid ? m : 0, mir = 2 * N - 1 - mm;
        const double mu = g.mu[mm];
        const double qa = d.wa * P0a[mm] * c4pi, qam = d.wa * P0a[mir] * c4pi;
        const double pa = P0a[mm], pam = P0a[mir], pr = P0r[mm], prm = P0r[mir];
        const bool near = fabs(mu + mu0) < 0.0001;          // spec:111
        const bool node = m > N - 2;                         // the mu = 0- node (spec:128-131)
        const double gd = mu0 / (mu0 + mu), gs = mu0 / (mu0 - mu);
        const double rmu = 1.0 / mu;
        double Ib = 0, vlast = 0;
        for (int z = 0; z < d.nz; ++z) {
            const double qx = (d.wa * pa * d.fa[z] + d.wr[z] * pr * d.fr[z]) * c4pi;       // spec:149
            const double qxm = (d.wa * pam * d.fa[z] + d.wr[z] * prm * d.fr[z]) * c4pi;
            const double q = d.mix[z] ? qx : qa, qm = d.mix[z] ? qxm : qam;
            const double t_bd = z ? s_tau[d.r0[z] - 1] : 0.0;
            const double t_bs = z ? s_tau[d.r0[z]] : 0.0;
            const double e_bd = exp(-t_bd / mu0), e_bs = exp(-(T - t_bs) / mu0);
            const int r1 = d.r1[z];
            // The two attenuations of a row differ by a factor that depends on the zone and the direction only:
            // e^{(tt - t_bs)/mu} = e^{(tt - t_bd)/mu} e^{(t_bd - t_bs)/mu}.  One fp64 exponential per element instead of two (the
            // kernel is bound by them); the product is
            // within 2 ulp of the direct evaluation, far inside the parity bar.
            // (ks = e^{+dtau/|mu|} of the layer above the zone: it overflows where that layer is optically thick for the direction --
            // dtau/|mu| > 709, e.g. a thin thick slab at N = 256 -- while x underflows, and the product would be NaN where the
            // direct value is at most 1.  Such directions evaluate the second exponential directly; kFactorMax keeps the product
            // form well inside the range in which neither factor has lost bits.)
            const double ks = exp((t_bd - t_bs) / mu);
            const bool ks_ok = ks < kFactorMax;
            auto row = [&](int t) {
                const double tt = s_tau[t], e0 = s_e0[t], eT = s_eT[t];
                // ((tt - t_bd) rmu instead of (tt - t_bd) / mu: one rounding of the argument more, |argument| x 1e-16 relative in
                // the exponential -- inside its own rounding; the division was a fifth of the kernel)
                const double x = exp((tt - t_bd) * rmu);
                double xs = x * ks;
                if (!ks_ok) xs = exp((tt - t_bs) * rmu);
                const double before = z ? Ib * x : 0.0;
                const double direct = near ? q * F0 * e0 * (tt - t_bd) / mu0 : gd * q * F0 * (e0 - e_bd * x);
                const double surf = gs * qm * R * (eT - e_bs * xs);
                const double vnode = gd * q * F0 * e0 + gs * qm * R * eT;
                return node ? vnode : before + direct + surf;
            };
            if (!up_half) {                                  // (the upward half only needs the last row of each zone)
                for (int tb = first_from(d.r0[z]); tb <= r1; tb += FU * nblk) {
                    double v[FU];
#pragma unroll
                    for (int u = 0; u < FU; ++u) v[u] = row(min(tb + u * nblk, r1));
#pragma unroll
                    for (int u = 0; u < FU; ++u) {
                        const int t = tb + u * nblk;
                        if (t <= r1 && valid) {
                            I1[(size_t)t * D + m] = v[u];
                            if (Iacc) Iacc[(size_t)t * D + m] = v[u];
                            if (sv) sv[(size_t)t * D + m] = v[u];
                        }
                    }
                }
            }
            vlast = row(r1);                                  // the zone's last row: boundary row of the next zone, surface row
            Ib = vlast;
        }
        s_sfc[tid] = vlast;                                   // I1[L-1][m]
        rdn = 1.0 / vlast;
    }
    if (!up_half) return;
    __syncthreads();
    // ---- upward, m = N + tid, zones bottom to top ----
    {
        const int j = valid ? tid : 0, m = N + j, mir = N - 1 - j;
        const double mu = g.mu[m];
        const double qa = d.wa * P0a[m] * c4pi, qam = d.wa * P0a[mir] * c4pi;
        const double pa = P0a[m], pam = P0a[mir], pr = P0r[m], prm = P0r[mir];
        const bool near = fabs(mu - mu0) < 0.0001;           // spec:204
        const bool node = j < 1;                             // the mu = 0+ node (spec:221-224)
        const double gd = mu0 / (mu0 + mu), gs = mu0 / (mu0 - mu);
        double Bv = rho * s_sfc[mir];                          // spec:211
        const double rmu = 1.0 / mu;
        double vlast = 0;
        for (int z = d.nz - 1; z >= 0; --z) {
            const bool bottom = z == d.nz - 1;
            const double qx = (d.wa * pa * d.fa[z] + d.wr[z] * pr * d.fr[z]) * c4pi;       // spec:242
            const double qxm = (d.wa * pam * d.fa[z] + d.wr[z] * prm * d.fr[z]) * c4pi;
            const double q = d.mix[z] ? qx : qa, qm = d.mix[z] ? qxm : qam;
            const double t_bu = bottom ? d.tau_bottom : s_tau[d.r1[z] + 1];
            const double t_bb = bottom ? s_tau[L - 1] : t_bu;
            const double t_su = bottom ? T : s_tau[d.r1[z]];
            const double e_bu = exp(-t_bu / mu0), e_su = exp(-(T - t_su) / mu0);
            const int r0 = d.r0[z], r1 = d.r1[z];
            // (one exponential per element, as in the downward half: the other two attenuations are that one times a factor
            // of the zone and the direction; above the bottom zone t_bb = t_bu)
            // (above the bottom zone ksu = e^{+dtau/mu} of the layer below the zone: the same overflow as ks of the downward half)
            const double ku = bottom ? exp(-(t_bu - t_bb) / mu) : 1.0, ksu = exp(-(t_su - t_bb) / mu);
            const bool ksu_ok = ksu < kFactorMax;
            auto row = [&](int t) {
                const double tt = s_tau[t], e0 = s_e0[t], eT = s_eT[t];
                const double yb = exp(-(t_bb - tt) * rmu);
                const double yu = bottom ? yb * ku : yb;
                double ys = yb * ksu;
                if (!ksu_ok) ys = exp(-(t_su - tt) * rmu);
                const double before = Bv * yb;
                const double direct = gd * q * F0 * (e0 - e_bu * yu);
                const double surf = near ? qm * R * eT * (t_su - tt) / mu0 : gs * qm * R * (eT - e_su * ys);
                const double vnode = gd * q * F0 * e0 + gs * qm * R * eT;
                return node ? vnode : before + direct + surf;
            };
            for (int tb = last_upto(r1); tb >= r0; tb -= FU * nblk) {
                double v[FU];
#pragma unroll
                for (int u = 0; u < FU; ++u) v[u] = row(max(tb - u * nblk, r0));
#pragma unroll
                for (int u = 0; u < FU; ++u) {
                    const int t = tb - u * nblk;
                    if (t >= r0 && valid) {
                        I1[(size_t)t * D + m] = v[u];
                        if (Iacc) Iacc[(size_t)t * D + m] = v[u];
                        if (sv) sv[(size_t)t * D + m] = v[u];
                    }
                }
            }
            vlast = row(r0);                                  // row r0 of this zone feeds the zone above
            Bv = vlast;
        }
        rup = 1.0 / vlast;                                    // I1[0][m]
    }
    if (do_conv && blk == 0) {
        const double a = block_pymax(rup, valid, s_red);
        const double bb = block_pymax(rdn, valid, s_red);
        const double r = outer_pymax(a, bb);
        if (tid == 0) {
            cv.ratio[b] = r;
            cv.norders[b] = 1;
            cv.status[b] = SOSRT_COL_OK;
            const int go = (r >= cv.tol) ? 1 : 0;
            cv.active[b] = go;
            if (go) atomicAdd(cv.nactive, 1);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_first_order_readme: the first order over a Lambertian surface as the reference's README writes it
// (README.md:126-171) -- a non-default option, PARITY UNPINNED: the reference has no runnable code for it (lam:274-276
// crashes; its other first-order blocks are the specular-beam ones above).  Per zone and direction:
// attenuated boundary row + scattering of the direct beam (as above) + scattering of the direct beam reflected
// isotropically by the ground,
//     int_0^1 mu'/(mu'-mu) omega P(mu,-mu')/(4 pi) 2 rho F0 e^{-T/mu0} (e^{-(T-tau)/mu'} - e^{-(T-tau_b)/mu'} e^{-|tau-tau_b|/|mu|}) dmu'
// by the trapezoid rule on the upward half of the direction grid; w_k P[m][2N-1-k] is the row k of the contraction
// matrices (the weights of the whole grid and of its upward half agree on that half).  The removable singularity at
// mu' = mu is replaced by its limit.  tau_b: the level a zone is entered at.  Ground: isotropic reflection of the
// downward first order (README.md:151 with the sign of README.md:215).  One workgroup per column, thread j = downward
// direction j, then upward direction N + j; FU rows at a time, their e^{-(T-tau)/mu'} shared through LDS.
// ------------------------------------------------------------------------------------------
__global__ void k_first_order_readme(Grid g, const double* __restrict__ w_all, const double* __restrict__ tau_all,
                                     const double* __restrict__ P0a_all, const double* __restrict__ P0r_all,
                                     const ColDesc* __restrict__ desc, double* __restrict__ I1_all, double* __restrict__ I_all,
                                     double* __restrict__ saved, size_t saved_col_stride, Conv cv, int do_conv) {
    constexpr int FU = 4;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int L = g.L, N = g.N, D = g.D, Wld = g.Wld;
    extern __shared__ double sm[];
    double* s_tau = sm;                      // [L]
    double* s_e0 = s_tau + L;                // [L] exp(-tau/mu0)
    double* s_mup = s_e0 + L;                // [N] quadrature nodes mu' = mu[N + k]
    double* s_e2 = s_mup + N;                // [N] exp(-(T - tau_b)/mu')
    double* s_e1 = s_e2 + N;                 // [FU][N] exp(-(T - tau_t)/mu') of the rows of a pass
    double* s_red = s_e1 + FU * N;           // [nw + 2]
    __shared__ ColDesc d;
    if (tid == 0) d = desc[b];
    const double* tau = tau_all + (size_t)b * L;
    const double* P0a = P0a_all + (size_t)b * D;
    const double* P0r = P0r_all ? P0r_all + (size_t)b * D : P0a;
    double* I1 = I1_all + (size_t)b * L * D;
    double* Iacc = I_all ? I_all + (size_t)b * L * D : nullptr;
    double* sv = saved ? saved + (size_t)b * saved_col_stride : nullptr;
    for (int t = tid; t < L; t += blockDim.x) s_tau[t] = tau[t];
    for (int k = tid; k < N; k += blockDim.x) s_mup[k] = g.mu[N + k];
    __syncthreads();
    const double mu0 = d.mu0, T = d.T, rho = d.rho;
    const double F0 = SOSRT_PI / mu0;
    const double R2 = 2 * rho * F0 * exp(-T / mu0);
    const double c4pi = 1.0 / (4 * SOSRT_PI);
    const bool valid = tid < N;
    for (int t = tid; t < L; t += blockDim.x) s_e0[t] = exp(-s_tau[t] / mu0);
    __syncthreads();
    auto store = [&](int t, int m, double v) {
        I1[(size_t)t * D + m] = v;
        if (Iacc) Iacc[(size_t)t * D + m] = v;
        if (sv) sv[(size_t)t * D + m] = v;
    };
    // exponentials of the quadrature nodes: one zone level, or the rows of a pass (mu' = 0: nothing arrives)
    auto fill_e2 = [&](double t_b) {
        __syncthreads();
        for (int k = tid; k < N; k += blockDim.x) s_e2[k] = s_mup[k] > 0 ? exp(-(T - t_b) / s_mup[k]) : 0.0;
    };
    auto fill_e1 = [&](int tb, int step, int lo, int hi) {
        __syncthreads();
        for (int i = tid; i < FU * N; i += blockDim.x) {
            const int u = i / N, k = i - u * N;
            const int t = min(max(tb + step * u, lo), hi);
            s_e1[i] = s_mup[k] > 0 ? exp(-(T - s_tau[t]) / s_mup[k]) : 0.0;
        }
        __syncthreads();
    };
    double rdn = 0, rup = 0;
    double sfc = 0;                                          // I1[L-1][m] of this thread's downward direction

    // ---- downward, m = tid ----
    {
        const int m = valid ? tid : 0;
        const double mu = g.mu[m];
        const double pa = P0a[m], pr = P0r[m];
        const bool near = fabs(mu + mu0) < 0.0001;
        const bool node = m > N - 2;                         // mu = 0-
        const double gd = mu0 / (mu0 + mu);
        double Ib = 0, vlast = 0;
        for (int z = 0; z < d.nz; ++z) {
            const double ca = d.wa * (d.mix[z] ? d.fa[z] : 1.0) * c4pi, cr = d.mix[z] ? d.wr[z] * d.fr[z] * c4pi : 0.0;
            const double q = ca * pa + cr * pr;
            const double t_b = z ? s_tau[d.r0[z] - 1] : 0.0;
            const double e_b = exp(-t_b / mu0);
            fill_e2(t_b);
            const int r0 = d.r0[z], r1 = d.r1[z];
            for (int tb = r0; tb <= r1; tb += FU) {
                fill_e1(tb, 1, r0, r1);
                double att[FU], acc[FU];
#pragma unroll
                for (int u = 0; u < FU; ++u) {
                    att[u] = node ? 0.0 : exp((s_tau[min(tb + u, r1)] - t_b) / mu);
                    acc[u] = 0;
                }
                for (int k = 0; k < N; ++k) {
                    const size_t wi = (size_t)(N + k) * Wld + m;
                    const double Qk = ca * g.Wa[wi] + (cr != 0.0 ? cr * g.Wr[wi] : 0.0);
                    const double mk = s_mup[k];
                    const double ker = node ? 1.0 : mk / (mk - mu);
                    const double e2 = s_e2[k];
#pragma unroll
                    for (int u = 0; u < FU; ++u) acc[u] += Qk * ker * (s_e1[u * N + k] - e2 * att[u]);
                }
#pragma unroll
                for (int u = 0; u < FU; ++u) {
                    const int t = tb + u;
                    if (t <= r1) {
                        const double tt = s_tau[t], e0 = s_e0[t];
                        const double before = z ? Ib * att[u] : 0.0;
                        const double direct = near ? q * F0 * e0 * (tt - t_b) / mu0 : gd * q * F0 * (e0 - e_b * att[u]);
                        const double v = node ? q * F0 * e0 + acc[u] * R2 : before + direct + acc[u] * R2;
                        if (valid) store(t, m, v);
                        vlast = v;
                    }
                }
            }
            Ib = vlast;
        }
        sfc = vlast;
        rdn = 1.0 / vlast;
    }
    // ---- ground: 2 rho int_0^1 I1_down(T, -mu') mu' dmu', isotropic ----
    double Bsfc;
    {
        const double term = valid ? w_all[tid] * sfc * (-g.mu[tid]) : 0.0;
        double v = term;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = v;
        __syncthreads();
        double S = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) S += s_red[i];
        Bsfc = 2 * rho * S;
    }
    // ---- upward, m = N + tid, zones bottom to top ----
    {
        const int j = valid ? tid : 0, m = N + j;
        const double mu = g.mu[m];
        const double pa = P0a[m], pr = P0r[m];
        const bool node = j < 1;                             // mu = 0+
        const double gd = mu0 / (mu0 + mu);
        double Bv = Bsfc, vlast = 0;
        for (int z = d.nz - 1; z >= 0; --z) {
            const bool bottom = z == d.nz - 1;
            const double ca = d.wa * (d.mix[z] ? d.fa[z] : 1.0) * c4pi, cr = d.mix[z] ? d.wr[z] * d.fr[z] * c4pi : 0.0;
            const double q = ca * pa + cr * pr;
            const double t_b = bottom ? s_tau[L - 1] : s_tau[d.r1[z] + 1];
            const double e_b = exp(-t_b / mu0);
            fill_e2(t_b);
            const int r0 = d.r0[z], r1 = d.r1[z];
            for (int tb = r1; tb >= r0; tb -= FU) {
                fill_e1(tb, -1, r0, r1);
                double att[FU], acc[FU], lim[FU];
#pragma unroll
                for (int u = 0; u < FU; ++u) {
                    const double tt = s_tau[max(tb - u, r0)];
                    att[u] = node ? 0.0 : exp(-(t_b - tt) / mu);
                    lim[u] = node ? 0.0 : exp(-(T - tt) / mu) * (t_b - tt) / mu;
                    acc[u] = 0;
                }
                for (int k = 0; k < N; ++k) {
                    const size_t wi = (size_t)(N + k) * Wld + m;
                    const double Qk = ca * g.Wa[wi] + (cr != 0.0 ? cr * g.Wr[wi] : 0.0);
                    const double mk = s_mup[k];
                    const bool same = !node && fabs(mk - mu) < 0.0001;           // the node mu' = mu itself
                    const double ker = node ? 1.0 : mk / (same ? 1.0 : mk - mu);
                    const double e2 = s_e2[k];
#pragma unroll
                    for (int u = 0; u < FU; ++u) acc[u] += Qk * (same ? lim[u] : ker * (s_e1[u * N + k] - e2 * att[u]));
                }
#pragma unroll
                for (int u = 0; u < FU; ++u) {
                    const int t = tb - u;
                    if (t >= r0) {
                        const double e0 = s_e0[t];
                        const double v = node ? q * F0 * e0 + acc[u] * R2 : Bv * att[u] + gd * q * F0 * (e0 - e_b * att[u]) + acc[u] * R2;
                        if (valid) store(t, m, v);
                        vlast = v;
                    }
                }
            }
            Bv = vlast;
        }
        rup = 1.0 / vlast;
    }
    if (do_conv) {
        const double a = block_pymax(rup, valid, s_red);
        const double bb = block_pymax(rdn, valid, s_red);
        const double r = outer_pymax(a, bb);
        if (tid == 0) {
            cv.ratio[b] = r;
            cv.norders[b] = 1;
            cv.status[b] = SOSRT_COL_OK;
            const int go = (r >= cv.tol) ? 1 : 0;
            cv.active[b] = go;
            if (go) atomicAdd(cv.nactive, 1);
        }
    }
}

static int round64(int n) { return (n + 63) / 64 * 64; }

void launch_first_order_readme(hipStream_t s, const Grid& g, const double* w, int B, const double* tau, const double* P0a,
                               const double* P0r, const ColDesc* desc, double* I1_out, double* I_out, double* saved,
                               size_t saved_col_stride, Conv cv, int do_conv) {
    const int nt = round64(g.N);
    const size_t shm = (size_t)(2 * g.L + 2 * g.N + 4 * g.N + nt / 64 + 4) * sizeof(double);
    hipLaunchKernelGGL(k_first_order_readme, dim3(B), dim3(nt), shm, s, g, w, tau, P0a, P0r, desc, I1_out, I_out, saved,
                       saved_col_stride, cv, do_conv);
}

void launch_first_order(hipStream_t s, const Grid& g, int B, const double* tau, const double* P0a, const double* P0r,
                        const ColDesc* desc, double* I1_out, double* I_out, double* saved, size_t saved_col_stride,
                        Conv cv, int do_conv) {
    const int nt = round64(g.N);
    const size_t shm = (size_t)(3 * g.L + nt + nt / 64 + 4) * sizeof(double);
    // row blocks per half column: enough workgroups to give every SIMD several waves (the rows are chains of fp64
    // exponentials), at least ~24 rows per block (each block also evaluates one boundary row per zone)
    int nblk = (4096 + 2 * B - 1) / (2 * B);
    nblk = nblk < 1 ? 1 : (nblk > 8 ? 8 : nblk);
    while (nblk > 1 && g.L / nblk < 24) --nblk;
    hipLaunchKernelGGL(k_first_order, dim3(B, 2, nblk), dim3(nt), shm, s, g, tau, P0a, P0r, desc, I1_out, I_out, saved,
                       saved_col_stride, cv, do_conv);
}

// ------------------------------------------------------------------------------------------
// improved_asymptotic_downward_radiance (In_limit:70-109) for one (layer, lane): the slice is
// zone-local [zs, t] (spec:334,353,372).
// ------------------------------------------------------------------------------------------
// k_smallmu: workgroups per (column, small lane, block of rows): the lane's L values of J (2 N doubles apart in memory) and
// the column's optical depths are staged in LDS once, then kSmallP threads share the window [tau_t - 5 |mu|, tau_t] of row t -- up
// to ~64 rows at N = 128, each an fp64 exponential; thread q takes the rows t-1-q, t-1-q-kSmallP, ... and receives the neighbouring
// integrand from the thread beside it, so every exponential is evaluated once; the partial trapezoid sums are added at the
// end.  (Until round 3 one thread walked the whole window in global memory, a chain of ~64 dependent strided loads and
// exponentials: 25 us per order for 64 columns, as long as the transport of the same order; now 7.)
constexpr int kSmallP = 8;              // threads per row
constexpr int kSmallRows = 256 / kSmallP;
__global__ __launch_bounds__(256) void k_smallmu(Grid g, const double* __restrict__ tau_all, const double* __restrict__ Jn_all,
                                                 double* __restrict__ In_all, const ColDesc* __restrict__ desc,
                                                 const int* __restrict__ active) {
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const int m = g.small_lanes[blockIdx.x];
    const ColDesc& d = desc[b];
    bool any = false;                                          // rewritten by the extrapolation in every zone: nothing to do
    for (int z = 0; z < d.nz; ++z) any = any || m < g.N - d.nfix[z];
    if (!any) return;
    extern __shared__ double sm_small[];
    double* s_tau = sm_small;                                  // [L]
    double* s_J = s_tau + g.L;                                 // [L]
    const double* tau = tau_all + (size_t)b * g.L;
    const double* J = Jn_all + (size_t)b * g.L * g.D + m;
    for (int t = threadIdx.x; t < g.L; t += blockDim.x) {
        s_tau[t] = tau[t];
        s_J[t] = J[(size_t)t * g.D];
    }
    __syncthreads();
    const double mu = g.mu[m];
    const int q = threadIdx.x & (kSmallP - 1);
    for (int t0 = blockIdx.z * kSmallRows; t0 < g.L; t0 += gridDim.z * kSmallRows) {     // (uniform trip count: shuffles inside)
        const int t = t0 + threadIdx.x / kSmallP;
        const bool row = t < g.L;
        const int tc = row ? t : g.L - 1;
        int z = 0;
        while (z + 1 < d.nz && tc > d.r1[z]) ++z;
        const int zs = d.r0[z];
        const bool wanted = row && m < g.N - d.nfix[z];          // else rewritten by the extrapolation anyway
        const double Jt = s_J[tc], tt = s_tau[tc];
        double val;
        if (fabs(mu) < kMuVerySmall) {                            // both Taylor branches (In_limit:79-93)
            double slope = 0.0;
            if (tc > zs) slope = (Jt - s_J[tc - 1]) / (tt - s_tau[tc - 1]);
            val = -Jt + mu * slope;
        } else {
            const double lim = tt - 5 * fabs(mu);                 // In_limit:97-98
            double acc = 0;
            double fcarry = Jt * exp((tt - tt) / mu);             // integrand at the row above this round's first (thread 0's neighbour)
            bool bad = !isfinite(fcarry);
            bool alive = wanted;
            const int gsh = (threadIdx.x & 63) & ~(kSmallP - 1);
            // round j: thread q evaluates row sq = t - 1 - q - kSmallP j; rows are taken while s >= zs and tau[s] >= lim (the
            // reference's slice ends at the first row that fails: a thread's row counts only if the rows before it did)
            for (int j = 0;; ++j) {
                const int sq = tc - 1 - q - kSmallP * j;
                const int sc = sq < 0 ? 0 : sq;
                const bool mine = alive && sq >= zs && s_tau[sc] >= lim;
                const unsigned long long bal = __ballot(mine);
                if (bal == 0) break;
                const unsigned grp = (unsigned)(bal >> gsh) & ((1u << kSmallP) - 1);
                const unsigned upto = (2u << q) - 1;
                const bool ok = (grp & upto) == upto;
                alive = grp == (1u << kSmallP) - 1;
                const double f = ok ? s_J[sc] * exp((tt - s_tau[sc]) / mu) : 0.0;
                double fup = __shfl_up(f, 1, kSmallP);             // the integrand one row below in the sum's order (s + 1)
                if (q == 0) fup = fcarry;
                if (ok) {
                    bad = bad || !isfinite(f);
                    acc += (s_tau[sc + 1] - s_tau[sc]) * (fup + f) / 2;
                }
                fcarry = __shfl(f, kSmallP - 1, kSmallP);
            }
#pragma unroll
            for (int o = 1; o < kSmallP; o <<= 1) acc += __shfl_xor(acc, o, kSmallP);
            int badi = bad;
#pragma unroll
            for (int o = 1; o < kSmallP; o <<= 1) badi |= __shfl_xor(badi, o, kSmallP);
            bad = badi != 0;
            val = (!(tt >= lim) || bad) ? -Jt : -acc / mu;        // empty window, In_limit:104-105
        }
        if (wanted && q == 0) In_all[((size_t)b * g.L + t) * g.D + m] = val;
    }
}

void launch_smallmu(hipStream_t s, const Grid& g, int B, const double* tau, const double* Jn, double* In,
                    const ColDesc* desc, const int* active) {
    if (g.nsmall <= 0) return;
    const size_t shm = 2 * (size_t)g.L * sizeof(double);
    if (shm > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(k_smallmu), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    // row blocks: enough workgroups for four per CU
    int nblk = (1024 + g.nsmall * B - 1) / (g.nsmall * B);
    const int most = (g.L + kSmallRows - 1) / kSmallRows;
    nblk = nblk < 1 ? 1 : (nblk > most ? most : nblk);
    hipLaunchKernelGGL(k_smallmu, dim3(g.nsmall, B, nblk), dim3(256), shm, s, g, tau, Jn, In, desc, active);
}

// ------------------------------------------------------------------------------------------
// k_transport: one order of transport for one column per workgroup.
//
// Thread j carries the downward direction m = j through the layers top to bottom as the
// first-order recurrence
//     D_t = E D_{t-1} - (dl/2) (J_{t-1} E + J_t) / mu ,   E = exp(dl / mu),  dl = tau_t - tau_{t-1}
// which is the reference's trapezoid sum  -trapz(J exp((tau_t - tau)/mu), tau) / mu  (spec:336-340)
// evaluated incrementally, then the upward direction m = N + j bottom to top as
//     U_t = E U_{t+1} + (dl/2) (J_t + J_{t+1} E) / mu ,   E = exp(-dl / mu), dl = tau_{t+1} - tau_t
// (spec:393-399), with the reference's zone restarts: the upward quadrature skips the interval
// under each zone (spec:413,433) and restarts from the already blended boundary row
// (SURVEY H5).  Rows are exchanged through LDS for the two mu -> 0 treatments that couple
// directions: the extrapolation of the last downward angles (In_limit:113-141, here a fixed
// linear map) and the upward second-difference search and blend (spec:402-409).
// ------------------------------------------------------------------------------------------
// The chunk loops are written branch-free (clamped row indices, selects instead of predicated
// blocks, reciprocal of mu hoisted) so that the compiler can interleave the TC independent
// chains of a chunk: with one wave per SIMD, instruction-level parallelism is the only latency
// hiding there is.  The loads run two chunks ahead of the recurrence.  With ETAB the layer
// attenuations exp(-dtau/|mu|) come from a table built once per solve (k_attenuation) instead of
// 2 L N exponentials per order.
// REPAIR: only the columns flagged by k_transport_fast, upward sweep only; the flagged sweep has
// already been added to I, so the correction I += (new - old) is applied (exact where new == old).
template <int MAXT, bool ACC, bool SAVED, bool ETAB, bool REPAIR = false>
__global__ __launch_bounds__(MAXT) void k_transport(TransportArgs a) {
    const int b = blockIdx.x;
    if (ACC && !a.cv.active[b]) return;
    if (REPAIR && !a.cv.redo[b]) return;
    const Grid& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int L = g.L, N = g.N, D = g.D;
    const int SR = blockDim.x + 2;
    extern __shared__ double sm[];
    double* s_tau = sm;                      // [L]
    double* s_rows = s_tau + L;              // [2][TC][SR]
    double* s_sfc = s_rows + 2 * TC * SR;    // [blockDim]
    double* s_mup = s_sfc + blockDim.x;      // [blockDim] upward half of the direction grid
    double* s_red = s_mup + blockDim.x;      // [nw + 1]
    __shared__ ColDesc d;                    // dynamically indexed: keep it out of scratch
    __shared__ FixTab s_fix[4];              // the extrapolation tables, one per tau_ref bucket (a zone names its bucket)
    if (tid == 0) d = a.desc[b];
    s_mup[tid] = tid < N ? g.mu[N + tid] : 1.0;
    {
        for (int zz = 0; zz < 4; ++zz) {
            const double* src = reinterpret_cast<const double*>(&g.fix[zz]);
            double* dst = reinterpret_cast<double*>(&s_fix[zz]);
            for (int i = tid; i < (int)(sizeof(FixTab) / sizeof(double)); i += blockDim.x) dst[i] = src[i];
        }
    }
    const double* __restrict__ tau = a.tau + (size_t)b * L;
    const double* __restrict__ J = a.Jn + (size_t)b * L * D;
    const double* __restrict__ Et = ETAB ? a.Etab + (size_t)(a.erep ? a.erep[b] : b) * L * D : nullptr;
    double* __restrict__ In = a.In + (size_t)b * L * D;
    double* __restrict__ Iacc = ACC ? a.I + (size_t)b * L * D : nullptr;
    double* __restrict__ sv = SAVED ? a.saved + (size_t)b * a.saved_col_stride : nullptr;
    for (int t = tid; t < L; t += blockDim.x) s_tau[t] = tau[t];
    __syncthreads();

    const bool valid = tid < N;
    const int tidc = valid ? tid : N - 1;    // clamped lane for loads of idle threads
    double rdn_v = 0, rdn_i = 1, rup_v = 0, rup_i = 1;
    int par = 0;                             // LDS row-buffer parity: one barrier per chunk

    // =============================== downward ===============================
    if (REPAIR) {
        const size_t row = (size_t)(L - 1) * D;
        const double x = In[row + tidc];
        s_sfc[tid] = x;
        rdn_v = x;
        rdn_i = ACC ? Iacc[row + tidc] : 1.0;
    } else {
        const int m = tidc;
        const double mu = valid ? g.mu[m] : -1.0;
        const bool tr = valid && tid <= N - 2;
        const bool small = tr && fabs(mu) < kMuThreshold;       // spec:333
        const bool stdl = tr && !small;
        const double rmu = 1.0 / (stdl ? mu : -1.0);
        double Dv = 0, Jprev = 0;
        int z = 0, nfx = 0, s0 = 0, ns = 0;
        double c[kFixMaxSrc] = {0, 0, 0, 0, 0};
        bool fixlane = false;
        auto load_fix = [&](int zz) {
            nfx = d.nfix[zz];
            const FixTab& ft = s_fix[d.fixtab[zz]];
            s0 = ft.s0; ns = ft.ns;
            fixlane = valid && nfx > 0 && tid >= N - nfx;
            const int i = fixlane ? N - 1 - tid : 0;
#pragma unroll
            for (int q = 0; q < kFixMaxSrc; ++q) c[q] = (fixlane && q < ns) ? ft.C[i * ns + min(q, ns - 1)] : 0.0;
        };
        load_fix(0);
        // Three register buffers rotate through the roles {being computed, next, after next}.  The
        // rotation is unrolled (no register copies): copying a buffer whose loads are still in
        // flight would force a wait on the newest loads and serialise the pipeline.
        double J0[TC], I0[TC], E0[TC], S0[TC], J1[TC], I1[TC], E1[TC], S1[TC], J2[TC], I2[TC], E2[TC], S2[TC];
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            S0[u] = S1[u] = S2[u] = 0; E0[u] = E1[u] = E2[u] = 0; I0[u] = I1[u] = I2[u] = 0;
        }
        int tp = 0, zp = 0;                                          // prefetch cursor
        // Always issues the same number of loads (rows are clamped past the end): the wait counters the
        // compiler derives are then exact and leave the younger chunk in flight.  The masked loads of the
        // small-mu lanes (values written by k_smallmu) go first so that they never are the youngest.
        auto fetch = [&](double (&Jx)[TC], double (&Ix)[TC], double (&Ex)[TC], double (&Sx)[TC]) {
            const int tq = min(tp, L - 1);
            if (small) {
#pragma unroll
                for (int u = 0; u < TC; ++u) Sx[u] = In[(size_t)min(tq + u, L - 1) * D + m];
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const size_t row = (size_t)min(tq + u, L - 1) * D;
                Jx[u] = (J + row)[m];
                if (ACC) Ix[u] = (Iacc + row)[m];
                if (ETAB) Ex[u] = (Et + row)[m];
            }
            if (tp < L) {
                int len = d.r1[zp] - tp + 1;
                if (len > TC) len = TC;
                tp += len;
                if (tp > d.r1[zp] && zp + 1 < d.nz) ++zp;
            }
        };
        int t0 = 0;
        auto process = [&](double (&Jc)[TC], double (&Ic)[TC], double (&Ec)[TC], double (&Sc)[TC]) {
            if (t0 > d.r1[z]) { ++z; load_fix(z); }
            int len = d.r1[z] - t0 + 1;
            if (len > TC) len = TC;
            double E[TC], cc[TC], v[TC];
            // independent part: attenuation and source term of every row of the chunk
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int t = min(t0 + u, L - 1);
                const double dl = s_tau[t] - s_tau[max(t - 1, 0)];   // 0 at t = 0: E = 1, source 0
                const double Jp = u == 0 ? Jprev : Jc[u - 1];
                E[u] = ETAB ? Ec[u] : exp(dl * rmu);
                cc[u] = -(dl * 0.5) * (Jp * E[u] + Jc[u]) * rmu;
            }
            // sequential part
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const double Dn = Dv * E[u] + cc[u];
                Dv = u < len ? Dn : Dv;
                v[u] = stdl ? Dn : (small ? Sc[u] : 0.0);
            }
            if (nfx > 0) {                                              // block-uniform
                double* rows = s_rows + par * TC * SR;
#pragma unroll
                for (int u = 0; u < TC; ++u) rows[u * SR + tid] = v[u];
                __syncthreads();
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    double acc = 0;
#pragma unroll
                    for (int q = 0; q < kFixMaxSrc; ++q) acc += c[q] * rows[u * SR + s0 + min(q, ns - 1)];
                    v[u] = fixlane ? acc : v[u];
                }
                par ^= 1;
            }
            if (valid) {
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const size_t row = (size_t)(t0 + u) * D;
                    if (u < len) {
                        (In + row)[m] = v[u];
                        if (ACC) (Iacc + row)[m] = Ic[u] + v[u];
                        if (SAVED) (sv + row)[m] = v[u];
                    }
                }
            }
            double vlast = 0, Jlast = 0, Ilast = 0;
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                vlast = u == len - 1 ? v[u] : vlast;
                Jlast = u == len - 1 ? Jc[u] : Jlast;
                Ilast = u == len - 1 ? Ic[u] + v[u] : Ilast;
            }
            Jprev = Jlast;
            if (t0 + len == L) { s_sfc[tid] = vlast; rdn_v = vlast; rdn_i = Ilast; }
            if (t0 + len - 1 == d.r1[z]) Dv = vlast;        // the next zone starts from the final row (spec:359,378)
            t0 += len;
        };
        fetch(J0, I0, E0, S0);
        fetch(J1, I1, E1, S1);
        for (;;) {
            if (t0 >= L) break;
            fetch(J2, I2, E2, S2); process(J0, I0, E0, S0);
            if (t0 >= L) break;
            fetch(J0, I0, E0, S0); process(J1, I1, E1, S1);
            if (t0 >= L) break;
            fetch(J1, I1, E1, S1); process(J2, I2, E2, S2);
        }
    }
    __syncthreads();

    // =============================== surface ===============================
    double Bv = 0;
    if (d.surface == SOSRT_SURFACE_SPECULAR) {
        Bv = valid ? d.rho * s_sfc[N - 1 - tid] : 0.0;              // spec:397
    } else if (d.surface == SOSRT_SURFACE_LAMBERTIAN || d.surface == SOSRT_SURFACE_LAMBERTIAN_README) {
        // -2 rho trapz(In[L-1, rev] mu[rev], mu[rev]), rev = N-2 .. 0   (lam:399), descending abscissae
        double term = 0;
        if (tid <= N - 3) {
            const int k0 = N - 2 - tid, k1 = k0 - 1;
            const double x0 = g.mu[k0], x1 = g.mu[k1];
            term = (x1 - x0) * (s_sfc[k1] * x1 + s_sfc[k0] * x0) / 2;
        }
        const double S = block_sum(term, s_red);
        Bv = (d.surface == SOSRT_SURFACE_LAMBERTIAN_README ? 2 : -2) * d.rho * S;      // lam:399 as coded (negative), or README.md:215
    }

    // =============================== upward ===============================
    int status = SOSRT_COL_OK;
    {
        const int j = tidc;
        const bool tr = valid && tid > 0;
        const double mu = tr ? s_mup[j] : 1.0;
        const double rmu = 1.0 / mu;
        double U = Bv, Jnext = 0;
        int z = d.nz - 1;
        double J0[TC], I0[TC], E0[TC], O0[TC], J1[TC], I1[TC], E1[TC], O1[TC], J2[TC], I2[TC], E2[TC], O2[TC];
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            E0[u] = E1[u] = E2[u] = 0; I0[u] = I1[u] = I2[u] = 0; O0[u] = O1[u] = O2[u] = 0;
        }
        int tp = L - 1, zp = d.nz - 1;
        auto fetch = [&](double (&Jx)[TC], double (&Ix)[TC], double (&Ex)[TC], double (&Ox)[TC]) {
            const int tq = max(tp, 0);
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const size_t row = (size_t)max(tq - u, 0) * D + N;
                Jx[u] = (J + row)[j];
                if (ACC) Ix[u] = (Iacc + row)[j];
                if (REPAIR) Ox[u] = (In + row)[j];           // what the flagged sweep added to I
                if (ETAB) Ex[u] = (Et + row)[j];
            }
            if (tp >= 0) {
                int len = tp - d.r0[zp] + 1;
                if (len > TC) len = TC;
                tp -= len;
                if (tp < d.r0[zp] && zp > 0) --zp;
            }
        };
        int t0 = L - 1;
        auto process = [&](double (&Jc)[TC], double (&Ic)[TC], double (&Ec)[TC], double (&Oc)[TC]) {
            if (t0 < d.r0[z]) --z;
            int len = t0 - d.r0[z] + 1;
            if (len > TC) len = TC;
            const int zr1 = d.r1[z];
            double E[TC], cc[TC], v[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int t = max(t0 - u, 0);
                const double dl = s_tau[min(t + 1, L - 1)] - s_tau[t];   // 0 at t = L-1: E = 1, source 0
                const double Jx = u == 0 ? Jnext : Jc[u - 1];
                E[u] = ETAB ? Ec[u] : exp(-dl * rmu);
                // first row of a zone: attenuate the boundary only (spec:413-419,433-439)
                const double src = (dl * 0.5) * (Jc[u] + Jx * E[u]) * rmu;
                cc[u] = (t == zr1) ? 0.0 : src;
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const double Un = U * E[u] + cc[u];
                U = u < len ? Un : U;
                v[u] = (tid == 0) ? Jc[u] : Un;                         // spec:401
            }
            double* rows = s_rows + par * TC * SR;
#pragma unroll
            for (int u = 0; u < TC; ++u) rows[u * SR + tid] = v[u];
            __syncthreads();
            par ^= 1;
            // second-difference search from the first upward angle (spec:403-406), all rows of the chunk
            unsigned long long mk[TC];
            const int lc = min(max(lane, 1), N - 3);
            const bool cand = lane >= 1 && lane <= N - 3;
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const double* row = rows + u * SR;
                const double x0 = row[lc], x1 = row[lc + 1], x2 = row[lc + 2];
                mk[u] = __ballot(cand && !(fabs((x0 - x1) - (x1 - x2)) > 0.0001));
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                if (u < len) {
                    const double* row = rows + u * SR;
                    int ks = -1;
                    if (mk[u]) ks = __ffsll((long long)mk[u]) - 1;
                    else {
                        for (int q = 64; q <= N - 3; ++q) {
                            const double x0 = row[q], x1 = row[q + 1], x2 = row[q + 2];
                            if (!(fabs((x0 - x1) - (x1 - x2)) > 0.0001)) { ks = q; break; }
                        }
                        if (ks < 0) status = SOSRT_COL_INDEXERROR;       // the reference raises IndexError here
                    }
                    const int kf = max(ks, 0) + 1;
                    const double w = mu / s_mup[kf];
                    const double bl = (1 - w) * row[0] + w * row[kf];    // spec:407-409
                    v[u] = (tr && tid < kf) ? bl : v[u];
                }
            }
            if (status != SOSRT_COL_OK) { t0 = -1; return; }
            if (valid) {
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const size_t row = (size_t)max(t0 - u, 0) * D + N;
                    if (u < len) {
                        (In + row)[j] = v[u];
                        if (ACC) (Iacc + row)[j] = REPAIR ? Ic[u] + (v[u] - Oc[u]) : Ic[u] + v[u];
                        if (SAVED) (sv + row)[j] = v[u];
                    }
                }
            }
            double vlast = 0, Jlast = 0, Ilast = 0;
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                vlast = u == len - 1 ? v[u] : vlast;
                Jlast = u == len - 1 ? Jc[u] : Jlast;
                Ilast = u == len - 1 ? (REPAIR ? Ic[u] + (v[u] - Oc[u]) : Ic[u] + v[u]) : Ilast;
            }
            Jnext = Jlast;
            if (t0 - len + 1 == 0) { rup_v = vlast; rup_i = Ilast; }
            if (t0 - len + 1 == d.r0[z] && z > 0 && tr) U = vlast;       // blended row feeds the zone above
            t0 -= len;
        };
        fetch(J0, I0, E0, O0);
        fetch(J1, I1, E1, O1);
        for (;;) {
            if (t0 < 0) break;
            fetch(J2, I2, E2, O2); process(J0, I0, E0, O0);
            if (t0 < 0) break;
            fetch(J0, I0, E0, O0); process(J1, I1, E1, O1);
            if (t0 < 0) break;
            fetch(J1, I1, E1, O1); process(J2, I2, E2, O2);
        }
    }
    if (status != SOSRT_COL_OK) {
        if (tid == 0) {
            a.cv.status[b] = status;
            if (ACC) {
                a.cv.active[b] = 0;
                a.cv.norders[b] = a.order;
                atomicSub(a.cv.nactive, 1);
            }
        }
        return;
    }
    if (ACC) {
        const double ra = block_pymax(rup_v / rup_i, valid, s_red);
        const double rb = block_pymax(rdn_v / rdn_i, valid, s_red);
        const double r = outer_pymax(ra, rb);
        if (tid == 0) {
            a.cv.ratio[b] = r;
            a.cv.norders[b] = a.order;
            if (!(r >= a.cv.tol)) {
                a.cv.active[b] = 0;
                atomicSub(a.cv.nactive, 1);
            }
        }
    } else if (tid == 0 && a.cv.status) {
        a.cv.status[b] = SOSRT_COL_OK;
    }
}

// attenuation table of a column: E[t][m] = exp((tau_t - tau_{t-1}) / mu_m) for the downward lanes,
// exp(-(tau_{t+1} - tau_t) / mu_m) for the upward ones; 0 for lanes that are not transported.
__global__ void k_attenuation(Grid g, int B, const double* __restrict__ tau_all, double* __restrict__ E_all,
                              const int* __restrict__ erep) {
    const int b = blockIdx.y;                                 // grid (slices, columns): whole workgroups of the
    if (erep && erep[b] != b) return;                        // columns that share an earlier column's table leave at once
    const int LD = g.L * g.D;
    const double* tau = tau_all + (size_t)b * g.L;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < LD; i += gridDim.x * blockDim.x) {
        const int t = i / g.D, m = i % g.D;
        const double mu = g.mu[m];
        double E = 0;
        if (m < g.N) {
            const bool stdl = m <= g.N - 2 && !(fabs(mu) < kMuThreshold);
            const double dl = tau[t] - tau[max(t - 1, 0)];
            if (stdl) E = exp(dl * (1.0 / mu));
        } else {
            const double dl = tau[min(t + 1, g.L - 1)] - tau[t];
            if (m > g.N) E = exp(-dl * (1.0 / mu));
        }
        E_all[(size_t)b * LD + i] = E;
    }
}

void launch_attenuation(hipStream_t s, const Grid& g, int B, const double* tau, double* Etab, const int* erep) {
    // with shared tables few columns have work: more slices per column then
    const int slices = erep ? 64 : 8;
    hipLaunchKernelGGL(k_attenuation, dim3(slices, B), dim3(256), 0, s, g, B, tau, Etab, erep);
}

// Columns of a parameter sweep usually share a few optical-depth profiles (the headline sweep: 8
// profiles for 512 columns), and the attenuation table depends on nothing else.  Sharing one table
// per profile keeps the tables in L2 / MALL instead of streaming one per column from HBM in every
// order.  k_tau_hash: one 64-bit hash per column; k_tau_rep: the first earlier column with the same
// hash and, checked value by value, the same profile.
__global__ void k_tau_rep(int L, const double* __restrict__ tau_all, const unsigned long long* __restrict__ hash,
                          int* __restrict__ erep, const int* __restrict__ need_small, int* __restrict__ host_flag, int tag) {
    const int b = blockIdx.x, lane = threadIdx.x;            // one wave per column
    // k_prepare (the previous kernel of the stream) has decided whether any |mu| < 0.01 lane keeps its k_smallmu value: tell
    // the host now, so that the order loop can drop those launches -- and the ring kernel their LDS rows, which lets two
    // columns share a CU -- from the second order on instead of the fifth
    if (host_flag && b == 0 && lane == 0) {
        __hip_atomic_store(host_flag, need_small[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_flag + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned long long hb = hash[b];
    const double* tb = tau_all + (size_t)b * L;
    int rep = b;
    for (int c0 = 0; c0 < b && rep == b; c0 += 64) {
        const int c = c0 + lane;
        unsigned long long cand = __ballot(c < b && hash[c] == hb);
        while (cand && rep == b) {
            const int cc = c0 + __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const double* tc = tau_all + (size_t)cc * L;
            bool same = true;
            for (int i = lane; i < L; i += 64) same = same && (tb[i] == tc[i]);
            if (__ballot(!same) == 0) rep = cc;
        }
    }
    if (lane == 0) erep[b] = rep;
}
void launch_tau_groups(hipStream_t s, const Grid& g, int B, const double* tau, unsigned long long* hash, int* erep,
                       const int* need_small, int* host_flag, int tag) {
    // (the hashes come from k_prepare, the kernel before this one)
    hipLaunchKernelGGL(k_tau_rep, dim3(B), dim3(64), 0, s, g.L, tau, hash, erep, need_small, host_flag, tag);
}

template <int MAXT>
static void launch_transport_t(hipStream_t s, dim3 grid, dim3 block, size_t shm, const TransportArgs& a, int mode) {
    // mode 0: general kernel, 1: wave-independent fast kernel, 2: general kernel repairing flagged columns,
    // 3: the fast kernel's sweeps fed through an LDS ring, 4: the chunk-parallel kernel (transport_scan.hip)
    if (mode == 1) {
        launch_transport_fast(s, grid, block, a);
    } else if (mode == 3) {
        launch_transport_ring(s, grid, a, g_ring_slots);
    } else if (mode == 4) {
        launch_transport_scan(s, grid, a);
    } else if (mode == 2) {
        if (a.accumulate) {
            if (a.saved) hipLaunchKernelGGL((k_transport<MAXT, true, true, true, true>), grid, block, shm, s, a);
            else hipLaunchKernelGGL((k_transport<MAXT, true, false, true, true>), grid, block, shm, s, a);
        } else {
            hipLaunchKernelGGL((k_transport<MAXT, false, false, true, true>), grid, block, shm, s, a);
        }
    } else if (a.accumulate) {
        if (a.Etab) {
            if (a.saved) hipLaunchKernelGGL((k_transport<MAXT, true, true, true>), grid, block, shm, s, a);
            else hipLaunchKernelGGL((k_transport<MAXT, true, false, true>), grid, block, shm, s, a);
        } else {
            if (a.saved) hipLaunchKernelGGL((k_transport<MAXT, true, true, false>), grid, block, shm, s, a);
            else hipLaunchKernelGGL((k_transport<MAXT, true, false, false>), grid, block, shm, s, a);
        }
    } else {
        if (a.Etab) hipLaunchKernelGGL((k_transport<MAXT, false, false, true>), grid, block, shm, s, a);
        else hipLaunchKernelGGL((k_transport<MAXT, false, false, false>), grid, block, shm, s, a);
    }
}

unsigned long long* g_transport_stamps = nullptr;
int g_ring_slots = 3, g_ring_debug = 0;   // set by sosrt_debug_stamps (diagnostics)

// True when both mu -> 0 treatments of a column fit in wave 0 of k_transport_fast.
bool transport_fast_ok(const Plan& plan) {
    // every rewritten downward direction and its source directions must sit in the last wave
    for (int bkt = 0; bkt < 4; ++bkt)
        if (plan.fix[bkt].idx > 0 && (plan.fix[bkt].s0 >> 6) != ((plan.N - 1) >> 6)) return false;
    return true;
}

void launch_transport(hipStream_t s, const Grid& g, int B, const double* tau, const double* Jn, double* In, double* I,
                      double* saved, size_t saved_col_stride, const ColDesc* desc, Conv cv, int order,
                      int accumulate, const double* Etab, int mode, const int* erep, int live, const int* live_list,
                      int ring_slots, int scan_split, double* scan_scratch, int* scan_sync, int nzcap) {
    const int nt = round64(g.N);
    const size_t shm = (size_t)(g.L + 2 * TC * (nt + 2) + 2 * nt + nt / 64 + 2) * sizeof(double);
    TransportArgs a{g, tau, Jn, In, I, accumulate ? saved : nullptr, saved_col_stride, desc, cv, order, accumulate, Etab, Etab ? erep : nullptr, g_transport_stamps};
    a.slots = ring_slots;
    a.nzcap = nzcap > kRingZones ? nzcap : kRingZones;
    if (mode == 4 && scan_split && scan_scratch && scan_sync) { a.scan_split = 1; a.scan_scratch = scan_scratch; a.scan_sync = scan_sync; }
    if ((mode == 3 || mode == 4) && accumulate && live > 0 && live < B && live_list) {       // ring / scan kernel over the live columns only
        a.live = live;
        a.live_list = live_list;
        B = live;
    }
    // the register budget follows the workgroup size: one column needs few waves, so they may be fat
    if (nt <= 128) launch_transport_t<128>(s, dim3(B), dim3(nt), shm, a, mode);
    else if (nt <= 256) launch_transport_t<256>(s, dim3(B), dim3(nt), shm, a, mode);
    else if (nt <= 512) launch_transport_t<512>(s, dim3(B), dim3(nt), shm, a, mode);
    else launch_transport_t<1024>(s, dim3(B), dim3(nt), shm, a, mode);
}

// columns still iterating when the order budget is exhausted
// (and the caller's copies of the order counts and the status words, instead of two device-to-device copies behind it)
__global__ void k_finalize(int B, Conv cv, int max_orders, int* __restrict__ n_out, int* __restrict__ status_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int st = cv.status[b];
    if (cv.active[b]) { st = SOSRT_COL_MAXORDERS; cv.status[b] = st; }
    if (n_out) n_out[b] = cv.norders[b];
    if (status_out) status_out[b] = st;
}
void launch_finalize(hipStream_t s, int B, Conv cv, int max_orders, int* n_out, int* status_out) {
    hipLaunchKernelGGL(k_finalize, dim3((B + 127) / 128), dim3(128), 0, s, B, cv, max_orders, n_out, status_out);
}

// ------------------------------------------------------------------------------------------
// k_fluxes: one wavefront per (column, layer)    (graphe:157-158, crit:380-381)
// ------------------------------------------------------------------------------------------
__global__ void k_fluxes(Grid g, int B, const double* __restrict__ tau_all, const double* __restrict__ I_all,
                         const ColDesc* __restrict__ desc, int beam_norm, double* __restrict__ fdn,
                         double* __restrict__ fup) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B * g.L) return;
    const int lane = threadIdx.x & 63, b = row / g.L, t = row % g.L;
    const double* I = I_all + (size_t)row * g.D;
    double sd = 0, su = 0;
    for (int k = lane; k < g.N; k += 64) {
        sd += g.wflux_dn[k] * I[k];
        su += g.wflux_up[k] * I[g.N + k];
    }
    sd = wave_sum(sd);
    su = wave_sum(su);
    if (lane == 0) {
        const ColDesc& d = desc[b];
        const double* tau = tau_all + (size_t)b * g.L;
        const double F0 = SOSRT_PI / d.mu0;
        const double fb = beam_norm ? F0 : F0 / (4 * SOSRT_PI);
        fdn[row] = sd - fb * exp(-tau[t] / d.mu0);
        fup[row] = su + fb * d.rho * exp(-(2 * tau[g.L - 1] - tau[t]) / d.mu0);
    }
}
void launch_fluxes(hipStream_t s, const Grid& g, int B, const double* tau, const double* I, const ColDesc* desc,
                   int beam_norm, double* fdn, double* fup) {
    const int rows = B * g.L;
    hipLaunchKernelGGL(k_fluxes, dim3((rows + 3) / 4), dim3(256), 0, s, g, B, tau, I, desc, beam_norm, fdn, fup);
}

// ------------------------------------------------------------------------------------------
// helper-level kernels (In_limit:70,113)
// ------------------------------------------------------------------------------------------
__global__ void k_limit_rows(Grid g, int R, int table, const double* __restrict__ rows, double* __restrict__ out) {
    const FixTab& ft = g.fix[table];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * ft.idx) return;
    const int r = e / ft.idx, i = e % ft.idx;
    double acc = 0;
    for (int q = 0; q < ft.ns; ++q) acc += ft.C[i * ft.ns + q] * rows[(size_t)r * g.N + ft.s0 + q];
    out[e] = acc;
}
void launch_limit_rows(hipStream_t s, const Grid& g, int R, int table, const double* rows, double* out) {
    const int n = R * kFixMaxIdx;
    hipLaunchKernelGGL(k_limit_rows, dim3((n + 127) / 128), dim3(128), 0, s, g, R, table, rows, out);
}

__global__ void k_asymptotic(int R, int stride, const int* __restrict__ len, const double* __restrict__ J,
                             const double* __restrict__ tau, const double* __restrict__ tau_t,
                             const double* __restrict__ mu, double* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int n = len[r];
    if (n <= 0) { out[r] = 0.0; return; }                       // In_limit:75
    const double* Jr = J + (size_t)r * stride;
    const double* tr = tau + (size_t)r * stride;
    const double m = mu[r], tt = tau_t[r];
    const double Jl = Jr[n - 1];
    if (fabs(m) < kMuVerySmall) {
        double slope = 0;
        if (n > 1) slope = (Jl - Jr[n - 2]) / (tr[n - 1] - tr[n - 2]);
        out[r] = -Jl + m * slope;
        return;
    }
    // general slice: tau_t need not be tau[n-1]; the window is every index with tau >= tau_t - 5|mu|
    const double lim = tt - 5 * fabs(m);
    double acc = 0, fprev = 0, xprev = 0;
    bool have = false, bad = false;
    for (int s = 0; s < n; ++s) {
        if (!(tr[s] >= lim)) continue;
        const double f = Jr[s] * exp((tt - tr[s]) / m);
        bad = bad || !isfinite(f);
        if (have) acc += (tr[s] - xprev) * (f + fprev) / 2;
        have = true; fprev = f; xprev = tr[s];
    }
    if (!have || bad) { out[r] = -Jl; return; }
    out[r] = -acc / m;
}
void launch_asymptotic(hipStream_t s, int R, int stride, const int* len, const double* J, const double* tau,
                       const double* tau_t, const double* mu, double* out) {
    hipLaunchKernelGGL(k_asymptotic, dim3((R + 63) / 64), dim3(64), 0, s, R, stride, len, J, tau, tau_t, mu, out);
}


// ------------------------------------------------------------------------------------------
// machine peaks measured on the box (denominators of the roofline fractions)
// ------------------------------------------------------------------------------------------
template <int NACC>
__global__ __launch_bounds__(256) void k_bench_mfma_f64(double* out, int iters) {
    f64x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double sres = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) sres += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sres == 12345.678) out[0] = sres;          // keeps the loop alive
}

__global__ __launch_bounds__(256) void k_bench_fma_f64(double* out, int iters) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 1e-3 * (threadIdx.x + i);
    const double a = 1.0 - 1e-12, b = 1e-12;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
    }
    double sres = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) sres += x[i];
    if (sres == 12345.678) out[0] = sres;
}

__global__ __launch_bounds__(256) void k_bench_copy(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

void launch_bench(hipStream_t s, int which, double* a, double* b, size_t n, int iters) {
    // which >= 10: MFMA variants, (which-10) = nacc_code*10 + blocks_per_cu_code; flops are per launch_bench_flops
    if (which >= 10) {
        const int code = which - 10, nacc = code / 10, bpc = code % 10;     // bpc: workgroups (4 waves) per CU
        const dim3 grid(256 * (bpc ? bpc : 1)), block(256);
        if (nacc == 0) hipLaunchKernelGGL(k_bench_mfma_f64<2>, grid, block, 0, s, a, iters);
        else if (nacc == 1) hipLaunchKernelGGL(k_bench_mfma_f64<4>, grid, block, 0, s, a, iters);
        else if (nacc == 2) hipLaunchKernelGGL(k_bench_mfma_f64<8>, grid, block, 0, s, a, iters);
        else hipLaunchKernelGGL(k_bench_mfma_f64<16>, grid, block, 0, s, a, iters);
        return;
    }
    if (which == 0) hipLaunchKernelGGL(k_bench_mfma_f64<8>, dim3(2048), dim3(256), 0, s, a, iters);
    else if (which == 1) hipLaunchKernelGGL(k_bench_copy, dim3(2048), dim3(256), 0, s, (const double2*)a, (double2*)b, n / 2);
    else hipLaunchKernelGGL(k_bench_fma_f64, dim3(2048), dim3(256), 0, s, a, iters);
}

}  // namespace sosrt
