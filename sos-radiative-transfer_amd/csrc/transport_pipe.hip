// k_transport_pipe: one order of transport (spec:326-449) for the latency-bound regime -- fewer live columns
// than compute units, so that a column is one workgroup alone on its CU and the time of an order is the time of
// one column.
//
// k_transport_ring gives a column two computing waves that do everything for a chunk of TC rows -- source terms,
// the TC dependent multiply-adds of the recurrence, the two mu -> 0 treatments, the running total, 16-24 stores --
// one chunk after the other: 2.1k cycles per chunk, 48 us per column and order, nearly all of it exposed latency
// of one wave per SIMD.  Here the same arithmetic (bit for bit) is split into a pipeline of specialised waves
// that work on different chunks at the same time, one workgroup barrier per tick:
//
//   tick q:   loaders    DMA chunk q+R of Jn and of the attenuation table into the LDS ring
//             chain      chunk q:   source terms and recurrence from the ring, raw rows into s_D[q mod 3]
//             treatment  chunk q-1: the mu -> 0 treatment of its rows, patched into s_D in place
//                                   (In_limit:113-141 as a linear map of the source lanes; spec:402-409 search + blend,
//                                   any number of candidates: no repair pass for these rows)
//             store      chunk q-2: In, running total (its rows of I prefetched a tick ahead), saved order
//
// The recurrence state crosses a zone boundary as the *treated* boundary row (spec:359,378) and the
// surface / top rows feed the reflection and the convergence test: for those few rows the chain wave evaluates the
// treatment itself, inline (the code of the ring kernel), so the pipeline never stalls on its own later stages.
//
// Lane mapping, chunk sequence (downward sweep rows ascending, then upward sweep rows descending), ring protocol
// and arithmetic are those of transport_ring.hip; N <= 128 (one 1-KiB piece per half row).
#include <type_traits>

#include "../../include/sosrt.h"
#include "kernels.hpp"
#include "transport_util.hpp"

namespace sosrt {

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr size_t kPipeLdsBytes = 152 * 1024;
constexpr int PIPE_NWL = 2;                                  // loader waves
constexpr int RS = 128;                                      // doubles per LDS row (one half row of the field)

template <int CNT>
__device__ __forceinline__ void pwait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}
__device__ __forceinline__ void pbarrier() {                 // leaves vector-memory operations in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void wait_outstanding(int n) {    // at most n (rounded down to a multiple of 4) loads in flight
    switch (min(n, 60) >> 2) {
        case 0: pwait_vm<0>(); break;
        case 1: pwait_vm<4>(); break;
        case 2: pwait_vm<8>(); break;
        case 3: pwait_vm<12>(); break;
        case 4: pwait_vm<16>(); break;
        case 5: pwait_vm<20>(); break;
        case 6: pwait_vm<24>(); break;
        case 7: pwait_vm<28>(); break;
        case 8: pwait_vm<32>(); break;
        case 9: pwait_vm<36>(); break;
        case 10: pwait_vm<40>(); break;
        case 11: pwait_vm<44>(); break;
        case 12: pwait_vm<48>(); break;
        case 13: pwait_vm<52>(); break;
        case 14: pwait_vm<56>(); break;
        default: pwait_vm<60>(); break;
    }
}

// zone bounds and chunk sequence of a column, by value (plain scalars for the helpers below)
struct Seq {
    int zend0, zend1, zbeg1, zbeg2, L, NCH, NQ;
};
__device__ __forceinline__ bool seq_touches(const Seq z, int lo, int hi) {
    return (z.zend0 >= lo && z.zend0 <= hi) || (z.zend1 >= lo && z.zend1 <= hi) || (z.zbeg1 >= lo && z.zbeg1 <= hi) ||
           (z.zbeg2 >= lo && z.zbeg2 <= hi);
}
// chunk q: downward rows q*TC + u (q < NCH), upward rows L-1 - (q-NCH)*TC - u
__device__ __forceinline__ int seq_first_row(const Seq z, int q) { return q < z.NCH ? q * TC : z.L - 1 - (q - z.NCH) * TC; }
__device__ __forceinline__ bool seq_special(const Seq z, int q) {
    const int t0 = seq_first_row(z, q);
    return q < z.NCH ? (t0 + TC >= z.L || seq_touches(z, t0, t0 + TC - 1)) : (t0 - TC < 0 || seq_touches(z, t0 - TC + 1, t0));
}
__device__ __forceinline__ int seq_zone_of(const Seq z, int t) {
    return (z.zbeg1 >= 0 && t >= z.zbeg1 ? 1 : 0) + (z.zbeg2 >= 0 && t >= z.zbeg2 ? 1 : 0);
}

template <bool ACC, bool SAVED>
__global__ __launch_bounds__(576) void k_transport_pipe(TransportArgs a, int NS) {
    int b = blockIdx.x;
    if (ACC && a.live > 0) {
        b = a.live_list[blockIdx.x];
        if (b < 0) return;                                   // fewer live columns than the host's (lagging) count
    } else if (ACC && !a.cv.active[b]) {
        return;
    }
    const Grid& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int L = g.L, N = g.N, D = g.D;
    const int nwc = (N + 63) >> 6;                           // chain waves (one per 64 directions)
    // roles
    const bool chain = wid < nwc;
    const bool loader = wid >= nwc && wid < nwc + PIPE_NWL;
    const int lid = wid - nwc;
    const bool treat = wid == nwc + PIPE_NWL;
    const int kw = wid - (nwc + PIPE_NWL + 1);               // store waves: rows 2 kw, 2 kw + 1 of a chunk
    const bool w0 = wid == 0;
    const bool wl = wid == ((N - 1) >> 6);                   // the chain wave that holds the mu -> 0- lanes
    constexpr int SLOT = 2 * TC * RS;                        // Jn, attenuation
    extern __shared__ double sm[];
    double* ring = sm;                                       // [NS][2][TC][RS]
    double* s_D = ring + (size_t)NS * SLOT;                  // [3][TC][RS] rows of In on their way from the chain to the stores
    double* s_sfc = s_D + 3 * TC * RS;                       // [128] surface row by downward lane
    double* s_red = s_sfc + 128;                             // [nw + 2]
    double* s_hd = s_red + 16;                               // [L + 1] half layer thicknesses
    double* s_mup = s_hd + L + 1;                            // [128] mu of the upward directions
    double* s_prmu = s_mup + 128;                            // [128] their reciprocals
    double* s_S = s_prmu + 128;                              // [nsmall][L] the |mu| < 0.01 lanes as written by k_smallmu
    __shared__ FixTab s_fix[kMaxZones];
    __shared__ int s_flag[2];                                // [0] redo with the general kernel, [1] IndexError
    const ColDesc* __restrict__ dg = a.desc + b;
    const int nz = dg->nz;
    const int zend0 = nz > 1 ? dg->r1[0] : -9, zend1 = nz > 2 ? dg->r1[1] : -9;
    const int zbeg1 = nz > 1 ? dg->r0[1] : -9, zbeg2 = nz > 2 ? dg->r0[2] : -9;
    const int nfix0 = dg->nfix[0], nfix1 = dg->nfix[1], nfix2 = dg->nfix[2];
    const int surface = dg->surface;
    const double rho = dg->rho;
    const int fbytes = L * D * 8, RB = D * 8;
    // buffer descriptors are built where a role needs them (they are four scalar registers each)
    const int NCH = (L + TC - 1) / TC, NQ = 2 * NCH, R = NS - 1;
    const Seq sq{zend0, zend1, zbeg1, zbeg2, L, NCH, NQ};
#define first_row(q_) seq_first_row(sq, (q_))
#define special(q_) seq_special(sq, (q_))
#define zone_of(t_) seq_zone_of(sq, (t_))

    // ------------------------------- loaders -------------------------------
    const __amdgpu_buffer_rsrc_t rJ = make_rsrc(a.Jn + (size_t)b * L * D, loader ? fbytes : 0);
    const __amdgpu_buffer_rsrc_t rE = make_rsrc(a.Etab + (size_t)(a.erep ? a.erep[b] : b) * L * D, loader ? fbytes : 0);
    auto issue = [&](int q) {
        const bool up = q >= NCH;
        const int j = up ? q - NCH : q;
        double* dst = ring + (size_t)(q % NS) * SLOT;
        const int half = up ? N * 8 : 0;
        const int vo = lane * 16;
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            if ((u & (PIPE_NWL - 1)) != lid) continue;
            const int row = up ? max(L - 1 - j * TC - u, 0) : min(j * TC + u, L - 1);
            const int so = row * RB + half;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (0 * TC + u) * RS), 16, vo, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rE, (lds_ptr_t)(dst + (1 * TC + u) * RS), 16, vo, so, 0, 0);
        }
    };
    constexpr int CNT = (TC / PIPE_NWL) * 2;                 // DMA instructions of a loader per chunk

    const int ncomp = nwc * 64;
    if (loader) {
        for (int q = 0; q < min(R, NQ); ++q) issue(q);
    } else if (chain) {
        for (int zz = 0; zz < kMaxZones; ++zz) {
            const double* src = reinterpret_cast<const double*>(&g.fix[dg->fixtab[zz]]);
            double* dst = reinterpret_cast<double*>(&s_fix[zz]);
            for (int i = tid; i < (int)(sizeof(FixTab) / sizeof(double)); i += ncomp) dst[i] = src[i];
        }
        if (tid < 2) s_flag[tid] = 0;
        const double* __restrict__ tau = a.tau + (size_t)b * L;
        for (int t = tid; t <= L; t += ncomp) s_hd[t] = (t == 0 || t == L) ? 0.0 : (tau[t] - tau[t - 1]) * 0.5;
        for (int k = tid; k < 128; k += ncomp) {
            const double m = k < N ? g.mu[N + k] : 1.0;
            s_mup[k] = m;
            s_prmu[k] = (k > 0 && k < N) ? 1.0 / m : 0.0;
        }
        const double* __restrict__ In0 = a.In + (size_t)b * L * D;
        for (int i = tid; i < g.nsmall * L; i += ncomp) {
            const int k = i / L, t = i - k * L;
            s_S[i] = In0[(size_t)t * D + g.small_lanes[k]];
        }
    }
    __syncthreads();                                          // (drains the loaders once: chunks 0 .. R-1 have landed)

    const bool valid = tid < N;                               // chain waves: thread = downward direction tid, then upward direction N + tid
    const int tidc = valid ? tid : N - 1;
    const int seam_barriers = surface == SOSRT_SURFACE_NONE ? 0 : (surface == SOSRT_SURFACE_SPECULAR ? 1 : 3);
    double rdn_v = 0, rdn_i = 1, rup_v = 0, rup_i = 1;
    // the two rows of the running total that the convergence test reads (spec:309), fetched before the sweeps: the
    // store waves rewrite them only after the chain has passed them
    double Itot_sfc = 0, Itot_top = 0;
    if (ACC && chain) {
        const double* __restrict__ Ib = a.I + (size_t)b * L * D;
        Itot_sfc = Ib[(size_t)(L - 1) * D + tidc];
        Itot_top = Ib[N + tidc];
    }

    // ------------------------------- chain state -------------------------------
    double sfc_own = 0, Bv = 0;
    // downward
    const double mu_dn = chain ? g.mu[tidc] : -1.0;
    const bool tr_dn = chain && valid && tidc <= N - 2;
    const bool small = tr_dn && fabs(mu_dn) < kMuThreshold;  // spec:333
    const bool stdl = tr_dn && !small;
    const double nrmu = stdl ? -1.0 / mu_dn : 0.0;
    const bool has_small = chain && wl && g.nsmall > 0;
    const int sbase = small ? (tidc - g.small_lanes[0]) * L : 0;
    double Dv = 0, Jprev = 0;
    double c[kFixMaxSrc] = {0, 0, 0, 0, 0};
    int sl[kFixMaxSrc] = {0, 0, 0, 0, 0};
    int nfx = 0;
    bool fixlane = false;
    auto load_fix = [&](int zz) {
        const FixTab& ft = s_fix[zz];
        nfx = zz == 0 ? nfix0 : (zz == 1 ? nfix1 : nfix2);
        const int ns = nfx < 2 ? 2 : (nfx < kFixMaxSrc ? nfx : kFixMaxSrc);     // In_limit:118-141
        fixlane = valid && nfx > 0 && tidc >= N - nfx;
        const int i = fixlane ? N - 1 - tidc : 0;
        const int s0 = nfx < 2 ? N - nfx - 2 : N - nfx - ns;
#pragma unroll
        for (int q = 0; q < kFixMaxSrc; ++q) {
            c[q] = (fixlane && q < ns) ? ft.C[i * ns + min(q, ns - 1)] : 0.0;
            sl[q] = (s0 + min(q, ns - 1)) & 63;
        }
    };
    if (chain && wl) load_fix(0);
    // upward
    const double mu_up = (chain && valid && tid > 0) ? g.mu[N + tidc] : 1.0;
    const bool tr_up = chain && valid && tid > 0;
    const double prmu = tr_up ? 1.0 / mu_up : 0.0;
    const int last_cand = min(N - 3, 61);
    bool notfound = false;
    double U = 0, Jnext = 0;
    // spec:401-409 for one row held across wave 0 (the rows whose treated value the recurrence needs)
    auto blend = [&](double x) {
        const double x1 = lane_up1(x), x2 = lane_up1(x1);
        const bool stop = lane >= 1 && lane <= last_cand && !(fabs((x - x1) - (x1 - x2)) > 0.0001);
        const unsigned long long mk = __ballot(stop);
        const int kf = mk ? __ffsll((long long)mk) : 1;
        notfound |= (mk == 0);
        const double r0 = readlane_f64(x, 0), rk = readlane_f64(x, kf);
        const double w = blend_weight(mu_up, readlane_f64(prmu, kf));
        const double bl = blend_val(w, r0, rk);
        return (tr_up && tid < kf) ? bl : x;
    };

    // ------------------------------- treatment wave -------------------------------
    // work item (uT, pT): row uT of the chunk, position pT next to mu = 0
    const int uT = lane >> 3, pT = lane & 7;
    // ------------------------------- store waves -------------------------------
    // diagnostics (sosrt_debug_stamps): cycles each wave spends working, i.e. outside the tick barriers
    unsigned long long busy = 0, t_in = a.stamps ? clock64() : 0;
    const unsigned long long t_begin = t_in;
#define TICK_BARRIER()                                            \
    do {                                                          \
        if (a.stamps) busy += clock64() - t_in;                   \
        pbarrier();                                               \
        if (a.stamps) t_in = clock64();                           \
    } while (0)
    // ------------------------------- the pipeline -------------------------------
    // Every role runs its own loop over the NQ + 2 ticks (disjoint paths: the live ranges of one role do not burden
    // the others) with the same barriers: one per tick, plus those of the surface between the two sweeps.
    if (chain) {
        for (int q = 0; q < NQ + 2; ++q) {
            if (q == NCH) {
                // surface: the upward sweep starts from the reflected (treated) surface row
                if (surface != SOSRT_SURFACE_NONE) {
                    if (valid) s_sfc[tid] = sfc_own;
                    pbarrier();
                }
                if (surface == SOSRT_SURFACE_SPECULAR) {
                    Bv = valid ? rho * s_sfc[N - 1 - tid] : 0.0;                // spec:397
                } else if (surface == SOSRT_SURFACE_LAMBERTIAN) {
                    // -2 rho trapz(In[L-1, rev] mu[rev], mu[rev]), rev = N-2 .. 0   (lam:399), descending abscissae
                    double term = 0;
                    if (tid <= N - 3) {
                        const int k0 = N - 2 - tid, k1 = k0 - 1;
                        const double x0 = g.mu[k0], x1 = g.mu[k1];
                        term = (x1 - x0) * (s_sfc[k1] * x1 + s_sfc[k0] * x0) / 2;
                    }
                    const double ws = wave_sum_(term);
                    pbarrier();
                    if (lane == 0) s_red[tid >> 6] = ws;
                    pbarrier();
                    double S = 0;
                    for (int i = 0; i < nwc; ++i) S += s_red[i];
                    Bv = -2 * rho * S;
                }
                U = Bv;
            }
            if (q < NQ) {   // one chunk of the chain: raw rows into s_D, the recurrence state carried in registers
                const bool up = q >= NCH;
                const double* sp = ring + (size_t)(q % NS) * SLOT + tid;
                double* dD = s_D + (size_t)(q % 3) * TC * RS + tid;
                double Jc[TC], Ec[TC];
        #pragma unroll
                for (int u = 0; u < TC; ++u) {
                    Jc[u] = sp[(0 * TC + u) * RS];
                    Ec[u] = sp[(1 * TC + u) * RS];
                }
                const bool sp_chunk = special(q);
                if (!up) {
                    const int t0 = q * TC;
                    double cc[TC], Sc[TC];
        #pragma unroll
                    for (int u = 0; u < TC; ++u) Sc[u] = 0;
                    if (has_small) {
        #pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            const double sv = s_S[sbase + min(t0 + u, L - 1)];
                            Sc[u] = small ? sv : 0.0;
                        }
                    }
        #pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        const int t = sp_chunk ? min(t0 + u, L - 1) : t0 + u;
                        const double hk = s_hd[t];
                        const double Jp = u == 0 ? Jprev : Jc[u - 1];
                        cc[u] = rec_src(rec_hr(hk, nrmu), Jp, Ec[u], Jc[u]);
                    }
                    if (!sp_chunk) {
        #pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            Dv = rec_step(Dv, Ec[u], cc[u]);
                            double v = Dv;
                            if (has_small) v = rec_add(v, Sc[u]);
                            dD[u * RS] = v;
                        }
                    } else {
        #pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            const int t = t0 + u;
                            if (wl && (t == zbeg1 || t == zbeg2)) load_fix(t == zbeg1 ? 1 : 2);
                            const double Dn = rec_step(Dv, Ec[u], cc[u]);
                            double x = has_small ? rec_add(Dn, Sc[u]) : Dn;
                            dD[u * RS] = x;                                    // raw: the treatment wave rewrites the extrapolated lanes
                            if (wl && nfx > 0) {
                                double acc = 0;
        #pragma unroll
                                for (int k = 0; k < kFixMaxSrc; ++k) acc = fix_acc(c[k], readlane_f64(x, sl[k]), acc);
                                x = fixlane ? acc : x;
                            }
                            const bool zone_end = t == zend0 || t == zend1;    // the next zone starts from the final row (spec:359,378)
                            Dv = t < L ? (zone_end ? (stdl ? x : 0.0) : Dn) : Dv;
                            if (t == L - 1) {
                                sfc_own = x; rdn_v = x; rdn_i = Itot_sfc + x;
                            }
                        }
                    }
                    Jprev = Jc[TC - 1];
                } else {
                    const int t0 = L - 1 - (q - NCH) * TC;
                    double cc[TC];
        #pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        const int t = sp_chunk ? max(t0 - u, 0) : t0 - u;
                        const double hk = s_hd[t + 1];
                        const double Jx = u == 0 ? Jnext : Jc[u - 1];
                        const double src = rec_src(rec_hr(hk, prmu), Jx, Ec[u], Jc[u]);
                        // first row of a zone: attenuate the boundary only (spec:413-419,433-439)
                        cc[u] = (sp_chunk && (t == zend0 || t == zend1)) ? 0.0 : src;
                    }
                    if (!sp_chunk) {
        #pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            U = rec_step(U, Ec[u], cc[u]);
                            dD[u * RS] = tid == 0 ? Jc[u] : U;                 // spec:401
                        }
                    } else {
        #pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            const int t = t0 - u;
                            const double Un = rec_step(U, Ec[u], cc[u]);
                            const double raw = tid == 0 ? Jc[u] : Un;
                            dD[u * RS] = raw;
                            double x = Un;
                            if (w0 && t >= 0) x = blend(raw);
                            const bool zone_start = t == zbeg1 || t == zbeg2;  // blended row feeds the zone above
                            U = t >= 0 ? ((zone_start && tr_up) ? x : Un) : U;
                            if (t == 0) {
                                rup_v = x; rup_i = Itot_top + x;
                            }
                        }
                    }
                    Jnext = Jc[TC - 1];
                }
            }
            TICK_BARRIER();
        }
    } else if (loader) {
        for (int q = 0; q < NQ + 2; ++q) {
            if (q == NCH)
                for (int i = 0; i < seam_barriers; ++i) pbarrier();
            if (q < NQ) {
                if (q + R < NQ) issue(q + R);
                if (q + 1 < NQ) {                              // chunk q+1 must have landed; the younger ones may stay in flight
                    wait_outstanding(max(min(q + R, NQ - 1) - (q + 1), 0) * CNT);
                }
            }
            TICK_BARRIER();
        }
    } else if (treat) {
        for (int q = 0; q < NQ + 2; ++q) {
            if (q == NCH)
                for (int i = 0; i < seam_barriers; ++i) pbarrier();
            if (q >= 1 && q <= NQ) {
                const int qt = q - 1;
                const bool up = qt >= NCH;
                double* row = s_D + (size_t)(qt % 3) * TC * RS + uT * RS;
                const int t = up ? first_row(qt) - uT : first_row(qt) + uT;
                const bool rv = t >= 0 && t < L;
                if (!up) {
                    // In_limit:113-141 as a linear map of the source lanes: rewritten direction N-1-i = sum_k C[i][k] x[s0 + k]
                    const int zz = zone_of(rv ? t : 0);
                    const int nf = rv ? (zz == 0 ? nfix0 : (zz == 1 ? nfix1 : nfix2)) : 0;
                    const int nfmax = max(nfix0, max(nfix1, nfix2));
                    const FixTab& ft = s_fix[zz];
                    const int ns = nf < 2 ? 2 : (nf < kFixMaxSrc ? nf : kFixMaxSrc);
                    const int s0 = nf < 2 ? N - nf - 2 : N - nf - ns;
                    double xs[kFixMaxSrc];
        #pragma unroll
                    for (int k = 0; k < kFixMaxSrc; ++k) xs[k] = row[max(s0 + min(k, ns - 1), 0)];
                    for (int i0 = 0; i0 < nfmax; i0 += 8) {
                        const int i = i0 + pT;
                        double acc = 0;
        #pragma unroll
                        for (int k = 0; k < kFixMaxSrc; ++k) {
                            const double ck = (i < nf && k < ns) ? ft.C[i * ns + min(k, ns - 1)] : 0.0;
                            acc = fix_acc(ck, xs[k], acc);
                        }
                        if (i < nf) row[N - 1 - i] = acc;
                    }
                } else {
                    // spec:401-409: first k >= 1 whose second difference passes, then lanes 1 .. k are blended between lane 0 and lane k+1
                    int kf = rv ? 0 : 1;                              // 0: not found yet
                    bool missing = false;
                    for (int k0 = 1; k0 <= N - 3; k0 += 8) {
                        const int k = k0 + pT;
                        const int kc = min(k, N - 3);
                        const double xa = row[kc], xb = row[kc + 1], xc = row[kc + 2];
                        const bool stop = kf == 0 && k <= N - 3 && !(fabs((xa - xb) - (xb - xc)) > 0.0001);
                        const unsigned long long mk = __ballot(stop);
                        const unsigned bits = (unsigned)(mk >> (8 * uT)) & 0xffu;
                        if (kf == 0 && bits) kf = k0 + (__ffs((int)bits) - 1) + 1;
                        if (__ballot(kf == 0) == 0) break;
                    }
                    if (kf == 0) { missing = true; kf = 1; }           // the reference raises IndexError (spec:404)
                    if (__ballot(missing) != 0 && lane == 0) s_flag[1] = 1;
                    const double r0 = row[0], rk = row[min(kf, N - 1)];
                    const double pk = s_prmu[min(kf, N - 1)];
                    for (int m0 = 1; __ballot(rv && m0 < kf) != 0; m0 += 8) {
                        const int m = m0 + pT;
                        if (rv && m < kf) {
                            const double w = blend_weight(s_mup[m], pk);           // mu_m / mu_kf
                            row[m] = blend_val(w, r0, rk);
                        }
                    }
                }
            }
            TICK_BARRIER();
        }
    } else {
        // Store waves: wave kw takes rows 2 kw, 2 kw + 1 of a chunk, lane p the directions 2p, 2p + 1 (16 bytes per lane
        // and instruction: the time of a tick is the number of vector-memory instructions the CU has to issue, not
        // their bytes).  The rows of the running total come straight from memory into registers, PD ticks ahead of
        // their use (a tick is far shorter than the memory latency): PD register sets, the tick loop unrolled by PD.
        const __amdgpu_buffer_rsrc_t rIn = make_rsrc(a.In + (size_t)b * L * D, fbytes);
        const __amdgpu_buffer_rsrc_t rI = make_rsrc(ACC ? a.I + (size_t)b * L * D : a.In, ACC ? fbytes : 0);
        const __amdgpu_buffer_rsrc_t rS = make_rsrc(SAVED ? a.saved + (size_t)b * a.saved_col_stride : a.In, SAVED ? fbytes : 0);
        constexpr int PD = 6, KR = 2;
        const int nkw = 4;                                           // store waves (TC / KR)
        const int krow = (kw >= 0 ? kw : 0) * KR;
        const bool kval = kw >= 0 && kw < nkw && 2 * lane < N;
        u32x4 Ipre[PD][KR];                                         // raw, as loaded: repacking would wait for the load at once
#pragma unroll
        for (int sI = 0; sI < PD; ++sI)
#pragma unroll
            for (int r = 0; r < KR; ++r) Ipre[sI][r] = (u32x4){0u, 0u, 0u, 0u};
        const int pl = min(2 * lane, N - 2);
        const int kvo_dn = pl * 8, kvo_up = (N + pl) * 8;
        auto fetch = [&](u32x4 (&buf)[KR], int c) {                  // rows of chunk c for this wave
            const bool up = c >= NCH;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int u = krow + r;
                const int t = up ? max(first_row(c) - u, 0) : min(first_row(c) + u, L - 1);
                buf[r] = __builtin_amdgcn_raw_buffer_load_b128(rI, up ? kvo_up : kvo_dn, t * RB, 0);
            }
        };
        if (ACC) {
#pragma unroll
            for (int c = 0; c < PD - 2; ++c)
                if (c < NQ) fetch(Ipre[(c + 2) % PD], c);            // chunk c is stored at tick c + 2
        }
        for (int q0 = 0; q0 < NQ + 2; q0 += PD) {
#pragma unroll
            for (int sI = 0; sI < PD; ++sI) {
                const int q = q0 + sI;
                if (q < NQ + 2) {
                    if (q == NCH)
                        for (int i = 0; i < seam_barriers; ++i) pbarrier();
                    if (q >= 2) {
                        const int qs = q - 2;
                        const bool up = qs >= NCH;
                        const double* dD = s_D + (size_t)(qs % 3) * TC * RS + pl;
                        const int vo = up ? kvo_up : kvo_dn;
#pragma unroll
                        for (int r = 0; r < KR; ++r) {
                            const int u = krow + r;
                            const int t = up ? first_row(qs) - u : first_row(qs) + u;
                            const double2 v = *reinterpret_cast<const double2*>(dD + u * RS);
                            if (kval && t >= 0 && t < L) {
                                const int so = t * RB;
                                bstore2(rIn, vo, so, v);
                                if (ACC) {
                                    const u32x4 w = Ipre[sI][r];
                                    bstore2(rI, vo, so, make_double2(__hiloint2double((int)w.y, (int)w.x) + v.x,
                                                                     __hiloint2double((int)w.w, (int)w.z) + v.y));
                                }
                                if (SAVED) bstore2(rS, vo, so, v);
                            }
                        }
                    }
                    if (ACC && q - 2 + PD < NQ) fetch(Ipre[sI], q - 2 + PD);
                    TICK_BARRIER();
                }
            }
        }
    }
#undef TICK_BARRIER
    if (a.stamps && lane == 0 && wid < 15) {
        a.stamps[(size_t)b * 16 + wid] = busy;
        if (wid == 0) a.stamps[(size_t)b * 16 + 15] = clock64() - t_begin;
    }
    if (loader) pwait_vm<0>();
    if (chain && w0 && notfound && lane == 0) s_flag[N - 3 <= 61 ? 1 : 0] = 1;
    __syncthreads();
    if (s_flag[1]) {                                                    // the reference raises IndexError (spec:404)
        if (tid == 0) {
            a.cv.status[b] = SOSRT_COL_INDEXERROR;
            if (ACC) { a.cv.active[b] = 0; a.cv.norders[b] = a.order; atomicSub(a.cv.nactive, 1); }
        }
        return;
    }
    if (s_flag[0]) {                                                    // let the general kernel redo the upward sweep
        if (tid == 0) a.cv.redo[b] = 1;
        return;
    }
    if (ACC) {
        const bool cv_lane = chain && valid;
        const double ra = block_pymax_(rup_v / rup_i, cv_lane, s_red, 0);
        const double rb = block_pymax_(rdn_v / rdn_i, cv_lane, s_red, 0);
        const double r = (rb > ra) ? rb : ra;                               // the outer max() of spec:309
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

#undef first_row
#undef special
#undef zone_of

inline size_t pipe_extra_doubles(const Grid& g) {
    return (size_t)3 * TC * RS + 128 + 16 + g.L + 1 + 128 + 128 + (size_t)g.nsmall * g.L;
}

}  // namespace

int g_pipe_slots = 6;      // ring depth of the pipeline kernel (SOSRT_PIPE_SLOTS): a tick is far shorter than the memory latency
int g_pipe_max = 256;      // at or below this many live columns the ring kernel hands over to the pipeline kernel (SOSRT_PIPE_MAX; 0: never)

bool transport_pipe_ok(const Grid& g) {
    if (g.N % 2 || g.N < 4 || g.N > 128) return false;
    return (3 * (size_t)2 * TC * RS + pipe_extra_doubles(g)) * sizeof(double) <= kPipeLdsBytes;
}

void launch_transport_pipe(hipStream_t s, dim3 grid, const TransportArgs& a, int slots) {
    const int nwc = (a.g.N + 63) / 64;
    const dim3 block((nwc + PIPE_NWL + 1 + 4) * 64);
    const size_t slot_bytes = (size_t)2 * TC * RS * sizeof(double);
    const size_t extra = pipe_extra_doubles(a.g) * sizeof(double);
    int NS = slots < 2 ? 2 : slots;
    while (NS > 2 && NS * slot_bytes + extra > kPipeLdsBytes) --NS;
    const size_t shm = NS * slot_bytes + extra;
#define SOSRT_PIPE_LAUNCH(ACC_, SAVED_)                                                                        \
    do {                                                                                                       \
        auto kern = k_transport_pipe<ACC_, SAVED_>;                                                            \
        static bool big_lds = false;                                                                           \
        if (!big_lds) {                                                                                        \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)kPipeLdsBytes);                                                           \
            big_lds = true;                                                                                    \
        }                                                                                                      \
        hipLaunchKernelGGL(kern, grid, block, shm, s, a, NS);                                                  \
    } while (0)
    if (a.accumulate) {
        if (a.saved) SOSRT_PIPE_LAUNCH(true, true);
        else SOSRT_PIPE_LAUNCH(true, false);
    } else {
        SOSRT_PIPE_LAUNCH(false, false);
    }
#undef SOSRT_PIPE_LAUNCH
}

}  // namespace sosrt
