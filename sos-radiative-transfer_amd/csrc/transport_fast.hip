// k_transport_fast: one order of transport (spec:326-449) with wave-independent sweeps.
//
// Thread j carries the downward direction m = j and then the upward direction m = N+j.  The two
// mu -> 0 neighbourhoods are each confined to one wave: the extrapolated downward angles and their
// source angles (In_limit:113-141) are the last lanes of the last wave, the upward
// second-difference search (spec:403-406) almost always ends within the first lanes of wave 0.
// Those two waves resolve the treatments with cross-lane reads (DPP shifts, v_readlane); every
// other wave is a pure recurrence
//     D_t = E D_{t-1} - (dl/2)(J_{t-1} E + J_t)/mu        U_t = E U_{t+1} + (dl/2)(J_t + J_{t+1} E)/mu
// with E from the attenuation table.  There is no LDS exchange and no barrier inside the sweeps.
//
// Rows go in chunks of TC.  A chunk is "plain" unless it touches a zone boundary, the first row
// of a sweep or the ragged end: plain chunks run a branch-free body (about 20 instructions per
// row and lane: 3 loads, 2 stores, 6 fp64 ops), the few others a general body with the zone
// restarts (spec:359,378).  Three register buffers rotate through the roles
// {being computed, next, after next}; the rotation is unrolled so that no register of a load in
// flight is ever copied, and every fetch issues the same number of loads so that the compiler's
// s_waitcnt vmcnt(N) leaves the younger chunks in flight.
//
// If a search does not end within wave 0 the column is flagged (cv.redo) and
// k_transport<.., REPAIR> redoes its upward sweep with the general exchange.
#include <type_traits>

#include "../../include/sosrt.h"
#include "kernels.hpp"
#include "transport_util.hpp"

namespace sosrt {

namespace {

template <int MAXT, bool ACC, bool SAVED, bool FULL>
__global__ __launch_bounds__(MAXT) void k_transport_fast(TransportArgs a) {
    const int b = blockIdx.x;
    if (ACC && !a.cv.active[b]) return;
    const Grid& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    // wave 0, as a value the compiler knows to be wave-uniform (anything derived from threadIdx is
    // divergent to it and would turn every wave-0 block into an exec-masked region)
    const int wid = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const bool w0 = wid == 0;
    const bool wl = wid == ((g.N - 1) >> 6);                   // the wave that holds the mu -> 0- lanes
    const int L = g.L, N = g.N, D = g.D;
    auto stamp = [&](int i) {
        if (a.stamps && (tid == 0 || tid == 64)) a.stamps[((size_t)b * 2 + (tid >> 6)) * 8 + i] = clock64();
    };
    stamp(0);
    extern __shared__ double sm[];
    double* s_sfc = sm;                      // [blockDim] surface row by downward lane m (Lambertian only)
    double* s_red = s_sfc + blockDim.x;      // [nw + 1]
    __shared__ FixTab s_fix[kRingZones];
    __shared__ int s_flag[2];                // [0] redo with the general kernel, [1] IndexError
    const ColDesc* __restrict__ dg = a.desc + b;     // uniform address: scalar loads
    const int nz = dg->nz;
    const int zend0 = nz > 1 ? dg->r1[0] : -9, zend1 = nz > 2 ? dg->r1[1] : -9;   // last rows of the non-bottom zones
    const int zbeg1 = nz > 1 ? dg->r0[1] : -9, zbeg2 = nz > 2 ? dg->r0[2] : -9;   // first rows of the non-top zones
    const int nfix0 = dg->nfix[0], nfix1 = dg->nfix[1], nfix2 = dg->nfix[2];
    const int surface = dg->surface;
    const double rho = dg->rho;
    for (int zz = 0; zz < kRingZones; ++zz) {
        const double* src = reinterpret_cast<const double*>(&g.fix[dg->fixtab[zz]]);
        double* dst = reinterpret_cast<double*>(&s_fix[zz]);
        for (int i = tid; i < (int)(sizeof(FixTab) / sizeof(double)); i += blockDim.x) dst[i] = src[i];
    }
    if (tid < 2) s_flag[tid] = 0;
    const double* __restrict__ tau = a.tau + (size_t)b * L;
    const int fbytes = L * D * 8, RB = D * 8;                  // bytes of a field of this column / of a row
    const __amdgpu_buffer_rsrc_t rJ = make_rsrc(a.Jn + (size_t)b * L * D, fbytes);
    const __amdgpu_buffer_rsrc_t rE = make_rsrc(a.Etab + (size_t)(a.erep ? a.erep[b] : b) * L * D, fbytes);
    const __amdgpu_buffer_rsrc_t rIn = make_rsrc(a.In + (size_t)b * L * D, fbytes);
    const __amdgpu_buffer_rsrc_t rI = make_rsrc(ACC ? a.I + (size_t)b * L * D : a.In, ACC ? fbytes : 0);
    const __amdgpu_buffer_rsrc_t rS = make_rsrc(SAVED ? a.saved + (size_t)b * a.saved_col_stride : a.In, SAVED ? fbytes : 0);
    __syncthreads();

    const bool valid = FULL || tid < N;
    const int tidc = valid ? tid : N - 1;
    // a chunk [lo, hi] of rows touches a zone boundary
    auto touches = [&](int lo, int hi) {
        return (zend0 >= lo && zend0 <= hi) || (zend1 >= lo && zend1 <= hi) || (zbeg1 >= lo && zbeg1 <= hi) ||
               (zbeg2 >= lo && zbeg2 <= hi);
    };
    double rdn_v = 0, rdn_i = 1, rup_v = 0, rup_i = 1;
    double sfc_own = 0;                      // In[L-1][tid]
    stamp(1);

    // =============================== downward ===============================
    {
        const int m = tidc;
        const int vo = m * 8;                                       // lane byte offset inside a row
        const double mu = g.mu[m];
        const bool tr = valid && m <= N - 2;
        const bool small = tr && fabs(mu) < kMuThreshold;       // spec:333
        const bool stdl = tr && !small;
        const double nrmu = stdl ? -1.0 / mu : 0.0;             // 0 turns the recurrence off for the other lanes
        const bool has_small = wl && g.nsmall > 0;              // the small-mu lanes are the last lanes of the last wave
        double Dv = 0, Jprev = 0;
        double c[kFixMaxSrc] = {0, 0, 0, 0, 0};
        int sl[kFixMaxSrc] = {0, 0, 0, 0, 0};                   // source lanes of the extrapolation (uniform)
        int nfx = 0;
        bool fixlane = false;
        auto load_fix = [&](int zz) {                          // direction N-1-i owns row i of the table
            const FixTab& ft = s_fix[zz];
            nfx = zz == 0 ? nfix0 : (zz == 1 ? nfix1 : nfix2);
            const int ns = nfx < 2 ? 2 : (nfx < kFixMaxSrc ? nfx : kFixMaxSrc);     // In_limit:118-141
            fixlane = valid && nfx > 0 && m >= N - nfx;
            const int i = fixlane ? N - 1 - m : 0;
            const int s0 = nfx < 2 ? N - nfx - 2 : N - nfx - ns;                     // first source direction
#pragma unroll
            for (int q = 0; q < kFixMaxSrc; ++q) {
                c[q] = (fixlane && q < ns) ? ft.C[i * ns + min(q, ns - 1)] : 0.0;
                sl[q] = (s0 + min(q, ns - 1)) & 63;             // lane of source q inside the last wave
            }
        };
        if (wl) load_fix(0);
        double J0[TC], I0[TC], E0[TC], S0[TC], J1[TC], I1[TC], E1[TC], S1[TC], J2[TC], I2[TC], E2[TC], S2[TC];
#pragma unroll
        for (int u = 0; u < TC; ++u) { S0[u] = S1[u] = S2[u] = 0; I0[u] = I1[u] = I2[u] = 0; }
        int tp = 0;
        auto fetch = [&](double (&Jx)[TC], double (&Ix)[TC], double (&Ex)[TC], double (&Sx)[TC]) {
            const int tq = min(tp, L - 1);
            if (small) {                                            // values written by k_smallmu
#pragma unroll
                for (int u = 0; u < TC; ++u) Sx[u] = bload(rIn, vo, min(tq + u, L - 1) * RB);
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int so = min(tq + u, L - 1) * RB;
                Jx[u] = bload(rJ, vo, so);
                if (ACC) Ix[u] = bload(rI, vo, so);
                Ex[u] = bload(rE, vo, so);
            }
            tp += TC;
        };
        int t0 = 0;
        auto process = [&](auto special_t, double (&Jc)[TC], double (&Ic)[TC], double (&Ec)[TC], double (&Sc)[TC]) {
            constexpr bool SP = decltype(special_t)::value;
            double cc[TC], v[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) {                         // independent: source term of every row
                const int t = SP ? min(t0 + u, L - 1) : t0 + u;
                const double hk = (tau[t] - tau[SP ? max(t - 1, 0) : t - 1]) * 0.5;   // uniform: scalar unit
                const double Jp = u == 0 ? Jprev : Jc[u - 1];
                cc[u] = (hk * nrmu) * (Jp * Ec[u] + Jc[u]);
            }
            if (!SP) {
#pragma unroll
                for (int u = 0; u < TC; ++u) {                     // sequential: the recurrence
                    Dv = Dv * Ec[u] + cc[u];
                    v[u] = has_small ? Dv + Sc[u] : Dv;
                }
                if (wl && nfx > 0) {                               // In_limit:113-141 as a linear map of the source lanes
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        double acc = 0;
#pragma unroll
                        for (int q = 0; q < kFixMaxSrc; ++q) acc += c[q] * readlane_f64(v[u], sl[q]);
                        v[u] = fixlane ? acc : v[u];
                    }
                }
                if (valid) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        const int so = (t0 + u) * RB;
                        bstore(rIn, vo, so, v[u]);
                        if (ACC) bstore(rI, vo, so, Ic[u] + v[u]);
                        if (SAVED) bstore(rS, vo, so, v[u]);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const int t = t0 + u;
                    if (wl && (t == zbeg1 || t == zbeg2)) load_fix(t == zbeg1 ? 1 : 2);
                    const double Dn = Dv * Ec[u] + cc[u];
                    double x = has_small ? Dn + Sc[u] : Dn;
                    if (wl && nfx > 0) {
                        double acc = 0;
#pragma unroll
                        for (int q = 0; q < kFixMaxSrc; ++q) acc += c[q] * readlane_f64(x, sl[q]);
                        x = fixlane ? acc : x;
                    }
                    v[u] = x;
                    const bool zone_end = t == zend0 || t == zend1;     // the next zone starts from the final row (spec:359,378)
                    Dv = t < L ? (zone_end ? (stdl ? x : 0.0) : Dn) : Dv;
                    if (valid && t < L) {
                        const int so = t * RB;
                        bstore(rIn, vo, so, x);
                        if (ACC) bstore(rI, vo, so, Ic[u] + x);
                        if (SAVED) bstore(rS, vo, so, x);
                    }
                }
            }
            if (t0 + TC >= L) {
#pragma unroll
                for (int u = 0; u < TC; ++u)
                    if (t0 + u == L - 1) { sfc_own = v[u]; rdn_v = v[u]; rdn_i = Ic[u] + v[u]; }
            }
            Jprev = Jc[TC - 1];
            t0 += TC;
        };
        auto step = [&](double (&Jc)[TC], double (&Ic)[TC], double (&Ec)[TC], double (&Sc)[TC]) {
            if (t0 == 0 || t0 + TC > L || touches(t0, t0 + TC - 1)) process(std::true_type{}, Jc, Ic, Ec, Sc);
            else process(std::false_type{}, Jc, Ic, Ec, Sc);
        };
        fetch(J0, I0, E0, S0);
        fetch(J1, I1, E1, S1);
        for (;;) {
            if (t0 >= L) break;
            fetch(J2, I2, E2, S2); step(J0, I0, E0, S0);
            if (t0 >= L) break;
            fetch(J0, I0, E0, S0); step(J1, I1, E1, S1);
            if (t0 >= L) break;
            fetch(J1, I1, E1, S1); step(J2, I2, E2, S2);
        }
    }

    stamp(2);
    // =============================== surface ===============================
    double Bv = 0;
    if (surface != SOSRT_SURFACE_NONE) {
        if (valid) s_sfc[tid] = sfc_own;
        __syncthreads();
    }
    if (surface == SOSRT_SURFACE_SPECULAR) {
        Bv = valid ? rho * s_sfc[N - 1 - tid] : 0.0;                // spec:397
    } else if (surface == SOSRT_SURFACE_LAMBERTIAN || surface == SOSRT_SURFACE_LAMBERTIAN_README) {
        // -2 rho trapz(In[L-1, rev] mu[rev], mu[rev]), rev = N-2 .. 0   (lam:399), descending abscissae
        double term = 0;
        if (tid <= N - 3) {
            const int k0 = N - 2 - tid, k1 = k0 - 1;
            const double x0 = g.mu[k0], x1 = g.mu[k1];
            term = (x1 - x0) * (s_sfc[k1] * x1 + s_sfc[k0] * x0) / 2;
        }
        double ws = wave_sum_(term);
        __syncthreads();
        if (lane == 0) s_red[tid >> 6] = ws;
        __syncthreads();
        double S = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) S += s_red[i];
        Bv = (surface == SOSRT_SURFACE_LAMBERTIAN_README ? 2 : -2) * rho * S;          // lam:399 as coded (negative), or README.md:215
    }

    stamp(3);
    // =============================== upward ===============================
    {
        const int mj = N + tidc;
        const int vo = mj * 8;
        const bool tr = valid && tid > 0;
        const double mu = tr ? g.mu[mj] : 1.0;
        const double prmu = tr ? 1.0 / mu : 0.0;
        const int last_cand = min(N - 3, 61);                      // lanes whose two right neighbours are in wave 0
        bool notfound = false;                                     // wave-uniform
        double U = Bv, Jnext = 0;
        double J0[TC], I0[TC], E0[TC], J1[TC], I1[TC], E1[TC], J2[TC], I2[TC], E2[TC];
#pragma unroll
        for (int u = 0; u < TC; ++u) { I0[u] = I1[u] = I2[u] = 0; }
        int tp = L - 1;
        auto fetch = [&](double (&Jx)[TC], double (&Ix)[TC], double (&Ex)[TC]) {
            const int tq = max(tp, 0);
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int so = max(tq - u, 0) * RB;
                Jx[u] = bload(rJ, vo, so);
                if (ACC) Ix[u] = bload(rI, vo, so);
                Ex[u] = bload(rE, vo, so);
            }
            tp -= TC;
        };
        // spec:401-409 for one row held across wave 0: x is the raw row, returns the blended value
        auto blend = [&](double x) {
            const double x1 = lane_up1(x), x2 = lane_up1(x1);
            const bool stop = lane >= 1 && lane <= last_cand && !(fabs((x - x1) - (x1 - x2)) > 0.0001);
            const unsigned long long mk = __ballot(stop);
            // ks + 1; an empty mask is recorded once per sweep (IndexError in the reference, or the search
            // leaves the wave) and the column is redone / flagged after the sweep
            const int kf = mk ? __ffsll((long long)mk) : 1;
            notfound |= (mk == 0);
            const double r0 = readlane_f64(x, 0), rk = readlane_f64(x, kf);
            const double w = mu * readlane_f64(prmu, kf);              // mu_m / mu_kf to one rounding
            const double bl = (1 - w) * r0 + w * rk;
            return (tr && tid < kf) ? bl : x;
        };
        int t0 = L - 1;
        auto process = [&](auto special_t, double (&Jc)[TC], double (&Ic)[TC], double (&Ec)[TC]) {
            constexpr bool SP = decltype(special_t)::value;
            double cc[TC], v[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int t = SP ? max(t0 - u, 0) : t0 - u;
                const double hk = (tau[SP ? min(t + 1, L - 1) : t + 1] - tau[t]) * 0.5;
                const double Jx = u == 0 ? Jnext : Jc[u - 1];
                const double src = (hk * prmu) * (Jc[u] + Jx * Ec[u]);
                // first row of a zone: attenuate the boundary only (spec:413-419,433-439)
                cc[u] = (SP && (t == zend0 || t == zend1)) ? 0.0 : src;
            }
            if (!SP) {
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    U = U * Ec[u] + cc[u];
                    v[u] = U;
                }
                if (w0) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) v[u] = blend(tid == 0 ? Jc[u] : v[u]);   // spec:401
                }
                if (valid) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        const int so = (t0 - u) * RB;
                        bstore(rIn, vo, so, v[u]);
                        if (ACC) bstore(rI, vo, so, Ic[u] + v[u]);
                        if (SAVED) bstore(rS, vo, so, v[u]);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const int t = t0 - u;
                    const double Un = U * Ec[u] + cc[u];
                    double x = Un;
                    if (w0 && t >= 0) x = blend(tid == 0 ? Jc[u] : Un);
                    v[u] = x;
                    const bool zone_start = t == zbeg1 || t == zbeg2;   // blended row feeds the zone above
                    U = t >= 0 ? ((zone_start && tr) ? x : Un) : U;
                    if (valid && t >= 0) {
                        const int so = t * RB;
                        bstore(rIn, vo, so, x);
                        if (ACC) bstore(rI, vo, so, Ic[u] + x);
                        if (SAVED) bstore(rS, vo, so, x);
                    }
                }
            }
            if (t0 - TC < 0) {
#pragma unroll
                for (int u = 0; u < TC; ++u)
                    if (t0 - u == 0) { rup_v = v[u]; rup_i = Ic[u] + v[u]; }
            }
            Jnext = Jc[TC - 1];
            t0 -= TC;
        };
        auto step = [&](double (&Jc)[TC], double (&Ic)[TC], double (&Ec)[TC]) {
            if (t0 == L - 1 || t0 - TC < -1 || touches(t0 - TC + 1, t0)) process(std::true_type{}, Jc, Ic, Ec);
            else process(std::false_type{}, Jc, Ic, Ec);
        };
        fetch(J0, I0, E0);
        fetch(J1, I1, E1);
        for (;;) {
            if (t0 < 0) break;
            fetch(J2, I2, E2); step(J0, I0, E0);
            if (t0 < 0) break;
            fetch(J0, I0, E0); step(J1, I1, E1);
            if (t0 < 0) break;
            fetch(J1, I1, E1); step(J2, I2, E2);
        }
        if (w0 && notfound && lane == 0) s_flag[N - 3 <= 61 ? 1 : 0] = 1;
    }
    stamp(4);
    __syncthreads();
    stamp(5);
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
        const double ra = block_pymax_(rup_v / rup_i, valid, s_red, 0);
        const double rb = block_pymax_(rdn_v / rdn_i, valid, s_red, 0);
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

template <int MAXT, bool FULL>
void launch_t(hipStream_t s, dim3 grid, dim3 block, size_t shm, const TransportArgs& a) {
    if (a.accumulate) {
        if (a.saved) hipLaunchKernelGGL((k_transport_fast<MAXT, true, true, FULL>), grid, block, shm, s, a);
        else hipLaunchKernelGGL((k_transport_fast<MAXT, true, false, FULL>), grid, block, shm, s, a);
    } else {
        hipLaunchKernelGGL((k_transport_fast<MAXT, false, false, FULL>), grid, block, shm, s, a);
    }
}

}  // namespace

void launch_transport_fast(hipStream_t s, dim3 grid, dim3 block, const TransportArgs& a) {
    const int nt = (int)block.x;
    const size_t shm = (size_t)(nt + nt / 64 + 2) * sizeof(double);
    const bool full = a.g.N == nt;
    if (nt <= 128) { if (full) launch_t<128, true>(s, grid, block, shm, a); else launch_t<128, false>(s, grid, block, shm, a); }
    else if (nt <= 256) { if (full) launch_t<256, true>(s, grid, block, shm, a); else launch_t<256, false>(s, grid, block, shm, a); }
    else if (nt <= 512) { if (full) launch_t<512, true>(s, grid, block, shm, a); else launch_t<512, false>(s, grid, block, shm, a); }
    else { if (full) launch_t<1024, true>(s, grid, block, shm, a); else launch_t<1024, false>(s, grid, block, shm, a); }
}

}  // namespace sosrt
