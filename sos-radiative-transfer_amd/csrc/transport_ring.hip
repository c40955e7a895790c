// k_transport_ring: one order of transport (spec:326-449), rows streamed through an LDS ring.
//
// Same arithmetic and the same lane mapping as k_transport_fast (thread j = downward direction j,
// then upward direction N+j; the mu -> 0 treatments live in the last / first wave), but the rows
// of Jn, of the attenuation table and of the running total do not come through registers of the
// computing waves.  A workgroup is  nwc = ceil(N/64) computing waves + NWL loader waves.  The
// loaders copy chunks of TC rows into an LDS ring of NS slots with `buffer_load_dwordx4 ... lds`
// (1 KiB per wave-instruction, no VGPR destination), R = NS-1 chunks ahead of the computing waves;
// the computing waves read a chunk from LDS, run the recurrence and store.
//
// Why: a lone column is latency-bound (a wave may have 63 vector-memory operations outstanding,
// loads and stores together, and with 8-byte lanes that is 32 KiB).  Here the loads in flight are
// R chunks of 24 KiB per column and they do not share the counter with the stores.
//
// Chunk sequence q = 0 .. 2 NCH-1: the NCH chunks of the downward sweep (rows ascending, lanes
// [0, N)), then the NCH chunks of the upward sweep (rows descending, lanes [N, 2N)); the loaders
// run across the seam, so the upward sweep starts with its first chunks already in LDS.
//
// Protocol (one s_barrier per chunk, raw: __syncthreads() would drain the loads in flight):
//   every loader carries 1/NWL of the rows of every chunk.  In iteration q it issues its part of
//   chunk q+R into slot (q+R) mod NS -- the slot read in iteration q-1 -- and waits with a counted
//   vmcnt until only its parts of the younger chunks are outstanding; then the barrier.
//   computing waves in iteration q read slot q mod NS (landed before the barrier that ended
//   iteration q-1), wait lgkmcnt(0) and join the barrier.
#include <type_traits>

#include "../../include/sosrt.h"
#include "kernels.hpp"
#include "transport_util.hpp"

namespace sosrt {

namespace {

// cache policy of the field rows' stores (In, I, saved) -- experiment of round 4 (-DSOSRT_RING_STORE_AUX=17: write-through, so that
// the kernel leaves no dirty lines for the end-of-kernel write-back): default 0, plain stores
#ifndef SOSRT_RING_STORE_AUX
#define SOSRT_RING_STORE_AUX 0
#endif
__device__ __forceinline__ void rstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x) { bstore_aux<SOSRT_RING_STORE_AUX>(r, voff, soff, x); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr size_t kRingLdsBytes = 152 * 1024;               // of the 160 KiB of a CU

template <int CNT>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}
// workgroup barrier that leaves vector-memory operations in flight
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool ACC, bool SAVED, int PIECES, bool MZ>
__global__ __launch_bounds__(512) void k_transport_ring(TransportArgs a, int NS, int dbg_arg, int fixcap) {
    constexpr int NWL = 2;                                     // loader waves
#ifdef SOSRT_RING_DEBUG
    const int dbg = dbg_arg;            // diagnostic builds only (-DSOSRT_RING_DEBUG): timing switches that corrupt the results
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
    int b = blockIdx.x;
    if (ACC && a.live > 0) {
        b = a.live_list[blockIdx.x];
        if (b < 0) return;                          // fewer live columns than the host's (lagging) count
    } else if (ACC && !a.cv.active[b]) {
        return;
    }
    const Grid& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int L = g.L, N = g.N, D = g.D;
    const int nwc = (N + 63) >> 6;                             // computing waves
    const bool loader = wid >= nwc;
    const int lid = wid - nwc;
    const bool w0 = wid == 0 && !(dbg & 8);
    const bool wl = wid == ((N - 1) >> 6) && !(dbg & 4);                     // the wave that holds the mu -> 0- lanes
    auto stamp = [&](int i) {
        if (a.stamps && (tid == 0 || tid == 64)) a.stamps[((size_t)b * 2 + (tid >> 6)) * 8 + i] = clock64();
    };
    stamp(0);
    constexpr int NARR = ACC ? 3 : 2;                          // Jn, attenuation, [running total]
    constexpr int RS = 128 * PIECES;                           // doubles per LDS row (one half row of the field)
    constexpr int SLOT = NARR * TC * RS;                       // doubles per ring slot
    constexpr int IPC = NARR * TC * PIECES;                    // wave-instructions per chunk
    extern __shared__ double sm[];
    double* ring = sm;                                         // [NS][NARR][TC][RS]
    // [max(RS, coefficient tables)]: the extrapolation coefficients of each zone, compact ([zone][fixcap rewritten directions]
    // [5 sources]), during the downward sweep; the surface row by downward lane m from the turn-round on (every wave has
    // passed the barrier of the last downward chunk by then)
    double* s_sfc = ring + (size_t)NS * SLOT;
    double* s_fixc = s_sfc;
    double* s_red = s_sfc + max(RS, (MZ ? a.nzcap : kRingZones) * fixcap * kFixMaxSrc);      // [nw + 1]  (nzcap: most zones of any column of the batch, >= 3)
    double* s_hd = s_red + (blockDim.x >> 6) + 2;              // [L + 1] half layer thicknesses: hd[t] = (tau[t] - tau[t-1]) / 2
    double* s_S = s_hd + L + 1;                                // [nsmall][L] the |mu| < 0.01 lanes as written by k_smallmu
    double* s_x = s_S + (size_t)g.nsmall * L;                  // [TC][16] the 16 lanes of a mu -> 0 neighbourhood, rows of a chunk
    double* s_prmu = s_x + TC * 16;                            // [16] 1/mu of the first upward directions
    int* s_nf = reinterpret_cast<int*>(s_prmu + 16);           // [(L + 31) / 32] rows whose mu -> 0+ search left wave 0 (finish_flagged_rows)
    // lanes (of the wave that holds them) of the two neighbourhoods: the last 16 downward directions, the first 16 upward ones
    const int xb_dn = max(((N - 1) & 63) - 15, 0);
    // (all per-column tables are compact: with the ring and the row buffers a workgroup needs less than half the LDS of a
    // CU, so two columns share one -- or one column and two workgroups of the contraction of the other column group)
    __shared__ int s_flag[3];                                  // [0] redo the upward sweep row by row, [1] IndexError, [2] some row is flagged in s_nf
    const ColDesc* __restrict__ dg = a.desc + b;
    const int nz = dg->nz;
    const ZoneRows<MZ> zr(dg);                                     // zone boundaries in scalars (MZ: all of them, else the reference's two)
    const int nfix0 = dg->nfix[0], nfix1 = dg->nfix[1], nfix2 = dg->nfix[2];
    const int surface = dg->surface;
    const double rho = dg->rho;
    const int fbytes = L * D * 8, RB = D * 8;
    // dbg (timing experiments only, results are wrong): 1 drops the loads, 2 the stores (zero-sized descriptors)
    const int lbytes = (dbg & 1) ? 0 : fbytes, sbytes = (dbg & 2) ? 0 : fbytes;
    const __amdgpu_buffer_rsrc_t rJ = make_rsrc(a.Jn + (size_t)b * L * D, lbytes);
    const __amdgpu_buffer_rsrc_t rE = make_rsrc(a.Etab + (size_t)(a.erep ? a.erep[b] : b) * L * D, lbytes);
    const __amdgpu_buffer_rsrc_t rIn = make_rsrc(a.In + (size_t)b * L * D, sbytes);
    const __amdgpu_buffer_rsrc_t rIl = make_rsrc(ACC ? a.I + (size_t)b * L * D : a.In, ACC ? lbytes : 0);
    const __amdgpu_buffer_rsrc_t rI = make_rsrc(ACC ? a.I + (size_t)b * L * D : a.In, ACC ? sbytes : 0);
    const __amdgpu_buffer_rsrc_t rS = make_rsrc(SAVED ? a.saved + (size_t)b * a.saved_col_stride : a.In, SAVED ? sbytes : 0);
    const int NCH = (L + TC - 1) / TC, NQ = 2 * NCH, R = NS - 1;
    const bool valid = tid < N;
    const int tidc = valid ? tid : N - 1;
    // Chunks with a zone boundary, and the last chunk of each sweep (its final row feeds the surface / the convergence test), go
    // through the general body: one bit per chunk of a sweep.  More than 64 chunks per sweep: every chunk takes the general body.
    unsigned long long sp_dn = NCH > 64 ? ~0ull : 1ull << (NCH - 1), sp_up = sp_dn;
    if (NCH <= 64) {
        sp_dn |= zr.boundary_chunks(L, TC, false);
        sp_up |= zr.boundary_chunks(L, TC, true);
    }
    // The plain chunks between two of those run in a loop of their own.  The general body in the same loop costs the plain
    // chunks a third of their time although they never execute it (register copies where its values join the loop, spilled
    // scalars reloaded in the hot path): a lone column 34 instead of 50 us without it (compiled out, in-kernel stamps).
    auto plain_run = [&](unsigned long long mask, int j) {          // plain chunks from chunk j of the sweep on
        const unsigned long long rest = NCH > 64 ? 1ull : mask >> j;
        return rest ? __builtin_ctzll(rest) : 0;
    };
    double rdn_v = 0, rdn_i = 1, rup_v = 0, rup_i = 1;
    double sfc_own = 0;
    int slot_rd = 0;                                          // slot of the next chunk to read

    // ------------------------------- loader side -------------------------------
    // chunk q -> slot q mod NS; rows clamped at the ragged ends exactly as the computing side expects.
    // Every loader carries its share of every chunk (rows u = lid mod NWL): the loads of a chunk are
    // issued in 1/NWL of the time, and that time is on the critical path of an iteration.
    // (a lone column is bound by the instruction streams of its waves -- a wave alone on its SIMD issues one
    // instruction every 8 to 12 cycles and pays some 30 for a branch, tools/microbench.hip -- so the loader's
    // iteration is straight-line code: rows by arithmetic, the ring slot by a running counter, the wait an immediate)
    int slot_ld = 0;                                          // slot of the next chunk to issue
    auto issue = [&](int q) {
        const bool up = q >= NCH;
        const int j = up ? q - NCH : q;
        double* dst = ring + (size_t)slot_ld * SLOT;
        slot_ld = slot_ld + 1 == NS ? 0 : slot_ld + 1;
        const int half = up ? N * 8 : 0;
        const int vo = lane * 16;
#pragma unroll
        for (int i = 0; i < TC / NWL; ++i) {
            const int u = i * NWL + lid;
            const int row = up ? max(L - 1 - j * TC - u, 0) : min(j * TC + u, L - 1);
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const int so = row * RB + half + p * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (0 * TC + u) * RS + p * 128), 16, vo, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rE, (lds_ptr_t)(dst + (1 * TC + u) * RS + p * 128), 16, vo, so, 0, 0);
                if (ACC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rIl, (lds_ptr_t)(dst + (2 * TC + u) * RS + p * 128), 16, vo, so, 0, 0);
            }
        }
    };
    // KEEP = vector-memory instructions of this wave that may stay in flight once chunk q+1 has landed: the R-1 younger chunks
    auto loader_phase = [&](auto keep_t, int qa, int qb) {
        constexpr int KEEP = decltype(keep_t)::value;
        for (int q = qa; q < qb; ++q) {
            if (q + R < NQ) {
                issue(q + R);
                wait_vm<KEEP>();
            } else {
                wait_vm<0>();                                  // the last R iterations issue nothing: everything has to land
            }
            wg_barrier();
        }
    };
    // Start-up: the loaders put the first R chunks in flight at once; meanwhile the computing waves
    // fill the small per-column tables.  They live in LDS because inside the sweeps a computing wave
    // must never wait for a global load: the wait would also cover its stores (one counter).
    const int ncomp = nwc * 64;
    if (loader) {
        for (int q = 0; q < min(R, NQ); ++q) issue(q);
    } else {
        for (int zz = 0; zz < (MZ ? nz : kRingZones); ++zz) {
            const FixTab& src = g.fix[dg->fixtab[zz]];
            for (int i = tid; i < fixcap * kFixMaxSrc; i += ncomp) s_fixc[zz * fixcap * kFixMaxSrc + i] = src.C[i];
        }
        if (tid < 3) s_flag[tid] = 0;
        for (int i = tid; i < (L + 31) / 32; i += ncomp) s_nf[i] = 0;
        const double* __restrict__ tau = a.tau + (size_t)b * L;
        for (int t = tid; t <= L; t += ncomp) s_hd[t] = (t == 0 || t == L) ? 0.0 : (tau[t] - tau[t - 1]) * 0.5;
        if (tid < 16) s_prmu[tid] = (tid > 0 && tid < N) ? 1.0 / g.mu[N + tid] : 0.0;
        const double* __restrict__ In0 = a.In + (size_t)b * L * D;
        for (int i = tid; i < g.nsmall * L; i += ncomp) {
            const int k = i / L, t = i - k * L;
            s_S[i] = In0[(size_t)t * D + g.small_lanes[k]];
        }
    }
    __syncthreads();                                            // (drains the loaders once: chunks 0 .. R-1 have landed)

    // The loader waves run their whole stream in this one branch and meet the computing waves at every
    // barrier; keeping the two roles on disjoint paths also keeps the compiler from ordering the
    // computing waves' LDS reads (and with them their stores) behind loads they never issue.
    const int seam_barriers = surface == SOSRT_SURFACE_NONE ? 0 : (surface == SOSRT_SURFACE_SPECULAR ? 1 : 3);
    double Bv = 0;
    if (loader) {
        constexpr int PER = IPC / NWL;                          // instructions of one loader per chunk
        auto run = [&](auto keep_t) {
            loader_phase(keep_t, 0, NCH);
            for (int i = 0; i < seam_barriers; ++i) wg_barrier();
            loader_phase(keep_t, NCH, NQ);
            wait_vm<0>();
        };
        if (R <= 1) run(std::integral_constant<int, 0>{});
        else if (R == 2) run(std::integral_constant<int, (PER < 60 ? PER : 60)>{});
        else if (R == 3) run(std::integral_constant<int, (2 * PER < 60 ? 2 * PER : 60)>{});
        else run(std::integral_constant<int, (3 * PER < 60 ? 3 * PER : 60)>{});
    } else {
    stamp(1);
    const double mu_up = (valid && tid > 0) ? g.mu[N + tidc] : 1.0;   // loaded here: no global load may follow the first store
    // Transposed work items for the two mu -> 0 treatments: lane = 8 * (row of the chunk) + (position).
    const int uT = lane >> 3, pT = lane & 7;
    const double muT = (pT + 1 < N) ? g.mu[N + pT + 1] : 1.0;          // upward direction N + pT + 1
    const double* xrow = s_x + uT * 16;
    // =============================== downward ===============================
    {
        const int m = tidc;
        const int vo = m * 8;
        const double mu = g.mu[m];
        const bool tr = valid && m <= N - 2;
        const bool small = tr && fabs(mu) < kMuThreshold;       // spec:333
        const bool stdl = tr && !small;
        const double nrmu = stdl ? -1.0 / mu : 0.0;
        const bool has_small = wl && g.nsmall > 0;
        double Dv = 0, Jprev = 0;
        double c[kFixMaxSrc] = {0, 0, 0, 0, 0};
        double cT[kFixMaxSrc] = {0, 0, 0, 0, 0};               // row pT of the table: work item (uT, pT) rewrites direction N-1-pT
        int sl[kFixMaxSrc] = {0, 0, 0, 0, 0};
        int nfx = 0;
        bool fixlane = false;
        auto load_fix = [&](int zz) {
            const double* ftC = s_fixc + zz * fixcap * kFixMaxSrc;
            nfx = zz == 0 ? nfix0 : (zz == 1 ? nfix1 : ((!MZ || zz == 2) ? nfix2 : zr.nfix(zz)));
            const int ns = nfx < 2 ? 2 : (nfx < kFixMaxSrc ? nfx : kFixMaxSrc);     // In_limit:118-141
            fixlane = valid && nfx > 0 && m >= N - nfx;
            const int i = fixlane ? N - 1 - m : 0;
            const int s0 = nfx < 2 ? N - nfx - 2 : N - nfx - ns;
#pragma unroll
            for (int q = 0; q < kFixMaxSrc; ++q) {
                c[q] = (fixlane && q < ns) ? ftC[i * ns + min(q, ns - 1)] : 0.0;
                cT[q] = (pT < nfx && q < ns) ? ftC[pT * ns + min(q, ns - 1)] : 0.0;
                sl[q] = (s0 + min(q, ns - 1)) & 63;
            }
        };
        if (wl) load_fix(0);
        const int sbase = small ? (m - g.small_lanes[0]) * L : 0;   // the small lanes are consecutive directions
        int t0 = 0, q = 0;
        // MODE (plain chunks only; chosen once per run of plain chunks, which lies inside one zone): 0 this wave has no treated
        // lane here, 1 the transposed extrapolation alone (the usual case of the wave with the mu -> 0- lanes), 2 anything else
        // (|mu| < 0.01 lanes, more than 8 rewritten directions).  Modes 0 and 1 update no row conditionally: no register
        // copies where the branches join.
        auto process = [&](auto special_t, auto mode_t, const double* slot, double (&Jc)[TC], double (&Ic)[TC], double (&Ec)[TC]) __attribute__((always_inline)) {
            constexpr bool SP = decltype(special_t)::value;
            constexpr int MODE = decltype(mode_t)::value;
            double cc[TC], v[TC], Sc[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) Sc[u] = 0;
            if ((SP || MODE == 2) && has_small) {
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const double sv = s_S[sbase + min(t0 + u, L - 1)];
                    Sc[u] = small ? sv : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int t = SP ? min(t0 + u, L - 1) : t0 + u;
                const double hk = s_hd[t];
                const double Jp = u == 0 ? Jprev : Jc[u - 1];
                cc[u] = rec_src(rec_hr(hk, nrmu), Jp, Ec[u], Jc[u]);
            }
            if (!SP) {
                // Chunk-local form of the recurrence (the definition shared with transport_scan.hip, whose waves evaluate the
                // chunks of a sweep at once: the same bits, so that which kernel transports a launch may follow the live
                // count): d_u = E_u d_{u-1} + c_u, p_u = E_u p_{u-1} from d = 0, p = 1, then S_u = fma(S_in, p_u, d_u).
                {
                    double d = 0, p = 1, dl[TC], pl[TC];
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        d = rec_step(d, Ec[u], cc[u]);
                        p = u == 0 ? Ec[0] : rec_hr(p, Ec[u]);
                        dl[u] = d; pl[u] = p;
                    }
#pragma unroll
                    for (int u = 0; u < TC; ++u) v[u] = rec_step(Dv, pl[u], dl[u]);
                    Dv = v[TC - 1];
                }
                if (MODE == 2 && has_small) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) v[u] = rec_add(v[u], Sc[u]);
                }
                // In_limit:113-141 as a linear map of the source lanes.  Up to 8 rewritten directions: the
                // chunk goes through LDS and work item (uT, pT) does row uT, direction N-1-pT -- one pass for
                // the 8 rows instead of 8 passes of cross-lane reads.
                const bool tfix = MODE == 1 || (MODE == 2 && wl && nfx > 0 && nfx <= 8 && sl[0] >= xb_dn);   // sources inside the 16-lane window
                if (tfix) {
#pragma unroll
                    for (int u = 0; u < TC; ++u)
                        if (lane >= xb_dn && lane < xb_dn + 16) s_x[u * 16 + lane - xb_dn] = v[u];
                    double acc = 0;
#pragma unroll
                    for (int k = 0; k < kFixMaxSrc; ++k) acc = fix_acc(cT[k], xrow[max(sl[k] - xb_dn, 0)], acc);
                    const int mT = N - 1 - pT;
                    const double IcT = ACC ? slot[(2 * TC + uT) * RS + max(mT, 0)] : 0.0;
                    if (pT < nfx) {
                        const int voT = (t0 + uT) * RB + mT * 8;
                        rstore(rIn, voT, 0, acc);
                        if (ACC) rstore(rI, voT, 0, IcT + acc);
                        if (SAVED) rstore(rS, voT, 0, acc);
                    }
                } else if (MODE == 2 && wl && nfx > 0) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        double acc = 0;
#pragma unroll
                        for (int k = 0; k < kFixMaxSrc; ++k) acc = fix_acc(c[k], readlane_f64(v[u], sl[k]), acc);
                        v[u] = fixlane ? acc : v[u];
                    }
                }
                if (valid && !(tfix && fixlane)) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        const int so = (t0 + u) * RB;
                        rstore(rIn, vo, so, v[u]);
                        if (ACC) rstore(rI, vo, so, Ic[u] + v[u]);
                        if (SAVED) rstore(rS, vo, so, v[u]);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const int t = t0 + u;
                    if (wl) { const int zs = zr.starts(t); if (zs) load_fix(zs); }
                    const double Dn = rec_step(Dv, Ec[u], cc[u]);
                    double x = has_small ? rec_add(Dn, Sc[u]) : Dn;
                    if (wl && nfx > 0) {
                        double acc = 0;
#pragma unroll
                        for (int k = 0; k < kFixMaxSrc; ++k) acc = fix_acc(c[k], readlane_f64(x, sl[k]), acc);
                        x = fixlane ? acc : x;
                    }
                    v[u] = x;
                    const bool zone_end = zr.ends(t);                   // the next zone starts from the final row (spec:359,378)
                    Dv = t < L ? (zone_end ? (stdl ? x : 0.0) : Dn) : Dv;
                    if (valid && t < L) {
                        const int so = t * RB;
                        rstore(rIn, vo, so, x);
                        if (ACC) rstore(rI, vo, so, Ic[u] + x);
                        if (SAVED) rstore(rS, vo, so, x);
                    }
                }
            }
            if (SP && t0 + TC >= L) {                           // (the last chunk of a sweep always takes the general body)
#pragma unroll
                for (int u = 0; u < TC; ++u)
                    if (t0 + u == L - 1) { sfc_own = v[u]; rdn_v = v[u]; rdn_i = Ic[u] + v[u]; }
            }
            Jprev = Jc[TC - 1];
            t0 += TC;
        };
        auto chunk = [&](auto special_t, auto mode_t) __attribute__((always_inline)) {
            const double* sp = ring + (size_t)slot_rd * SLOT + tid;
            slot_rd = slot_rd + 1 == NS ? 0 : slot_rd + 1;
            double Jc[TC], Ic[TC], Ec[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                Jc[u] = sp[(0 * TC + u) * RS];
                Ec[u] = sp[(1 * TC + u) * RS];
                Ic[u] = ACC ? sp[(2 * TC + u) * RS] : 0.0;
            }
            process(special_t, mode_t, sp - tid, Jc, Ic, Ec);
            wg_barrier();
        };
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        using M2 = std::integral_constant<int, 2>;
        while (q < NCH) {
            const int qe = q + plain_run(sp_dn, q);
            const int mode = (!wl || (nfx == 0 && !has_small)) ? 0 : ((!has_small && nfx <= 8 && sl[0] >= xb_dn) ? 1 : 2);
            if (mode == 0) for (; q < qe; ++q) chunk(std::false_type{}, M0{});
            else if (mode == 1) for (; q < qe; ++q) chunk(std::false_type{}, M1{});
            else for (; q < qe; ++q) chunk(std::false_type{}, M2{});
            // general body: zone boundaries, and the last chunk (its final row feeds the surface and the test)
            if (q < NCH) { chunk(std::true_type{}, M2{}); ++q; }
        }
    }

    stamp(2);
    // =============================== surface ===============================
    if (surface != SOSRT_SURFACE_NONE) {
        if (valid) s_sfc[tid] = sfc_own;
        wg_barrier();
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
        wg_barrier();
        if (lane == 0) s_red[tid >> 6] = ws;
        wg_barrier();
        double S = 0;
        for (int i = 0; i < nwc; ++i) S += s_red[i];
        Bv = (surface == SOSRT_SURFACE_LAMBERTIAN_README ? 2 : -2) * rho * S;          // lam:399 as coded (negative), or README.md:215
    }

    stamp(3);
    // =============================== upward ===============================
    {
        const int mj = N + tidc;
        const int vo = mj * 8;
        const bool tr = valid && tid > 0;
        const double mu = mu_up;
        const double prmu = tr ? 1.0 / mu : 0.0;
        const int last_cand = min(N - 3, 61);
        bool notfound = false;
        double U = Bv, Jnext = 0;
        // spec:401-409 for one row (t) held across wave 0: x is the raw row, returns the blended value.  No stop among the
        // candidates of this wave: the row stays raw and is finished after the sweep (finish_flagged_rows) -- unless it is the first
        // row of a zone, whose blended value is the state of the zone above: then the whole sweep is redone row by row
        auto blend = [&](double x, int t) {
            const double x1 = lane_up1(x), x2 = lane_up1(x1);
            const bool stop = lane >= 1 && lane <= last_cand && !(fabs((x - x1) - (x1 - x2)) > 0.0001);
            const unsigned long long mk = __ballot(stop);
            const int kf = mk ? __ffsll((long long)mk) : 1;
            if (mk == 0) {
                notfound = true;
                if (N - 3 > 61 && lane == 0) {
                    if (zr.starts(t)) s_flag[0] = 1;
                    else { flag_row(s_nf, t); s_flag[2] = 1; }
                }
            }
            const double r0 = readlane_f64(x, 0), rk = readlane_f64(x, kf);
            const double w = blend_weight(mu, readlane_f64(prmu, kf));         // mu_m / mu_kf
            const double bl = blend_val(w, r0, rk);
            return (tr && tid < kf) ? bl : x;
        };
        int t0 = L - 1, q = NCH;
        // MODE (plain chunks): 0 this wave has no mu -> 0+ lanes, 1 wave 0
        auto process = [&](auto special_t, auto mode_t, const double* slot, double (&Jc)[TC], double (&Ic)[TC], double (&Ec)[TC]) __attribute__((always_inline)) {
            constexpr bool SP = decltype(special_t)::value;
            constexpr int MODE = decltype(mode_t)::value;
            double cc[TC], v[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int t = SP ? max(t0 - u, 0) : t0 - u;
                const double hk = s_hd[t + 1];
                const double Jx = u == 0 ? Jnext : Jc[u - 1];
                const double src = rec_src(rec_hr(hk, prmu), Jx, Ec[u], Jc[u]);
                // first row of a zone: attenuate the boundary only (spec:413-419,433-439)
                cc[u] = (SP && zr.ends(t)) ? 0.0 : src;
            }
            if (!SP) {
                {   // chunk-local form, as in the downward sweep
                    double d = 0, p = 1, dl[TC], pl[TC];
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        d = rec_step(d, Ec[u], cc[u]);
                        p = u == 0 ? Ec[0] : rec_hr(p, Ec[u]);
                        dl[u] = d; pl[u] = p;
                    }
#pragma unroll
                    for (int u = 0; u < TC; ++u) v[u] = rec_step(U, pl[u], dl[u]);
                    U = v[TC - 1];
                }
                // spec:401-409.  The search almost always ends within the first few directions: the chunk goes
                // through LDS and work item (uT, pT) tests candidate k = pT+1 of row uT, then produces direction
                // N+k of that row (blended below the stop, raw above it) -- one pass for the 8 rows.  A row whose
                // search goes beyond 8 candidates sends the chunk through the row-by-row path.
                bool tblend = false;
                if (MODE == 1) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        v[u] = tid == 0 ? Jc[u] : v[u];                  // spec:401
                        if (lane < 16) s_x[u * 16 + lane] = v[u];
                    }
                    const int k = pT + 1;
                    const double xa = xrow[k], xb = xrow[k + 1], xc = xrow[k + 2];
                    const bool stop = (k <= last_cand) & !(fabs((xa - xb) - (xb - xc)) > 0.0001);
                    const unsigned long long mk = __ballot(stop);
                    const unsigned bits = (unsigned)(mk >> (8 * uT)) & 0xffu;    // the candidates of this row
                    tblend = __ballot(bits != 0) == ~0ull;
                    if (tblend) {
                        const int kf = __ffs((int)bits) + 1;                     // ks + 1, <= 9
                        const double r0 = xrow[0], rk = xrow[kf];
                        const double w = blend_weight(muT, s_prmu[kf]);              // mu_m / mu_kf
                        const double bl = blend_val(w, r0, rk);
                        const double val = k < kf ? bl : xa;
                        const double IcT = ACC ? slot[(2 * TC + uT) * RS + k] : 0.0;
                        if (k < N) {
                            const int voT = (t0 - uT) * RB + (N + k) * 8;
                            rstore(rIn, voT, 0, val);
                            if (ACC) rstore(rI, voT, 0, IcT + val);
                            if (SAVED) rstore(rS, voT, 0, val);
                        }
                    } else {
                        // (rare; stored here so that the rows are not updated conditionally: no copies at the join)
#pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            const double xb_ = blend(v[u], t0 - u);
                            if (valid) {
                                const int so = (t0 - u) * RB;
                                rstore(rIn, vo, so, xb_);
                                if (ACC) rstore(rI, vo, so, Ic[u] + xb_);
                                if (SAVED) rstore(rS, vo, so, xb_);
                            }
                        }
                    }
                }
                if (valid && (MODE == 0 || (tblend && !(tid >= 1 && tid <= 8)))) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        const int so = (t0 - u) * RB;
                        rstore(rIn, vo, so, v[u]);
                        if (ACC) rstore(rI, vo, so, Ic[u] + v[u]);
                        if (SAVED) rstore(rS, vo, so, v[u]);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const int t = t0 - u;
                    const double Un = rec_step(U, Ec[u], cc[u]);
                    double x = Un;
                    if (w0 && t >= 0) x = blend(tid == 0 ? Jc[u] : Un, t);
                    v[u] = x;
                    const bool zone_start = zr.starts(t) != 0;          // blended row feeds the zone above
                    U = t >= 0 ? ((zone_start && tr) ? x : Un) : U;
                    if (valid && t >= 0) {
                        const int so = t * RB;
                        rstore(rIn, vo, so, x);
                        if (ACC) rstore(rI, vo, so, Ic[u] + x);
                        if (SAVED) rstore(rS, vo, so, x);
                    }
                }
            }
            if (SP && t0 - TC < 0) {
#pragma unroll
                for (int u = 0; u < TC; ++u)
                    if (t0 - u == 0) { rup_v = v[u]; rup_i = Ic[u] + v[u]; }
            }
            Jnext = Jc[TC - 1];
            t0 -= TC;
        };
        auto chunk = [&](auto special_t, auto mode_t) __attribute__((always_inline)) {
            const double* sp = ring + (size_t)slot_rd * SLOT + tid;
            slot_rd = slot_rd + 1 == NS ? 0 : slot_rd + 1;
            double Jc[TC], Ic[TC], Ec[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                Jc[u] = sp[(0 * TC + u) * RS];
                Ec[u] = sp[(1 * TC + u) * RS];
                Ic[u] = ACC ? sp[(2 * TC + u) * RS] : 0.0;
            }
            process(special_t, mode_t, sp - tid, Jc, Ic, Ec);
            wg_barrier();
        };
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        while (q < NQ) {
            const int qe = q + plain_run(sp_up, q - NCH);
            if (!w0) for (; q < qe; ++q) chunk(std::false_type{}, M0{});
            else for (; q < qe; ++q) chunk(std::false_type{}, M1{});
            // general body: zone boundaries, and the last chunk (row 0 feeds the convergence test)
            if (q < NQ) { chunk(std::true_type{}, M1{}); ++q; }
        }
        if (w0 && notfound && lane == 0 && N - 3 <= 61) s_flag[1] = 1;     // every candidate was in this wave: IndexError
    }
    }   // computing waves
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
    if (!s_flag[0] && s_flag[2]) {
        // rows whose search went past the lanes of wave 0: finished one by one (the ring is free by now)
        if (finish_flagged_rows<ACC, SAVED>(s_nf, L, N, RB, g.mu, rIn, rI, rS, ring, rup_v, rup_i)) {
            if (tid == 0) {                                             // the reference raises IndexError (spec:404)
                a.cv.status[b] = SOSRT_COL_INDEXERROR;
                if (ACC) { a.cv.active[b] = 0; a.cv.norders[b] = a.order; atomicSub(a.cv.nactive, 1); }
            }
            return;
        }
    }
    if (s_flag[0]) {
        // The first row of a zone has no stop among the lanes of wave 0, and its blended value is the state of the zone above
        // (spec:403-406 has no bound): redo that sweep here, row by row (redo_upward_sweep; the ring is free by now).
        const bool missing = redo_upward_sweep<ACC, SAVED, MZ>(L, N, RB, zr, s_hd, g.mu, rJ, rE, rIn, rI, rS, Bv, ring,
                                                            rup_v, rup_i);
        if (missing) {                                                  // the reference raises IndexError (spec:404)
            if (tid == 0) {
                a.cv.status[b] = SOSRT_COL_INDEXERROR;
                if (ACC) { a.cv.active[b] = 0; a.cv.norders[b] = a.order; atomicSub(a.cv.nactive, 1); }
            }
            return;
        }
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

// doubles of LDS besides the ring: surface row, reduction slots, half thicknesses, small-mu rows
// rewritten directions the compact coefficient tables hold: the largest int(c N) of In_limit / I1_In:124-127
inline int ring_fixcap(const Grid& g) { return (int)(0.06 * g.N) + 1; }
inline size_t ring_extra_doubles(const Grid& g, int nt, int nzcap = kRingZones) {
    const int rs = g.N <= 128 ? 128 : 256, fc = nzcap * ring_fixcap(g) * kFixMaxSrc;
    return (size_t)(rs > fc ? rs : fc) + nt / 64 + 2 + g.L + 1 + (size_t)g.nsmall * g.L + TC * 16 + 16 + (g.L + 63) / 64;
}

template <int PIECES>
void launch_p(hipStream_t s, dim3 grid, dim3 block, const TransportArgs& a, int NS) {
    const int nt = (int)block.x;
    const int narr = a.accumulate ? 3 : 2;
    const size_t shm = ((size_t)NS * narr * TC * 128 * PIECES + ring_extra_doubles(a.g, nt, a.nzcap)) * sizeof(double);
#define SOSRT_RING_LAUNCH_Z(ACC_, SAVED_, MZ_)                                                                 \
    do {                                                                                                       \
        auto kern = k_transport_ring<ACC_, SAVED_, PIECES, MZ_>;                                               \
        static PerDeviceOnce big_lds;                                                                                 \
        if (big_lds.first()) {                                                                                        \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)kRingLdsBytes);                                                                   \
        }                                                                                                      \
        hipLaunchKernelGGL(kern, grid, block, shm, s, a, NS, g_ring_debug, ring_fixcap(a.g));                           \
    } while (0)
    // (a batch with a column of more than three zones: the instantiation that tests every boundary of the zone table, ZoneRows<true>)
#define SOSRT_RING_LAUNCH(ACC_, SAVED_)                                                                        \
    do {                                                                                                       \
        if (a.nzcap > kRingZones) SOSRT_RING_LAUNCH_Z(ACC_, SAVED_, true);                                     \
        else SOSRT_RING_LAUNCH_Z(ACC_, SAVED_, false);                                                         \
    } while (0)
    if (a.accumulate) {
        if (a.saved) SOSRT_RING_LAUNCH(true, true);
        else SOSRT_RING_LAUNCH(true, false);
    } else {
        SOSRT_RING_LAUNCH(false, false);
    }
#undef SOSRT_RING_LAUNCH
#undef SOSRT_RING_LAUNCH_Z
}

}  // namespace

// The ring needs half rows that are multiples of 16 bytes (N even), at most two 1-KiB pieces per
// half row, and at least two slots beside the per-column tables.
bool transport_ring_ok(const Grid& g) {
    if (g.N % 2 || g.N < 4 || g.N > 256) return false;
    const int nwc = (g.N + 63) / 64, pieces = g.N <= 128 ? 1 : 2;
    const size_t slot_bytes = (size_t)3 * TC * 128 * pieces * sizeof(double);
    // (the finishing of flagged rows works in the ring once the sweeps are over: it must fit two slots of the smallest form)
    if (flagged_rows_work_doubles(g.L, g.N) * sizeof(double) > 2 * slot_bytes * 2 / 3) return false;
    return 2 * slot_bytes + ring_extra_doubles(g, (nwc + 4) * 64) * sizeof(double) <= kRingLdsBytes;
}

// the same with the per-zone tables of a batch whose columns have up to nzcap zones
bool transport_ring_fits(const Grid& g, int nzcap) {
    if (!transport_ring_ok(g)) return false;
    const int nwc = (g.N + 63) / 64, pieces = g.N <= 128 ? 1 : 2;
    const size_t slot_bytes = (size_t)3 * TC * 128 * pieces * sizeof(double);
    return 2 * slot_bytes + ring_extra_doubles(g, (nwc + 4) * 64, nzcap < kRingZones ? kRingZones : nzcap) * sizeof(double) <= kRingLdsBytes;
}

// slots: ring depth wanted (2..6)
void launch_transport_ring(hipStream_t s, dim3 grid, const TransportArgs& a, int slots) {
    if (a.slots > 0) slots = a.slots;
    const int N = a.g.N;
    const int nwc = (N + 63) / 64;
    const int pieces = N <= 128 ? 1 : 2;
    const int narr = a.accumulate ? 3 : 2;
    const dim3 block((nwc + 2) * 64);
    const size_t slot_bytes = (size_t)narr * TC * 128 * pieces * sizeof(double);
    const size_t extra = ring_extra_doubles(a.g, (int)block.x, a.nzcap) * sizeof(double);
    int NS = slots < 2 ? 2 : slots;
    while (NS > 2 && NS * slot_bytes + extra > kRingLdsBytes) --NS;
    if (pieces == 1) launch_p<1>(s, grid, block, a, NS);
    else launch_p<2>(s, grid, block, a, NS);
}

}  // namespace sosrt
