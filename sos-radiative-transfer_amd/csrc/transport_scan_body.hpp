// k_transport_scan: one order of transport (spec:326-449) with the chunks of a sweep dealt to several waves.
//
// The ring kernel (transport_ring.hip) gives a column two computing waves that walk the 2 L rows of an order one after
// the other; a lone column is bound by the instruction stream of those waves (43-48 us per order whatever else runs on
// the GPU), and the last twenty orders of a sweep are lone columns.  Only ONE operation per row is serial, the step of the
// recurrence  D_t = E_t D_{t-1} + c_t;  the source terms, the mu -> 0 treatments, the running total and the stores are
// row-local.  Here a chunk of TC = 8 rows is evaluated in its chunk-local form
//     d_u = E_u d_{u-1} + c_u,   p_u = E_u p_{u-1}      (d_{-1} = 0, p_{-1} = 1: independent of every other chunk)
//     D_u = fma(p_u, D_in, d_u),                          D_in = the value carried out of the previous chunk
// so that the chunks of a sweep go round-robin to SW waves per lane group (64 directions), which work on their chunks
// at once and hand the carried value on through LDS (one fma per chunk on the serial path instead of eight recurrence
// steps plus everything else).  Chunks with a zone boundary, and the last chunk of a sweep, keep the serial form of the
// ring kernel (state resets at zone ends, spec:359,378; blended restarts): they are single hops of the chain.
//
// The chunk-local form rounds differently from the serial one (1e-16 per step).  It is therefore the definition for the ring
// kernel as well (transport_ring.hip evaluates its plain chunks the same way, one after the other): the two kernels give
// the same bits, and which of them transports a launch follows the number of live columns (api.hip: SOSRT_SCAN_COLS) --
// the ring kernel where the launch is HBM-bound, this one where it is latency-bound -- while a column's result stays
// independent of its batch.
//
// Roles.  A workgroup is  nwc * SW computing waves (lane group x chunk residue)  +  SW loader waves.  The rows of Jn and
// of the attenuation table reach the computing waves through LDS: a pool of NST stages, chunk g (0 .. 2 NCH - 1 over the two
// sweeps) in stage g mod NST; loader w copies the chunks g = w, w + SW, ... (1-KiB half rows, `buffer_load_dwordx4 ... lds`)
// as soon as the chunk that used the stage before has been taken out of it, and raises a flag; the computing waves take a
// stage into registers and hand it back at once, so with NST > SW a residue's next chunk is on its way before the current
// one is started (a chunk of 17 KiB takes some 4 000 cycles to arrive, more than a wave needs for a chunk).  A computing
// wave loads only its rows of the running total itself, which it needs last, for its stores.  Everything is flag-driven
// (LDS words polled by the waiting wave, plain accesses in program order: the LDS serves a wave's instructions in order),
// no barrier inside a sweep; every poll is bounded (SOSRT_COL_INTERNAL on expiry, never expected).
//
// Measured (MI355X, lone column, L = 200, N = 128): 41-42 us per order against 48 for the ring kernel; in-kernel stamps
// (-DSOSRT_SCAN_STAMPS, tools/stamps_scan.py): 35 us, of which a computing wave spends a third waiting (stage, carried value)
// -- the memory path of the one CU is the bound (six waves per lane group instead of four: the same time).  Hence the
// SPLIT form below, two workgroups per column: 33-35 us.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "../../include/sosrt.h"
#include "kernels.hpp"
#include "transport_util.hpp"

namespace sosrt {

namespace {


// One workgroup per column: 2 lane groups x 4 chunk residues.  SPLIT (two workgroups per column, chosen for launches with
// few live columns: a lone column is bound by the memory path of the one CU it lives on): a workgroup takes HALF the
// directions -- part 0 the |mu| < mu_mid half of both sweeps (upward directions N .. N+63 and their mirror images, the
// downward directions N-64 .. N-1: every mu -> 0 treatment), part 1 the rest -- so that the specular reflection, which maps a
// downward direction onto its mirror image, stays inside a workgroup, and no workgroup ever waits for another; one lane
// group x 8 chunk residues.  The arithmetic per direction is the same, so the choice is free (it follows the live count).
// WIDE (split form only): the instantiation for shapes beyond the BASELINE ones -- an odd direction count (the upward half of a
// row then starts 8 bytes off a 16-byte boundary: the stages are filled 4 bytes per lane instead of 16), more than 256
// directions (up to 512: eight workgroups per column), more than 64 chunks per sweep (up to 128: L <= 1024; the masks of the
// chunks with a zone boundary are two words) -- the reference's shipped N = 501, L = 800 (spec:33,57) is all three.  Ten stages
// instead of twelve and the |mu| < 0.01 lanes read from memory instead of an LDS table, so that the tables of such a shape
// fit beside the stages.  Same operations per direction as every other form.
template <bool SPLIT, bool WIDE = false> struct ScanCfg {
    static constexpr int SW = SPLIT ? 8 : 4;               // waves per lane group = chunks of a sweep in flight
    static constexpr int NST = SPLIT ? (WIDE ? 10 : 12) : 6;   // stages: chunk g (0 .. 2 NCH - 1 over the two sweeps) uses stage g mod NST
    static constexpr int SROW = SPLIT ? 64 : 128;          // doubles per staged row (half a 1-KiB half row when split)
    static constexpr int STAGE = (2 * TC + 1) * SROW;      // doubles per stage: Jn, attenuation, + the Jn row before the chunk
};
constexpr int NLOAD = 4;                                   // loader waves
constexpr int CR = 16;                                     // ring of carried values per lane group (> SW)
constexpr int kScanDirs = 512;                             // most directions per hemisphere of the split form
constexpr int kScanChunks = 128;                           // most chunks per sweep (WIDE; 64 otherwise: one mask word)
constexpr int kScanScratch = 5 * kScanDirs + 16;           // doubles per column of the split form's exchange: 4 test rows + surface row + 32 words of flagged rows
// (the mask of flagged rows is (L + 31) / 32 ints in those 16 doubles: at most kScanChunks chunks of TC rows per sweep)
static_assert(kScanChunks * TC / 32 <= 32, "the flagged-row mask of the split form's exchange holds 32 ints: 128 chunks of TC rows");
constexpr size_t kScanLdsBytes = 152 * 1024;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned kSpinLimit = 1u << 20;                  // (a carried value arrives within a few thousand polls)
constexpr unsigned kFusedSpinLimit = 1u << 21;             // polls of a word another workgroup publishes (FUSED): seconds
// NC: the number of directions per hemisphere as a compile-time constant (0: taken from the grid).  The computing waves of a lone
// column are bound by their instruction streams (a run-time stride in the stage reads alone cost 10 us per launch when it was
// tried); with N fixed the row stride, the direction offsets and the part arithmetic become immediates.  The two sizes of the
// BASELINE configurations are instantiated for the split form, which is the one that transports lone columns.
//
// The body of one order of one column is a device function: k_transport_scan (transport_scan.hip) calls it once per launch,
// the order-loop kernel (order_loop.hip, FUSED) once per order of a column it keeps for several orders.  FUSED changes no
// arithmetic; it changes how the data meets the other workgroups of the launch:
//   - the column's field rows (In, the running total I) are stored write-through and loaded `sc1`, and the source function
//     is loaded `sc1` (ScanAux): the contraction of the next order runs on OTHER workgroups of the same launch, whose XCDs'
//     L2s are not coherent with this one's, and this CU's own L1 still holds last order's lines;
//   - the loaders wait for this order's source function before their first request (ScanFused::jn_done);
//   - nothing returns early: every workgroup of a column leaves with the column's verdict (1: it stops -- converged, error
//     or out of its order budget), the workgroups that do not run the loop test wait for it (ScanFused::ord_done);
//   - the per-column tables are filled in the first order only.
struct ScanFused {
    int k = 1;                       // order of this launch (1 ..)
    int first = 1;                   // fill the per-column tables
    int last = 0;                    // the launch's order budget ends with this order
    const int* jn_done = nullptr;    // [1] tiles of the column's source function stored so far (order_loop.hip)
    int jn_need = 0;                 // ... and how many make this order's complete
    int* ord_done = nullptr;         // [1] orders this column has completed in this launch
    int* col_stop = nullptr;         // [1] set (before ord_done) when the column stops
    int* abort = nullptr;            // [1] launch-wide: a bounded poll expired somewhere
    int* orders = nullptr;           // [1] launch-wide: column.orders finished (statistics)
    unsigned long long* log = nullptr;   // diagnostic builds (-DSOSRT_OL_STAMPS): event log of the launch
    int wg = 0;
};
// diagnostic builds: one 64-bit event {workgroup:16, order:8, event:8, wall clock (10 ns):32} appended to the launch's log
#ifdef SOSRT_OL_STAMPS
__device__ __forceinline__ void ol_stamp(unsigned long long* log, int wg, int k, int ev) {
    if (!log) return;
    const unsigned long long i = __hip_atomic_fetch_add(log, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i < 65000) log[1 + i] = ((unsigned long long)(wg & 0xffff) << 48) | ((unsigned long long)(k & 0xff) << 40) | ((unsigned long long)(ev & 0xff) << 32) |
                                (unsigned long long)(wall_clock64() & 0xffffffffull);
}
#define SOSRT_OL_STAMP(log_, wg_, k_, ev_) ol_stamp(log_, wg_, k_, ev_)
#else
#define SOSRT_OL_STAMP(log_, wg_, k_, ev_)
#endif
// bounded poll of a word another workgroup publishes (agent-scope, past the L1); false: gave up, *abort set
__device__ __forceinline__ bool fused_wait_ge(const int* p, int want, int* abort) {
    for (unsigned it = 0;; ++it) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if ((it & 63) == 63 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        if (it > kFusedSpinLimit) { __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        __builtin_amdgcn_s_sleep(8);
    }
}

template <bool ACC, bool SAVED, bool SPLIT, int NC = 0, bool MZ = false, bool FUSED = false, bool WIDE = false>
__device__ __forceinline__ int transport_scan_order(const TransportArgs& a, int fixcap, const int b, const int part, const ScanFused& fu) {
    static_assert(!WIDE || (SPLIT && !FUSED && NC == 0), "WIDE is an instantiation of the plain split form");
    using Cfg = ScanCfg<SPLIT, WIDE>;
    constexpr int SW = Cfg::SW, NST = Cfg::NST, SROW = Cfg::SROW, STAGE = Cfg::STAGE;
    // SPLIT: ceil(N / 64) workgroups per column (two at N = 128, four at N = 256), part p
    const int nparts = SPLIT ? ((NC ? NC : a.g.N) + 63) >> 6 : 1;
    // cache policy of the field rows (see above): loads of data another workgroup (or this one, an order ago) stored; stores
    constexpr int LDX = FUSED ? 16 : 0, STX = FUSED ? 17 : 0;
    const Grid& g = a.g;
    // (FUSED: the body runs inside the order loop of order_loop.hip.  Everything below that derives from the thread id alone is
    // invariant in that loop, and the compiler would hoist it out and keep it -- lane constants of both sweeps -- in registers
    // across the whole body: 168 VGPRs + 28 spilled against 129.  An opaque copy of the thread id per order keeps the body what
    // it is in the one-order kernel.)
    int tid_ = threadIdx.x;
    if (FUSED) asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int L = g.L, N = NC ? NC : g.N, D = 2 * N;
    if (NC) fixcap = (int)(0.06 * NC) + 1;                     // (scan_fixcap, as an immediate)
    const int nwc = SPLIT ? 1 : (N + 63) >> 6;                 // lane groups of this workgroup
    const int ND = ((N + 63) >> 6) * 64;                       // stride of the per-direction LDS rows
    const int ncw = nwc * SW;                                  // computing waves; the NLOAD loader waves follow
    const bool loader = wid >= ncw;
    const int lg = wid % nwc, grp = loader ? wid - ncw : wid / nwc;   // this wave: 64 directions, chunks grp, grp + SW, ... (loaders: grp, grp + NLOAD, ...)
    // downward direction dir_dn, then upward direction N + dir; split: part p = upward 64p .. 64p+63 and their mirror images,
    // the downward directions N-64(p+1) .. N-64p-1 (the last part of an N that is no multiple of 64: from 0)
    const int dir = SPLIT ? part * 64 + lane : lg * 64 + lane;
    const int dn0 = SPLIT ? max(N - 64 * (part + 1), 0) : 0;   // first downward direction of this part
    const int dir_dn = SPLIT ? dn0 + lane : dir;
    const bool valid = dir < N;
    const bool valid_dn = SPLIT ? dir_dn < N - 64 * part : valid;
    const int dirc = valid ? dir : N - 1;
    const int dirc_dn = valid_dn ? dir_dn : 0;
    const bool w0 = !loader && (SPLIT ? part == 0 : lg == 0);                    // holds the mu -> 0+ lanes
    const bool wl = !loader && (SPLIT ? part == 0 : lg == ((N - 1) >> 6));       // holds the mu -> 0- lanes
    const int lane_last = SPLIT ? 63 : ((N - 1) & 63);         // lane of downward direction N-1 in the wave that holds it
    const int dn_base = SPLIT ? N - 64 : (((N - 1) >> 6) << 6);  // downward direction of lane 0 of that wave
    const int nwaves = blockDim.x >> 6;
    const int NCH = (L + TC - 1) / TC;

#ifdef SOSRT_SCAN_STAMPS    // diagnostic builds: cycle stamps of the first and the last wave, [b][2][8] (tools/stamps_scan.py)
    unsigned long long t_wait = 0, t_stage = 0, t_pre = 0, t_post = 0, t_c = 0;
    auto stamp = [&](int i) __attribute__((always_inline)) {     // [b][2][16]: 0-5 timeline, 6 carry wait, 7 stage wait, 8 pre-work, 9 post-work
        if (a.stamps && lane == 0 && part == 0 && (wid == 0 || wid == ncw - 1))
            a.stamps[((size_t)b * 2 + (wid == 0 ? 0 : 1)) * 16 + i] = i == 6 ? t_wait : (i == 7 ? t_stage : (i == 8 ? t_pre : (i == 9 ? t_post : clock64())));
    };
#define SCAN_T0() t_c = clock64()
#define SCAN_T1() t_pre += clock64() - t_c
#define SCAN_T2() t_c = clock64()
#define SCAN_T3() t_post += clock64() - t_c
#define SCAN_CB(gq_)
#elif defined(SOSRT_SCAN_CHAIN)   // diagnostic builds: per-chunk timeline [b][4096]: ((part * 128 + chunk) * 5 + k), tools/stamps_chain.py
    auto stamp = [](int) {};
    int cgq = 0;
#define SCAN_C(k_) do { if (a.stamps && lane == 0) a.stamps[(size_t)b * 4096 + (part * 128 + cgq) * 5 + (k_)] = clock64(); } while (0)
#define SCAN_CB(gq_) do { cgq = (gq_); SCAN_C(0); } while (0)
#define SCAN_T0() SCAN_C(1)
#define SCAN_T1() SCAN_C(2)
#define SCAN_T2() SCAN_C(3)
#define SCAN_T3() SCAN_C(4)
#else
    auto stamp = [](int) {};
#define SCAN_T0()
#define SCAN_T1()
#define SCAN_T2()
#define SCAN_T3()
#define SCAN_CB(gq_)
#endif
    stamp(0);
    extern __shared__ double sm[];
    double* s_stage = sm;                                      // [NST][STAGE] rows of the chunks in flight
    double* s_carry = s_stage + (size_t)NST * STAGE;            // [nwc][CR][64] ring of the values carried into the chunks
    double* s_sfc = s_carry + (size_t)nwc * CR * 64;           // [ND] surface row by downward direction
    double* s_fixc = s_sfc + ND;                         // [nzcap][fixcap][kFixMaxSrc] compact extrapolation tables (nzcap: most zones of any column of the batch, >= 3)
    double* s_red = s_fixc + (MZ ? a.nzcap : kRingZones) * fixcap * kFixMaxSrc;    // [nwaves + 2]
    double* s_hd = s_red + nwaves + 2;                         // [L + 1] half layer thicknesses
    double* s_S = s_hd + L + 1;                                // [nsmall][L] the |mu| < 0.01 lanes as written by k_smallmu (WIDE: read from memory)
    double* s_prmu = s_S + (WIDE ? 0 : (size_t)g.nsmall * L);  // [16] 1/mu of the first upward directions
    double* s_conv = s_prmu + 16;                              // [4][ND] last rows of the sweeps: value, running total
    double* s_xw = s_conv + 4 * ND;                      // [ncw][2][TC * 16] per-wave exchange: values, running totals
    // flags (ints): [nwc][CR] sequence number of the carried value in a slot, [NST] chunk landed in a stage,
    // [NST][nwc] chunk taken out of a stage by a lane group (chunk numbers g + 1)
    int* s_flagq = reinterpret_cast<int*>(s_xw + (size_t)ncw * 2 * TC * 16);
    int* s_landed = s_flagq + nwc * CR;
    int* s_taken = s_landed + NST;
    int* s_nf = s_taken + NST * nwc;                           // [(L + 31) / 32] rows whose mu -> 0+ search left the first lane group (finish_flagged_rows)
    int* s_nfz = s_nf + (L + 31) / 32;                         // [kMaxZones] rewritten downward directions per zone
    __shared__ int s_flag[4];                                  // [0] redo the upward sweep row by row, [1] IndexError, [2] internal, [3] some row is flagged in s_nf
    double* s_x = s_xw + (size_t)(loader ? 0 : wid) * 2 * TC * 16;
    double* s_xI = s_x + TC * 16;
    const int xb_dn = max(lane_last - 15, 0);

    const ColDesc* __restrict__ dg = a.desc + b;
    const int nz = dg->nz;
    const ZoneRows<MZ> zr(dg);                                     // zone boundaries in scalars (MZ: all of them, else the reference's two)
    const int nfix0 = dg->nfix[0], nfix1 = dg->nfix[1], nfix2 = dg->nfix[2];
    const int surface = dg->surface;
    const double rho = dg->rho;
    const int fbytes = L * D * 8, RB = D * 8;
    const __amdgpu_buffer_rsrc_t rJ = make_rsrc(a.Jn + (size_t)b * L * D, fbytes);
    const __amdgpu_buffer_rsrc_t rE = make_rsrc(a.Etab + (size_t)(a.erep ? a.erep[b] : b) * L * D, fbytes);
    const __amdgpu_buffer_rsrc_t rIn = make_rsrc(a.In + (size_t)b * L * D, fbytes);
    const __amdgpu_buffer_rsrc_t rI = make_rsrc(ACC ? a.I + (size_t)b * L * D : a.In, ACC ? fbytes : 0);
    (void)rJ; (void)rE;
    const __amdgpu_buffer_rsrc_t rS = make_rsrc(SAVED ? a.saved + (size_t)b * a.saved_col_stride : a.In, SAVED ? fbytes : 0);

    // chunks with a zone boundary and the last chunk of each sweep: serial form (one bit per chunk of a sweep)
    // (WIDE: up to 128 chunks per sweep, two words each)
    unsigned long long sp_dn = 0, sp_up = 0, sp_dn1 = 0, sp_up1 = 0;
    if (!WIDE) {
        sp_dn = sp_up = 1ull << (NCH - 1);
        sp_dn |= zr.boundary_chunks(L, TC, false);
        sp_up |= zr.boundary_chunks(L, TC, true);
    } else {
        sp_dn = zr.boundary_chunks(L, TC, false, 0); sp_dn1 = zr.boundary_chunks(L, TC, false, 1);
        sp_up = zr.boundary_chunks(L, TC, true, 0); sp_up1 = zr.boundary_chunks(L, TC, true, 1);
        if (NCH - 1 < 64) { sp_dn |= 1ull << (NCH - 1); sp_up |= 1ull << (NCH - 1); }
        else { sp_dn1 |= 1ull << (NCH - 1 - 64); sp_up1 |= 1ull << (NCH - 1 - 64); }
    }
    auto special_dn = [&](int q) __attribute__((always_inline)) { return WIDE ? (((q < 64 ? sp_dn : sp_dn1) >> (q & 63)) & 1) != 0 : ((sp_dn >> q) & 1) != 0; };
    auto special_up = [&](int j) __attribute__((always_inline)) { return WIDE ? (((j < 64 ? sp_up : sp_up1) >> (j & 63)) & 1) != 0 : ((sp_up >> j) & 1) != 0; };


    // The value carried into chunk q of this lane group, published by the wave of chunk q - 1.  Plain LDS accesses in
    // program order, no atomics: the LDS serves a wave's instructions in order, so a flag written after the value is seen
    // after the value; a release / acquire pair at workgroup scope would also wait for the wave's global loads and stores
    // (s_waitcnt vmcnt(0)) -- a memory round trip on every hop of the chain.
    typedef __attribute__((address_space(3))) volatile int lds_vint;          // (explicit LDS pointers: a generic pointer would
    typedef __attribute__((address_space(3))) volatile double lds_vdouble;    //  make these FLAT accesses, which count in vmcnt)
    lds_vint* const l_flagq = (lds_vint*)(s_flagq + lg * CR);
    lds_vdouble* const l_carry = (lds_vdouble*)(s_carry + (size_t)lg * CR * 64 + lane);
    lds_vint* const l_landed = (lds_vint*)s_landed;
    lds_vint* const l_taken = (lds_vint*)s_taken;
    auto spin = [&](lds_vint* f, int want) __attribute__((always_inline)) {
#ifdef SOSRT_SCAN_STAMPS
        const unsigned long long w0_ = clock64();
#endif
        unsigned it = 0;
        while (*f != want) {
            __builtin_amdgcn_s_sleep(1);
            if (++it > kSpinLimit) { s_flag[2] = 1; break; }      // (never: every wave of the workgroup is resident)
        }
#ifdef SOSRT_SCAN_STAMPS
        t_wait += clock64() - w0_;
#endif
    };
    // `cs` = sequence number of the chunk in this lane group's chain (1 .. 2 NCH over the two sweeps)
    auto wait_carry = [&](int cs) __attribute__((always_inline)) -> double {
        spin(l_flagq + (cs & (CR - 1)), cs);
        return l_carry[(cs & (CR - 1)) * 64];
    };
    auto publish = [&](int cs, double val) __attribute__((always_inline)) {
        l_carry[(cs & (CR - 1)) * 64] = val;
        if (lane == 0) l_flagq[cs & (CR - 1)] = cs;
    };
    // The rows of chunk g (0 .. NCH-1: downward, NCH ..: upward) out of its stage (the loader has raised the flag), and the
    // stage back to the loaders at once.  The LDS serves the reads before the word that follows them.  The running total
    // does not go through the stage (LDS): the wave asks for its rows itself here and needs them only for its stores.
    // (clamp_t: whether rows of the chunk may fall outside the column -- the last chunk of a sweep only; a plain chunk's eight
    // row offsets are then plain additions: no measurable difference, alternating builds on one box)
    auto take = [&](auto clamp_t, int gq, int vo_, int so0, int dso, double (&J)[TC], double (&E)[TC], double (&I)[TC], double& Jx) __attribute__((always_inline)) {
        constexpr bool CLAMP = decltype(clamp_t)::value;
        const int stg = gq % NST;
#ifdef SOSRT_SCAN_STAMPS
        const unsigned long long tw_ = t_wait;
#endif
        spin(l_landed + stg, gq + 1);
#ifdef SOSRT_SCAN_STAMPS
        t_stage += t_wait - tw_; t_wait = tw_;
#endif
        typedef __attribute__((address_space(3))) double lds_double;
        // (a staged row holds the whole half row by direction, or -- split -- this workgroup's 64 directions by lane)
        const lds_double* st = (const lds_double*)(s_stage + (size_t)stg * STAGE + (SPLIT ? lane : (gq < NCH ? dirc_dn : dirc)));
        asm volatile("" ::: "memory");                      // (compiler only: the reads stay between the two flag accesses)
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            J[u] = st[(0 * TC + u) * SROW];
            E[u] = st[(1 * TC + u) * SROW];
        }
        Jx = st[(2 * TC) * SROW];
        asm volatile("" ::: "memory");
        if (lane == 0) l_taken[stg * nwc + lg] = gq + 1;
        // (requested here, after the stage has been taken: asked for before the wait for the stage -- more lead time on paper --
        // a lone column's launch takes 37 us instead of 32, alternating builds on one box, round 3)
#pragma unroll
        for (int u = 0; u < TC; ++u) I[u] = ACC ? bload_aux<LDX>(rI, vo_, CLAMP ? min(max(so0 + u * dso, 0), (L - 1) * RB) : so0 + u * dso) : 0.0;
    };

    // ------------------------------- loader side -------------------------------
    auto issue = [&](int gq) __attribute__((always_inline)) {
        const bool up = gq >= NCH;
        const int q = up ? gq - NCH : gq;
        double* dst = s_stage + (size_t)(gq % NST) * STAGE;
        const int t0 = up ? L - 1 - q * TC : q * TC;
        // byte offset of this workgroup's directions in a row: the half row (1 KiB pieces), or -- split -- its 64 directions (512 B)
        const int half = SPLIT ? (up ? N * 8 + part * 512 : dn0 * 8) : (up ? N * 8 : 0);
        if (WIDE && (N & 1)) {
            // An odd direction count: a row is 16 N bytes, so rows start on 16-byte boundaries but their upward halves (and this
            // part's window of either half) start 8 bytes off one, which the 16-byte form of the LDS-DMA cannot address: the
            // window goes 4 bytes per lane, two instructions per 512-byte row (34 per chunk instead of 9; the loaders have the
            // slack -- they are a quarter as busy as the computing waves).  (The second half's 256 bytes go into the scalar
            // offset: the instruction's immediate offset would move the LDS address as well.)
            const int v4 = lane * 4;
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int row = up ? max(t0 - u, 0) : min(t0 + u, L - 1);
                const int so = row * RB + half;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (0 * TC + u) * SROW), 4, v4, so, 0, LDX);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (0 * TC + u) * SROW + 32), 4, v4, so + 256, 0, LDX);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rE, (lds_ptr_t)(dst + (1 * TC + u) * SROW), 4, v4, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rE, (lds_ptr_t)(dst + (1 * TC + u) * SROW + 32), 4, v4, so + 256, 0, 0);
            }
            const int rx = up ? min(t0 + 1, L - 1) : max(t0 - 1, 0);     // the row before the chunk (unused by the first chunk)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (2 * TC) * SROW), 4, v4, rx * RB + half, 0, LDX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (2 * TC) * SROW + 32), 4, v4, rx * RB + half + 256, 0, LDX);
        } else if (SPLIT) {
            // A staged row of the split form is 512 B, half of what one `buffer_load_dwordx4 ... lds` moves: the two halves of
            // the wave take two consecutive rows of the stage (the row offset goes into the per-lane address, the LDS
            // destination is contiguous), 9 instructions per chunk instead of 17: a lone column's launch 32.6 -> 30.6 us
            // (alternating builds on one box, round 3).
            const int hl = lane >> 5, vl = (lane & 31) * 16;
#pragma unroll
            for (int u = 0; u < TC; u += 2) {
                const int uu = u + hl;
                const int row = up ? max(t0 - uu, 0) : min(t0 + uu, L - 1);
                const int vo = vl + row * RB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (0 * TC + u) * SROW), 16, vo, half, 0, LDX);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rE, (lds_ptr_t)(dst + (1 * TC + u) * SROW), 16, vo, half, 0, 0);
            }
            const int rx = up ? min(t0 + 1, L - 1) : max(t0 - 1, 0);     // the row before the chunk (unused by the first chunk)
            if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (2 * TC) * SROW), 16, vl, rx * RB + half, 0, LDX);
        } else {
            const int vo = lane * 16;
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int row = up ? max(t0 - u, 0) : min(t0 + u, L - 1);
                const int so = row * RB + half;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (0 * TC + u) * SROW), 16, vo, so, 0, LDX);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rE, (lds_ptr_t)(dst + (1 * TC + u) * SROW), 16, vo, so, 0, 0);
            }
            const int rx = up ? min(t0 + 1, L - 1) : max(t0 - 1, 0);     // the row before the chunk (unused by the first chunk)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_ptr_t)(dst + (2 * TC) * SROW), 16, vo, rx * RB + half, 0, LDX);
        }
    };
    // Loader `grp` carries the chunks g = grp, grp + SW, ... of both sweeps.  Chunk g goes into stage g mod NST once every
    // lane group has taken chunk g - NST out of it: with NST > SW a residue's next chunk is requested before its current
    // one is even started.
    // (One chunk in flight per loader.  Up to three -- a counted vmcnt for the oldest, landed chunks flagged while the loader waits
    // for a stage, nothing drained -- was measured twice in round 3 on the split form, alternating builds on one box: 30.5 vs 30.5 us
    // per lone-column launch.  The per-chunk timeline (tools/stamps_chain.py) shows why: twelve chunks requested at once arrive
    // eight by cycle 2 400 and four by 5 000 - 7 000 -- a CU takes in a chunk per ~450-500 cycles however many are in flight.
    // Halving what a CU has to move -- parts of 32 directions on twice the CUs, half the lanes of every wave -- did not help
    // either (40.3 vs 40.2 us in a build that made the part width a run-time value, which by itself cost 10 us: these waves
    // are bound by their instruction streams).)
    auto load_chunks = [&](int g0, int g1, bool first_issued) __attribute__((always_inline)) {
        for (int gq = g0; gq < g1; gq += NLOAD) {
            if (gq >= NST)
                for (int i = 0; i < nwc; ++i) spin(l_taken + (gq % NST) * nwc + i, gq - NST + 1);
            if (!(first_issued && gq == g0)) issue(gq);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) l_landed[gq % NST] = gq + 1;
        }
    };
    // the chunks a loader may carry before the turn-round: those whose stage was last used by a downward chunk
    const int g_seam = min(NCH + NST, 2 * NCH);
    int g_next = grp;                                           // (loaders)

    const double mu_up = (valid && dir > 0) ? g.mu[N + dirc] : 1.0;
    // Transposed work items for the two mu -> 0 treatments: lane = 8 * (row of the chunk) + (position).
    const int uT = lane >> 3, pT = lane & 7;
    const double muT = (pT + 1 < N) ? g.mu[N + pT + 1] : 1.0;          // upward direction N + pT + 1
    const double* xrow = s_x + uT * 16;
    double Bv = 0;

    // The loaders put their first chunk in flight before the per-column tables are filled: nothing of it depends on them, and
    // the two memory latencies (tables, first stage) overlap instead of adding up at the head of every launch.
    // (FUSED: the source function of this order is being written by other workgroups of the launch -- the loaders wait for it
    // below, behind the flags' reset)
    if (!FUSED && loader && g_next < g_seam) issue(g_next);
    // ---- per-column tables (FUSED: they do not change between the orders of a launch; the flags do) ----
    {
        const int nt = blockDim.x;
        if (!FUSED || fu.first) {
            for (int zz = 0; zz < (MZ ? nz : kRingZones); ++zz) {
                const FixTab& src = g.fix[dg->fixtab[zz]];
                for (int i = tid; i < fixcap * kFixMaxSrc; i += nt) s_fixc[zz * fixcap * kFixMaxSrc + i] = src.C[i];
            }
            if (tid < kMaxZones) s_nfz[tid] = tid < nz ? dg->nfix[tid] : 0;
            const double* __restrict__ tau = a.tau + (size_t)b * L;
            for (int t = tid; t <= L; t += nt) s_hd[t] = (t == 0 || t == L) ? 0.0 : (tau[t] - tau[t - 1]) * 0.5;
            if (tid < 16) s_prmu[tid] = (tid > 0 && tid < N) ? 1.0 / g.mu[N + tid] : 0.0;
        }
        if (tid < 4) s_flag[tid] = 0;
        for (int i = tid; i < nwc * CR + NST + NST * nwc + (L + 31) / 32; i += nt) s_flagq[i] = 0;
        if (!FUSED && !WIDE) {                                  // (FUSED runs only while no |mu| < 0.01 lane keeps its k_smallmu value; WIDE reads them from memory)
            const double* __restrict__ In0 = a.In + (size_t)b * L * D;
            for (int i = tid; i < g.nsmall * L; i += nt) {
                const int k = i / L, t = i - k * L;
                s_S[i] = In0[(size_t)t * D + g.small_lanes[k]];
            }
        }
    }
    __syncthreads();
    stamp(1);
    if (FUSED && loader) {
        // every tile of this order's source function is stored (and acknowledged: order_loop.hip) before its count moves
        if (!fused_wait_ge(fu.jn_done, fu.jn_need, fu.abort) && lane == 0) s_flag[2] = 1;
        if (FUSED && wid == ncw && lane == 0) SOSRT_OL_STAMP(fu.log, fu.wg, fu.k, 2);      // source function complete
    }

    // =============================== downward ===============================
    if (loader) {
        // the chunks of the downward sweep and, across the seam, the first ones of the upward sweep
        load_chunks(g_next, g_seam, !FUSED);
        while (g_next < g_seam) g_next += NLOAD;
    } else {
        const int m = dirc_dn;
        const int vo = m * 8;
        const bool valid = valid_dn;                        // (of the downward direction, in this block)
        const double mu = g.mu[m];
        const bool tr = valid && m <= N - 2;
        const bool small = tr && fabs(mu) < kMuThreshold;       // spec:333
        const bool stdl = tr && !small;
        const double nrmu = stdl ? -1.0 / mu : 0.0;
        const bool has_small = wl && g.nsmall > 0;
        const int sbase = small ? (m - g.small_lanes[0]) * L : 0;   // the small lanes are consecutive directions
        const int lmT = lane_last - pT;                        // lane of direction N-1-pT in the wave that holds it
        auto zone_of = [&](int t) __attribute__((always_inline)) { return zr.of(t); };
        // (by shifts: a chain of selects over the three captured counts becomes a table of pointers on the stack, read back
        // with FLAT loads that wait for every global load and store in flight)
        const int nfix_packed = nfix0 | (nfix1 << 10) | (nfix2 << 20);
        // (zones beyond the third -- columns of more than one aerosol layer -- from the table in LDS)
        auto nfix_of = [&](int zz) __attribute__((always_inline)) { if constexpr (MZ) { if (zz > 2) return s_nfz[min(zz, kMaxZones - 1)]; }
            return (nfix_packed >> (10 * zz)) & 1023; };
        // one chunk: rows t0 .. t0 + TC - 1
        auto chunk = [&](auto special_t, auto mode_t, int q) __attribute__((always_inline)) {
            constexpr bool SP = decltype(special_t)::value;
            constexpr int MODE = decltype(mode_t)::value;
            const int t0 = q * TC;
            double Jc[TC], Ec[TC], Ic[TC], Jx_;
            SCAN_CB(q);
            take(special_t, q, vo, t0 * RB, RB, Jc, Ec, Ic, Jx_);
            SCAN_T0();
            const double Jprev = q > 0 ? Jx_ : 0.0;
            // extrapolation table of the zone (In_limit:113-141 as a linear map): chunk-local copies
            double c[kFixMaxSrc] = {0, 0, 0, 0, 0};
            double cT[kFixMaxSrc] = {0, 0, 0, 0, 0};           // row pT of the table: work item (uT, pT) rewrites direction N-1-pT
            int sl[kFixMaxSrc] = {0, 0, 0, 0, 0};
            int nfx = 0;
            bool fixlane = false;
            auto load_fix = [&](int zz) __attribute__((always_inline)) {
                const double* ftC = s_fixc + zz * fixcap * kFixMaxSrc;
                nfx = nfix_of(zz);
                const int ns = nfx < 2 ? 2 : (nfx < kFixMaxSrc ? nfx : kFixMaxSrc);     // In_limit:118-141
                fixlane = valid && nfx > 0 && m >= N - nfx;
                const int i = fixlane ? N - 1 - m : 0;
                const int s0 = nfx < 2 ? N - nfx - 2 : N - nfx - ns;
#pragma unroll
                for (int k = 0; k < kFixMaxSrc; ++k) {
                    if (SP || MODE == 2) c[k] = (fixlane && k < ns) ? ftC[i * ns + min(k, ns - 1)] : 0.0;     // (incl. the fast special form)
                    if (!SP) cT[k] = (pT < nfx && k < ns) ? ftC[pT * ns + min(k, ns - 1)] : 0.0;
                    sl[k] = s0 + min(k, ns - 1) - dn_base;            // lane of the source direction
                }
            };
            if (wl && ((SP && MODE != 3) || (!SP && MODE != 0))) load_fix(zone_of(t0));
            double cc[TC], v[TC], Sc[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) Sc[u] = 0;
            if (((SP && MODE != 3) || MODE == 2) && has_small) {
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    // (WIDE: from the field in memory, where k_smallmu left them -- this thread reads its element of a row before
                    // it stores that element at the end of the chunk)
                    const double sv = WIDE ? (small ? bload(rIn, vo, min(t0 + u, L - 1) * RB) : 0.0) : s_S[sbase + min(t0 + u, L - 1)];
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
                // chunk-local form, then the carried value
                double dl[TC], pl[TC];
                {
                    double d = 0, p = 1;
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        d = rec_step(d, Ec[u], cc[u]);
                        p = u == 0 ? Ec[0] : rec_hr(p, Ec[u]);
                        dl[u] = d; pl[u] = p;
                    }
                    // (the chunk-local values exist before the wave starts polling: the compiler would otherwise sink their
                    // computation below the wait, onto the serial path of the chain)
                    asm volatile("" ::"v"(d), "v"(p));
                }
                SCAN_T1();
                const double Din = q == 0 ? 0.0 : wait_carry(q);
                SCAN_T2();
                publish(q + 1, rec_step(Din, pl[TC - 1], dl[TC - 1]));      // on at once: the next chunk's wave waits for it
#pragma unroll
                for (int u = 0; u < TC; ++u) v[u] = rec_step(Din, pl[u], dl[u]);
                if (MODE == 2 && has_small) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) v[u] = rec_add(v[u], Sc[u]);
                }
                // Up to 8 rewritten directions: the chunk goes through LDS and work item (uT, pT) does row uT, direction N-1-pT.
                const bool tfix = MODE == 1 || (MODE == 2 && wl && nfx > 0 && nfx <= 8 && sl[0] >= xb_dn);   // sources inside the 16-lane window
                if (tfix) {
                    if (lane >= xb_dn && lane < xb_dn + 16) {
#pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            s_x[u * 16 + lane - xb_dn] = v[u];
                            if (ACC) s_xI[u * 16 + lane - xb_dn] = Ic[u];
                        }
                    }
                    double acc = 0;
#pragma unroll
                    for (int k = 0; k < kFixMaxSrc; ++k) acc = fix_acc(cT[k], xrow[max(sl[k] - xb_dn, 0)], acc);
                    const int mT = N - 1 - pT;
                    const double IcT = ACC ? s_xI[uT * 16 + min(max(lmT - xb_dn, 0), 15)] : 0.0;
                    if (pT < nfx) {
                        const int voT = (t0 + uT) * RB + mT * 8;
                        bstore_aux<STX>(rIn, voT, 0, acc);
                        if (ACC) bstore_aux<STX>(rI, voT, 0, IcT + acc);
                        if (SAVED) bstore_aux<STX>(rS, voT, 0, acc);
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
                        bstore_aux<STX>(rIn, vo, so, v[u]);
                        if (ACC) bstore_aux<STX>(rI, vo, so, Ic[u] + v[u]);
                        if (SAVED) bstore_aux<STX>(rS, vo, so, v[u]);
                    }
                }
            } else if (MODE == 3) {
                // A chunk with a zone boundary (or the last one) when every rewritten direction can go through the transposed
                // work items: the serial part is the recurrence alone -- the extrapolation is evaluated in line only for the
                // row at a zone end, whose final value is the state of the next zone (spec:359,378) -- and the carried value
                // leaves before the treatments and the stores.  Same operations per element as the general form below.
                SCAN_T1();
                double Dv = q == 0 ? 0.0 : wait_carry(q);
                SCAN_T2();
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const int t = t0 + u;
                    const double Dn = rec_step(Dv, Ec[u], cc[u]);
                    v[u] = Dn;
                    const bool zone_end = zr.ends(t);
                    double x = Dn;
                    if (zone_end && wl) {
                        load_fix(zone_of(t));
                        if (nfx > 0) {
                            double acc = 0;
#pragma unroll
                            for (int k = 0; k < kFixMaxSrc; ++k) acc = fix_acc(c[k], readlane_f64(Dn, sl[k]), acc);
                            x = fixlane ? acc : Dn;
                        }
                    }
                    Dv = t < L ? (zone_end ? (stdl ? x : 0.0) : Dn) : Dv;
                }
                if (q + 1 < NCH) publish(q + 1, Dv);
                if (wl) {
                    if (lane >= xb_dn && lane < xb_dn + 16) {
#pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            s_x[u * 16 + lane - xb_dn] = v[u];
                            if (ACC) s_xI[u * 16 + lane - xb_dn] = Ic[u];
                        }
                    }
                    // work item (uT, pT): row t0 + uT with the table of that row's zone
                    const int tT = t0 + uT;
                    const int zT = zone_of(min(tT, L - 1)), nfT = nfix_of(zT);
                    const int nsT = nfT < 2 ? 2 : (nfT < kFixMaxSrc ? nfT : kFixMaxSrc);
                    const int s0T = nfT < 2 ? N - nfT - 2 : N - nfT - nsT;
                    const double* ftT = s_fixc + zT * fixcap * kFixMaxSrc;
                    double acc = 0;
#pragma unroll
                    for (int k = 0; k < kFixMaxSrc; ++k) {
                        const double ck = (pT < nfT && k < nsT) ? ftT[pT * nsT + min(k, nsT - 1)] : 0.0;
                        const int sk = s0T + min(k, nsT - 1) - dn_base - xb_dn;
                        acc = fix_acc(ck, xrow[max(sk, 0)], acc);
                    }
                    const int mT = N - 1 - pT;
                    const double IcT = ACC ? s_xI[uT * 16 + min(max(lmT - xb_dn, 0), 15)] : 0.0;
                    if (pT < nfT && tT < L) {
                        const int voT = tT * RB + mT * 8;
                        bstore_aux<STX>(rIn, voT, 0, acc);
                        if (ACC) bstore_aux<STX>(rI, voT, 0, IcT + acc);
                        if (SAVED) bstore_aux<STX>(rS, voT, 0, acc);
                        if (tT == L - 1) {
                            s_sfc[mT] = acc;
                            s_conv[0 * ND + mT] = acc;
                            s_conv[1 * ND + mT] = IcT + acc;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const int t = t0 + u;
                    const int nfu = wl ? nfix_of(zone_of(min(t, L - 1))) : 0;
                    const bool fixu = nfu > 0 && m >= N - nfu;
                    if (valid && !fixu && t < L) {
                        const int so = t * RB;
                        bstore_aux<STX>(rIn, vo, so, v[u]);
                        if (ACC) bstore_aux<STX>(rI, vo, so, Ic[u] + v[u]);
                        if (SAVED) bstore_aux<STX>(rS, vo, so, v[u]);
                        if (t == L - 1) {
                            s_sfc[dir_dn] = v[u];
                            s_conv[0 * ND + dir_dn] = v[u];
                            s_conv[1 * ND + dir_dn] = Ic[u] + v[u];
                        }
                    }
                }
            } else {
                // serial form (as the ring kernel): zone boundaries, last chunk
                SCAN_T1();
                double Dv = q == 0 ? 0.0 : wait_carry(q);
                SCAN_T2();
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
                    const bool zone_end = zr.ends(t);     // the next zone starts from the final row (spec:359,378)
                    Dv = t < L ? (zone_end ? (stdl ? x : 0.0) : Dn) : Dv;
                    if (valid && t < L) {
                        const int so = t * RB;
                        bstore_aux<STX>(rIn, vo, so, x);
                        if (ACC) bstore_aux<STX>(rI, vo, so, Ic[u] + x);
                        if (SAVED) bstore_aux<STX>(rS, vo, so, x);
                    }
                }
                if (q + 1 < NCH) publish(q + 1, Dv);
                if (t0 + TC >= L) {
#pragma unroll
                    for (int u = 0; u < TC; ++u)
                        if (t0 + u == L - 1 && valid) {
                            s_sfc[dir_dn] = v[u];
                            s_conv[0 * ND + dir_dn] = v[u];
                            s_conv[1 * ND + dir_dn] = Ic[u] + v[u];
                        }
                }
            }
            SCAN_T3();
        };
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        using M2 = std::integral_constant<int, 2>;
        using M3 = std::integral_constant<int, 3>;
        for (int q = grp; q < NCH; q += SW) {
            if (special_dn(q)) {
                // fast form when every zone the chunk touches has at most 8 rewritten directions with their sources inside the
                // 16-lane window (and no |mu| < 0.01 lane keeps its k_smallmu value)
                bool fastsp = !has_small;
                if (wl) {
                    const int za = zone_of(q * TC), zb = zone_of(min(q * TC + TC - 1, L - 1));
                    for (int zz = za; zz <= zb; ++zz) {
                        const int nf = nfix_of(zz);
                        const int ns = nf < 2 ? 2 : (nf < kFixMaxSrc ? nf : kFixMaxSrc);
                        const int sl0 = (nf < 2 ? N - nf - 2 : N - nf - ns) - dn_base;
                        fastsp = fastsp && nf <= 8 && (nf == 0 || sl0 >= xb_dn);
                    }
                }
#ifdef SOSRT_SCAN_GENERAL_SPECIAL
                fastsp = false;
#endif
                if (fastsp) chunk(std::true_type{}, M3{}, q);
                else chunk(std::true_type{}, M2{}, q);
            } else {
                // MODE: 0 this wave has no treated lane here, 1 the transposed extrapolation alone, 2 anything else
                const int nfx = nfix_of(zone_of(q * TC));
                const int ns = nfx < 2 ? 2 : (nfx < kFixMaxSrc ? nfx : kFixMaxSrc);
                const int sl0 = (nfx < 2 ? N - nfx - 2 : N - nfx - ns) - dn_base;
                const int mode = (!wl || (nfx == 0 && !has_small)) ? 0 : ((!has_small && nfx <= 8 && sl0 >= xb_dn) ? 1 : 2);
                if (mode == 0) chunk(std::false_type{}, M0{}, q);
                else if (mode == 1) chunk(std::false_type{}, M1{}, q);
                else chunk(std::false_type{}, M2{}, q);
            }
        }
    }
    stamp(2);
    if (FUSED && tid == 0) SOSRT_OL_STAMP(fu.log, fu.wg, fu.k, 3);                          // (wave 0) downward sweep done
    __syncthreads();                                            // the surface row is complete
    stamp(3);

    // =============================== surface ===============================
    if (surface == SOSRT_SURFACE_SPECULAR) {
        Bv = valid ? rho * s_sfc[N - 1 - dir] : 0.0;                // spec:397
    } else if (surface == SOSRT_SURFACE_LAMBERTIAN || surface == SOSRT_SURFACE_LAMBERTIAN_README) {
        // -2 rho trapz(In[L-1, rev] mu[rev], mu[rev]), rev = N-2 .. 0   (lam:399), descending abscissae
        double term = 0;
        if (dir <= N - 3) {
            const int k0 = N - 2 - dir, k1 = k0 - 1;
            const double x0 = g.mu[k0], x1 = g.mu[k1];
            term = (x1 - x0) * (s_sfc[k1] * x1 + s_sfc[k0] * x0) / 2;
        }
        const double ws = wave_sum_(term);
        if (grp == 0 && lane == 0) s_red[lg] = ws;
        __syncthreads();
        double S = 0;
        for (int i = 0; i < nwc; ++i) S += s_red[i];
        Bv = (surface == SOSRT_SURFACE_LAMBERTIAN_README ? 2 : -2) * rho * S;          // lam:399 as coded (negative), or README.md:215
    }

    // =============================== upward ===============================
    if (loader) {
        load_chunks(g_next, 2 * NCH, false);
    } else {
        const int mj = N + dirc;
        const int vo = mj * 8;
        const bool tr = valid && dir > 0;
        const double mu = mu_up;
        const double prmu = tr ? 1.0 / mu : 0.0;
        const int last_cand = min(N - 3, 61);
        bool notfound = false;
        // spec:401-409 for one row held across the first lane group: x is the raw row, returns the blended value
        // (no stop among the candidates of this lane group: the row stays raw and is finished after the sweep,
        // finish_flagged_rows -- unless it is the first row of a zone, whose blended value is the state of the zone above:
        // then the whole sweep is redone row by row)
        auto blend = [&](double x, int t) __attribute__((always_inline)) {
            const double x1 = lane_up1(x), x2 = lane_up1(x1);
            const bool stop = lane >= 1 && lane <= last_cand && !(fabs((x - x1) - (x1 - x2)) > 0.0001);
            const unsigned long long mk = __ballot(stop);
            const int kf = mk ? __ffsll((long long)mk) : 1;
            if (mk == 0) {
                notfound = true;
                if (N - 3 > 61 && lane == 0 && t >= 0) {
                    if (zr.starts(t)) s_flag[0] = 1;
                    else { flag_row(s_nf, t); s_flag[3] = 1; }
                }
            }
            const double r0 = readlane_f64(x, 0), rk = readlane_f64(x, kf);
            const double w = blend_weight(mu, readlane_f64(prmu, kf));         // mu_m / mu_kf
            const double bl = blend_val(w, r0, rk);
            return (tr && dir < kf) ? bl : x;
        };
        // one chunk: rows t0, t0 - 1, ..., t0 - TC + 1
        // SP, MODE: plain chunk 0 (no mu -> 0+ lanes in this wave) / 1 (first lane group); chunk with a zone boundary or last
        // chunk: 1 general form, 2 / 3 fast form without / with the mu -> 0+ lanes
        auto chunk = [&](auto special_t, auto mode_t, int j) __attribute__((always_inline)) {
            constexpr bool SP = decltype(special_t)::value;
            constexpr int MODE = decltype(mode_t)::value;
            constexpr bool FAST = SP && MODE >= 2;
            constexpr int PM = FAST ? MODE - 2 : MODE;          // treatment of the rows: 0 none, 1 transposed blend
            const int t0 = L - 1 - j * TC;
            double Jc[TC], Ec[TC], Ic[TC], Jx_;
            SCAN_CB(NCH + j);
            take(special_t, NCH + j, vo, t0 * RB, -RB, Jc, Ec, Ic, Jx_);
            SCAN_T0();
            const double Jnext = j > 0 ? Jx_ : 0.0;
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
            if (!SP || FAST) {
                if (!SP) {
                    double dl[TC], pl[TC];
                    {
                        double d = 0, p = 1;
#pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            d = rec_step(d, Ec[u], cc[u]);
                            p = u == 0 ? Ec[0] : rec_hr(p, Ec[u]);
                            dl[u] = d; pl[u] = p;
                        }
                        asm volatile("" ::"v"(d), "v"(p));      // (before the wave starts polling, as in the downward sweep)
                    }
                SCAN_T1();
                    const double Uin = j == 0 ? Bv : wait_carry(NCH + j);
                SCAN_T2();
                    publish(NCH + j + 1, rec_step(Uin, pl[TC - 1], dl[TC - 1]));      // on at once: the next chunk's wave waits for it
#pragma unroll
                    for (int u = 0; u < TC; ++u) v[u] = rec_step(Uin, pl[u], dl[u]);
                } else {
                    // fast form of a chunk with a zone boundary (or the last one): the serial part is the recurrence alone; the
                    // search and blend are evaluated in line only for the first row of a zone, whose blended value is the
                    // state of the zone above; the carried value leaves before the treatments and the stores
                SCAN_T1();
                    double U = j == 0 ? Bv : wait_carry(NCH + j);
                SCAN_T2();
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        const int t = t0 - u;
                        const double Un = rec_step(U, Ec[u], cc[u]);
                        v[u] = Un;
                        const bool zone_start = zr.starts(t) != 0;
                        double x = Un;
                        if (zone_start && w0) x = blend(dir == 0 ? Jc[u] : Un, t);
                        U = t >= 0 ? ((zone_start && tr) ? x : Un) : U;
                    }
                    if (j + 1 < NCH) publish(NCH + j + 1, U);
                }
                // spec:401-409: the chunk goes through LDS and work item (uT, pT) tests candidate k = pT+1 of row uT, then
                // produces direction N+k of that row (blended below the stop, raw above it).  A row whose search goes
                // beyond 8 candidates sends the chunk through the row-by-row path.
                bool tblend = false;
                if (PM == 1) {
#pragma unroll
                    for (int u = 0; u < TC; ++u) v[u] = dir == 0 ? Jc[u] : v[u];                  // spec:401
                    if (lane < 16) {
#pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            s_x[u * 16 + lane] = v[u];
                            if (ACC) s_xI[u * 16 + lane] = Ic[u];
                        }
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
                        const double IcT = ACC ? s_xI[uT * 16 + k] : 0.0;
                        if (k < N && (!SP || t0 - uT >= 0)) {
                            const int voT = (t0 - uT) * RB + (N + k) * 8;
                            bstore_aux<STX>(rIn, voT, 0, val);
                            if (ACC) bstore_aux<STX>(rI, voT, 0, IcT + val);
                            if (SAVED) bstore_aux<STX>(rS, voT, 0, val);
                            if (SP && t0 - uT == 0) {
                                s_conv[2 * ND + k] = val;
                                s_conv[3 * ND + k] = IcT + val;
                            }
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < TC; ++u) {
                            const double xb_ = blend(v[u], t0 - u);
                            if (valid && (!SP || t0 - u >= 0)) {
                                const int so = (t0 - u) * RB;
                                bstore_aux<STX>(rIn, vo, so, xb_);
                                if (ACC) bstore_aux<STX>(rI, vo, so, Ic[u] + xb_);
                                if (SAVED) bstore_aux<STX>(rS, vo, so, xb_);
                                if (SP && t0 - u == 0) {
                                    s_conv[2 * ND + dir] = xb_;
                                    s_conv[3 * ND + dir] = Ic[u] + xb_;
                                }
                            }
                        }
                    }
                }
                if (valid && (PM == 0 || (tblend && !(dir >= 1 && dir <= 8)))) {
                    // SPLIT, the half without the mu -> 0+ lanes (part 1): its upward rows are what the OTHER workgroup reads back
                    // (x_old, running total) if that one has to redo the sweep row by row, and part 1 cannot know: it leaves
                    // them with write-through stores, so that they are in memory -- not dirty in this XCD's L2 -- when it arrives
                    constexpr int WT = (FUSED || (SPLIT && PM == 0)) ? 17 : 0;
#pragma unroll
                    for (int u = 0; u < TC; ++u) {
                        if (SP && t0 - u < 0) continue;
                        const int so = (t0 - u) * RB;
                        bstore_aux<WT>(rIn, vo, so, v[u]);
                        if (ACC) bstore_aux<WT>(rI, vo, so, Ic[u] + v[u]);
                        if (SAVED) bstore_aux<STX>(rS, vo, so, v[u]);
                        if (SP && t0 - u == 0) {
                            s_conv[2 * ND + dir] = v[u];
                            s_conv[3 * ND + dir] = Ic[u] + v[u];
                        }
                    }
                }
            } else {
                // general form (the ring kernel's; -DSOSRT_SCAN_GENERAL_SPECIAL builds only: the fast form covers every case)
                SCAN_T1();
                double U = j == 0 ? Bv : wait_carry(NCH + j);
                SCAN_T2();
#pragma unroll
                for (int u = 0; u < TC; ++u) {
                    const int t = t0 - u;
                    const double Un = rec_step(U, Ec[u], cc[u]);
                    double x = Un;
                    if (w0 && t >= 0) x = blend(dir == 0 ? Jc[u] : Un, t);
                    v[u] = x;
                    const bool zone_start = zr.starts(t) != 0;          // blended row feeds the zone above
                    U = t >= 0 ? ((zone_start && tr) ? x : Un) : U;
                    if (valid && t >= 0) {
                        const int so = t * RB;
                        bstore_aux<STX>(rIn, vo, so, x);
                        if (ACC) bstore_aux<STX>(rI, vo, so, Ic[u] + x);
                        if (SAVED) bstore_aux<STX>(rS, vo, so, x);
                    }
                }
                if (j + 1 < NCH) publish(NCH + j + 1, U);
                if (t0 - TC < 0) {
#pragma unroll
                    for (int u = 0; u < TC; ++u)
                        if (t0 - u == 0 && valid) {
                            s_conv[2 * ND + dir] = v[u];
                            s_conv[3 * ND + dir] = Ic[u] + v[u];
                        }
                }
            }
            SCAN_T3();
        };
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        using M2 = std::integral_constant<int, 2>;
        using M3 = std::integral_constant<int, 3>;
        // (the general form of the chunks with a zone boundary stays reachable: -DSOSRT_SCAN_GENERAL_SPECIAL builds)
        for (int j = grp; j < NCH; j += SW) {
            if (special_up(j)) {
#ifdef SOSRT_SCAN_GENERAL_SPECIAL
                chunk(std::true_type{}, M1{}, j);
#else
                if (!w0) chunk(std::true_type{}, M2{}, j);
                else chunk(std::true_type{}, M3{}, j);
#endif
            }
            else if (!w0) chunk(std::false_type{}, M0{}, j);
            else chunk(std::false_type{}, M1{}, j);
        }
        if (w0 && notfound && lane == 0 && N - 3 <= 61) s_flag[1] = 1;     // every candidate was in this lane group: IndexError
    }
    stamp(4);
    if (FUSED && tid == 0) SOSRT_OL_STAMP(fu.log, fu.wg, fu.k, 4);                          // (wave 0) upward sweep done
    // FUSED: the rows of this order are read by other workgroups as soon as the column's verdict is out: every wave's stores
    // (write-through) are acknowledged before the barrier that the publishing thread passes
    if (FUSED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(5);
    stamp(6);
    stamp(7);
    stamp(8);
    stamp(9);
    __shared__ int s_verdict;
    // FUSED: the column's verdict for the other workgroups of the launch -- its rows are final (ord_done), it stops (col_stop,
    // written first) -- and for the caller.  Every call site is workgroup-uniform.
    auto leave = [&](int stop) __attribute__((always_inline)) -> int {
        if constexpr (FUSED) {
            const int st = (stop || fu.last) ? 1 : 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the rows the end of the order rewrote (finishing, redo)
            __syncthreads();
            if (tid == 0) {
                SOSRT_OL_STAMP(fu.log, fu.wg, fu.k, 5);                                     // verdict about to be published
                if (fu.orders) __hip_atomic_fetch_add(fu.orders, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (st) __hip_atomic_store(fu.col_stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(fu.ord_done, fu.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return st;
        } else {
            return stop;
        }
    };
    if (SPLIT) {
        // The two workgroups of the column meet here without waiting: each leaves its test rows, its surface row and its
        // flags in global memory, and the one that arrives second runs the rest for the whole column.  The few exchanged
        // words go as device-scope atomics (performed at the coherence point: the workgroups may sit on different XCDs, whose
        // L2s are not coherent), acknowledged before the arrival counter moves -- no release / acquire fence, which would
        // write back and invalidate a whole L2 under the other columns' feet (measured: 111 us per launch with fences, 45
        // with the atomics, at 128 live columns).  This rests on what gfx950 does -- an agent-scope (sc1) store is performed
        // at memory before its vmcnt acknowledgement, and an agent-scope load misses the local L2 for lines another XCD may
        // own -- not on the C++ memory model, which would ask for release / acquire on the counter.
        // The row-by-row redo reads the OTHER half's field rows (x_old and the running total, to correct it).  Whichever
        // workgroup runs it, those rows must be in memory by then: part 0 (the only one that can ask for the redo) writes
        // its L2 back with a release fence in that rare case; part 1, which cannot know, stores its upward rows
        // write-through in the first place (round 2 left them dirty in its L2: part 0 arriving last read stale rows).
        double* gs = a.scan_scratch + (size_t)b * kScanScratch;            // [4 test rows + surface row][kScanDirs]
        int* sync = a.scan_sync + 2 * b;                                   // {arrivals, flags}
        int* gnf = reinterpret_cast<int*>(gs + 5 * kScanDirs);             // [32] bit mask of the flagged rows
        __shared__ int s_last;
        if (wid == 0) {                                                    // (lanes = this workgroup's directions)
            if (valid_dn) {
                __hip_atomic_store(gs + 0 * kScanDirs + dir_dn, s_conv[0 * ND + dir_dn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gs + 1 * kScanDirs + dir_dn, s_conv[1 * ND + dir_dn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gs + 4 * kScanDirs + dir_dn, s_sfc[dir_dn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (valid) {
                __hip_atomic_store(gs + 2 * kScanDirs + dir, s_conv[2 * ND + dir], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gs + 3 * kScanDirs + dir, s_conv[3 * ND + dir], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int f = (s_flag[0] ? 1 : 0) | (s_flag[1] ? 2 : 0) | (s_flag[2] ? 4 : 0) | (s_flag[3] ? 8 : 0);
            if (s_flag[3] && lane < (L + 31) / 32)                         // (part 0 only: it holds the mu -> 0+ lanes) the flagged rows
                __hip_atomic_store(gnf + lane, s_nf[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f && lane == 0) __hip_atomic_fetch_or(sync + 1, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // acknowledged: the words are where the other workgroup reads them
        }
        // part 1's write-through field rows of the upward sweep (see the stores): acknowledged by every wave before the arrival
        if (part != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (part 0 is the only one that can raise the flag; it writes its own rows back with a fence, once, in that rare case)
        if (s_flag[0] || s_flag[3]) __atomic_thread_fence(__ATOMIC_RELEASE);   // (uniform) the redo / the finishing will read this half's field rows
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nparts - 1;
        __syncthreads();
        if (!s_last) {
            if constexpr (FUSED) {
                // the workgroup that arrived last decides for the column; the others learn the verdict from it
                if (tid == 0) {
                    const bool ok = fused_wait_ge(fu.ord_done, fu.k, fu.abort);
                    s_verdict = (!ok || __hip_atomic_load(fu.col_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1 : 0;
                }
                __syncthreads();
                return s_verdict;
            }
            return 0;
        }
        if (tid == 0) {
            const int f = __hip_atomic_exchange(sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_flag[0] = f & 1; s_flag[1] = (f >> 1) & 1; s_flag[2] = (f >> 2) & 1; s_flag[3] = (f >> 3) & 1;
            __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next order
        }
        __syncthreads();
        if (s_flag[0] || s_flag[3]) __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (uniform, rare) the other half's field rows
        if (s_flag[3] && tid < (L + 31) / 32) s_nf[tid] = __hip_atomic_load(gnf + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = tid; i < N; i += blockDim.x) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_conv[k * ND + i] = __hip_atomic_load(gs + k * kScanDirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_sfc[i] = __hip_atomic_load(gs + 4 * kScanDirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (s_flag[2]) {                                                    // a carried value never arrived (internal error)
        if (tid == 0) {
            a.cv.status[b] = SOSRT_COL_INTERNAL;
            if (ACC) { a.cv.active[b] = 0; a.cv.norders[b] = a.order; atomicSub(a.cv.nactive, 1); }
        }
        return leave(1);
    }
    if (s_flag[1]) {                                                    // the reference raises IndexError (spec:404)
        if (tid == 0) {
            a.cv.status[b] = SOSRT_COL_INDEXERROR;
            if (ACC) { a.cv.active[b] = 0; a.cv.norders[b] = a.order; atomicSub(a.cv.nactive, 1); }
        }
        return leave(1);
    }
    // threads tid < N (the waves of chunk residue 0): direction tid
    const bool act = tid < N;
    double rup_v = act ? s_conv[2 * ND + tid] : 0.0, rup_i = act ? s_conv[3 * ND + tid] : 1.0;
    const double rdn_v = act ? s_conv[0 * ND + tid] : 0.0, rdn_i = act ? s_conv[1 * ND + tid] : 1.0;
    if (!s_flag[0] && s_flag[3]) {
        // rows whose search went past the lanes of the first lane group: finished one by one (the stages are free by now)
        __syncthreads();
        if (finish_flagged_rows<ACC, SAVED, LDX, STX>(s_nf, L, N, RB, g.mu, rIn, rI, rS, s_stage, rup_v, rup_i)) {
            if (tid == 0) {                                             // the reference raises IndexError (spec:404)
                a.cv.status[b] = SOSRT_COL_INDEXERROR;
                if (ACC) { a.cv.active[b] = 0; a.cv.norders[b] = a.order; atomicSub(a.cv.nactive, 1); }
            }
            return leave(1);
        }
    }
    if (s_flag[0]) {
        // The first row of a zone has no stop among the lanes of the first lane group, and its blended value is the state of
        // the zone above (spec:403-406 has no bound): redo that
        // sweep here, row by row (redo_upward_sweep).
        // (one workgroup per column: thread tid < N holds direction tid; split: from the surface row, specular or none)
        const double U0 = SPLIT ? ((act && surface == SOSRT_SURFACE_SPECULAR) ? rho * s_sfc[N - 1 - tid] : 0.0) : Bv;
        __syncthreads();
        const bool missing = redo_upward_sweep<ACC, SAVED, MZ, LDX, STX>(L, N, RB, zr, s_hd, g.mu, rJ, rE, rIn, rI, rS, U0,
                                                            s_stage, rup_v, rup_i);
        if (missing) {                                                  // the reference raises IndexError (spec:404)
            if (tid == 0) {
                a.cv.status[b] = SOSRT_COL_INDEXERROR;
                if (ACC) { a.cv.active[b] = 0; a.cv.norders[b] = a.order; atomicSub(a.cv.nactive, 1); }
            }
            return leave(1);
        }
    }
    if (ACC) {
        const double ra = block_pymax_(rup_v / rup_i, act, s_red, 0);
        const double rb = block_pymax_(rdn_v / rdn_i, act, s_red, 0);
        const double r = (rb > ra) ? rb : ra;                               // the outer max() of spec:309
        if (tid == 0) {
            a.cv.ratio[b] = r;
            a.cv.norders[b] = a.order;
            if (!(r >= a.cv.tol)) {
                a.cv.active[b] = 0;
                atomicSub(a.cv.nactive, 1);
            }
        }
        return leave(!(r >= a.cv.tol) ? 1 : 0);
    } else if (tid == 0 && a.cv.status) {
        a.cv.status[b] = SOSRT_COL_OK;
    }
    return leave(1);
}

inline int scan_fixcap(const Grid& g) { return (int)(0.06 * g.N) + 1; }
template <bool SPLIT, bool WIDE = false>
inline size_t scan_lds_bytes(const Grid& g, int nzcap = kRingZones) {
    using C = ScanCfg<SPLIT, WIDE>;
    const int nwc = SPLIT ? 1 : (g.N + 63) / 64, ncw = nwc * C::SW, nwaves = ncw + NLOAD, ND = (g.N + 63) / 64 * 64;
    const size_t doubles = (size_t)C::NST * C::STAGE + (size_t)nwc * CR * 64 + ND + (size_t)nzcap * scan_fixcap(g) * kFixMaxSrc + nwaves + 2 +
                           g.L + 1 + (WIDE ? 0 : (size_t)g.nsmall * g.L) + 16 + 4 * ND + (size_t)ncw * 2 * TC * 16;
    return doubles * sizeof(double) + ((size_t)(nwc * CR + C::NST + C::NST * nwc + (g.L + 31) / 32 + kMaxZones) * sizeof(int) + 7) / 8 * 8;
}

}  // namespace
}  // namespace sosrt
