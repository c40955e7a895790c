// Device helpers shared by the wave-independent transport kernels (transport_fast.hip,
// transport_ring.hip): cross-lane reads, raw buffer addressing, Python's max() over a row.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace sosrt {
namespace {

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Buffer addressing (raw buffer ops on a per-column descriptor): the row offset lives in an SGPR,
// the lane offset in one VGPR that never changes, so a load or store costs no address arithmetic.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
// the same past the L1 (glc): for data this workgroup stored earlier in the kernel
__device__ __forceinline__ double bload_glc(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 1);
    return __hiloint2double((int)v.y, (int)v.x);
}
// the same with a cache policy (aux bits of the raw buffer intrinsics on gfx940+: 1 = sc0, 16 = sc1): 16 = an agent-scope load,
// past this CU's L1 -- for rows another workgroup of the same launch has stored write-through (bstore_aux<17>), or this
// workgroup itself an order ago (order_loop.hip)
template <int AUX>
__device__ __forceinline__ double bload_aux(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x) {
    u32x2 v;
    v.x = (unsigned)__double2loint(x);
    v.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
// the same with a cache policy (aux bits of the raw buffer intrinsics on gfx940+: 1 = sc0, 16 = sc1; 17 = both: a write-through
// store, performed at the memory the other XCDs' L2s read from, acknowledged (vmcnt) once it is there)
template <int AUX>
__device__ __forceinline__ void bstore_aux(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x) {
    u32x2 v;
    v.x = (unsigned)__double2loint(x);
    v.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, AUX);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// two adjacent doubles per lane (16 bytes): half the vector-memory instructions of the 8-byte forms
__device__ __forceinline__ double2 bload2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}
template <int AUX>
__device__ __forceinline__ double2 bload2_aux(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}
__device__ __forceinline__ void bstore2(__amdgpu_buffer_rsrc_t r, int voff, int soff, double2 x) {
    u32x4 v;
    v.x = (unsigned)__double2loint(x.x); v.y = (unsigned)__double2hiint(x.x);
    v.z = (unsigned)__double2loint(x.y); v.w = (unsigned)__double2hiint(x.y);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}

// value of lane + 1 (wave_shl:1); lane 63 keeps its own
__device__ __forceinline__ double lane_up1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_fmax_(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The arithmetic of the sweeps, spelled out operation by operation (no contraction left to the compiler): the ring
// kernel and the pipeline kernel must produce the same bits, because which of them transports a column depends on
// how many columns of its batch are still live.
//   source term of a row (spec:336-340, 393-399):  (h / |mu|) (Ja E + Jb),  h = half the layer thickness
__device__ __forceinline__ double rec_src(double h_rmu, double Ja, double E, double Jb) {
#pragma clang fp contract(off)
    return h_rmu * __builtin_fma(Ja, E, Jb);
}
__device__ __forceinline__ double rec_hr(double h, double rmu) {
#pragma clang fp contract(off)
    return h * rmu;
}
//   one step of the recurrence:  S E + c
__device__ __forceinline__ double rec_step(double S, double E, double c) { return __builtin_fma(S, E, c); }
__device__ __forceinline__ double rec_add(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
//   spec:407-409:  (1 - w) r0 + w rk,  w = mu_m / mu_k as mu_m * (1 / mu_k)
__device__ __forceinline__ double blend_weight(double mu_m, double rmu_k) {
#pragma clang fp contract(off)
    return mu_m * rmu_k;
}
__device__ __forceinline__ double blend_val(double w, double r0, double rk) {
#pragma clang fp contract(off)
    const double a = (1 - w) * r0;
    return __builtin_fma(w, rk, a);
}
//   In_limit:113-141 as a linear map: acc + c x
__device__ __forceinline__ double fix_acc(double c, double x, double acc) { return __builtin_fma(c, x, acc); }

// ---- the zone table as the sweeps look at it -------------------------------------------------------------------------------
// Which rows end a zone that has another below it, which rows start a zone other than the first, which zone a row is in.  The
// reference's three zones (spec:113-449) are two boundaries held in scalars -- the sweeps of a (clear, slab, clear) column
// execute exactly what they did when the kernels knew nothing else.  MZ is the instantiation for a batch that holds a column of
// more aerosol layers (up to kMaxZones zones): the first rows of zones 3 .. 7 are scalars as well (-99 past the last
// zone), every test is a few more scalar compares, and a three-zone column of such a batch gets the bits it has anywhere else.
// Without MZ those compares are not compiled at all (as run-time branches they cost a lone column 1.5 us per order).
template <bool MZ> struct ZoneRows {
    const ColDesc* dg;
    int nz, zend0, zend1, zbeg1, zbeg2;
    int zb[MZ ? kMaxZones : 1];       // MZ: first rows of zones 3 .. (indices below 3 unused), -99 past the last zone: scalars too
    __device__ __forceinline__ explicit ZoneRows(const ColDesc* d) : dg(d) {
        nz = d->nz;
        zend0 = nz > 1 ? d->r1[0] : -9; zend1 = nz > 2 ? d->r1[1] : -9;
        zbeg1 = nz > 1 ? d->r0[1] : -9; zbeg2 = nz > 2 ? d->r0[2] : -9;
        if constexpr (MZ) {
#pragma unroll
            for (int k = 3; k < kMaxZones; ++k) zb[k] = k < nz ? d->r0[k] : -99;
        }
    }
    __device__ __forceinline__ bool ends(int t) const {           // last row of a zone that is followed by another
        bool e = t == zend0 || t == zend1;
        if constexpr (MZ) {
#pragma unroll
            for (int k = 3; k < kMaxZones; ++k) e = e || t == zb[k] - 1;
        }
        return e;
    }
    __device__ __forceinline__ int starts(int t) const {          // the zone (>= 1) whose first row is t, 0 if none
        int z = t == zbeg1 ? 1 : (t == zbeg2 ? 2 : 0);
        if constexpr (MZ) {
#pragma unroll
            for (int k = 3; k < kMaxZones; ++k) z = t == zb[k] ? k : z;
        }
        return z;
    }
    __device__ __forceinline__ int of(int t) const {              // the zone of row t (t may differ between lanes)
        int z = (zbeg2 >= 0 && t >= zbeg2) ? 2 : ((zbeg1 >= 0 && t >= zbeg1) ? 1 : 0);
        if constexpr (MZ) {
#pragma unroll
            for (int k = 3; k < kMaxZones; ++k) z = (zb[k] >= 0 && t >= zb[k]) ? k : z;
        }
        return z;
    }
    __device__ __forceinline__ int nfix(int zz) const { return dg->nfix[zz]; }
    // the same for sweeps of more than 64 chunks: bits of the chunks 64 word .. 64 word + 63
    __device__ __forceinline__ unsigned long long boundary_chunks(int L, int tc, bool up, int word) const {
        unsigned long long m = 0;
        if constexpr (!MZ) {
            const int rows[4] = {zend0, zend1, zbeg1, zbeg2};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = (up ? L - 1 - rows[i] : rows[i]) / tc;
                if (rows[i] >= 0 && (c >> 6) == word) m |= 1ull << (c & 63);
            }
        } else {
            for (int k = 1; k < nz; ++k) {
                const int rb = dg->r0[k], re = rb - 1;            // first row of zone k, last row of zone k - 1
                const int cb = (up ? L - 1 - rb : rb) / tc, ce = (up ? L - 1 - re : re) / tc;
                if ((cb >> 6) == word) m |= 1ull << (cb & 63);
                if ((ce >> 6) == word) m |= 1ull << (ce & 63);
            }
        }
        return m;
    }
    // bit q of the result: chunk q of a sweep (TC rows; up: counted from the last row) contains a zone boundary
    __device__ __forceinline__ unsigned long long boundary_chunks(int L, int tc, bool up) const {
        unsigned long long m = 0;
        if constexpr (!MZ) {
            const int rows[4] = {zend0, zend1, zbeg1, zbeg2};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (rows[i] >= 0) m |= 1ull << ((up ? L - 1 - rows[i] : rows[i]) / tc);
        } else {
            for (int k = 1; k < nz; ++k) {
                const int rb = dg->r0[k], re = rb - 1;            // first row of zone k, last row of zone k - 1
                m |= 1ull << ((up ? L - 1 - rb : rb) / tc);
                m |= 1ull << ((up ? L - 1 - re : re) / tc);
            }
        }
        return m;
    }
};

// ---- rows of the upward sweep whose mu -> 0+ search leaves the first wave ------------------------------------------------
// spec:403-406 has no bound: `while |second difference| > 1e-4: idx++`.  The sweeps test the candidates that fit the first
// wave of a row (lanes 1 .. 61); a row with no stop there is left RAW in memory (nothing blended: kf = 1) and flagged in an LDS
// bit mask, and is finished here after the sweep, on its own: the recurrence does not depend on the blend (only the first row of
// a zone hands its BLENDED value to the zone above, SURVEY H5 -- such a row takes the full row-by-row redo instead), so what
// remains for a flagged row is the search over all N directions, the blend of the directions below the stop and the correction
// of the running total, I += new - old.  Flagged rows go in batches of kNfBatch: their loads are issued together, then three
// barriers per batch.  (Round 2 redid the whole sweep row by row for any such row, 3 us per row: 620 us per launch for the EVA
// sweep, where 3 of 512 columns have a handful of such rows in the second order.)
constexpr int kNfBatch = 8;
// LDS doubles of the work area: kNfBatch rows of N + 2, the list of flagged rows, the stops, the count
__host__ __device__ inline size_t flagged_rows_work_doubles(int L, int N) { return (size_t)kNfBatch * (N + 2) + (L + kNfBatch + 2 + 1) / 2 + 1; }
__device__ __forceinline__ void flag_row(int* s_nf, int t) { atomicOr(&s_nf[t >> 5], 1 << (t & 31)); }
// All threads of the workgroup call it; thread tid < N is upward direction N + tid.  Returns true if some row has no stop at all
// (the reference raises IndexError, spec:404).
// (SOSRT_HELPER_INLINE: the cold helpers below MUST be inlined -- as calls they cost the transport kernels their register
// allocation: 168 VGPRs and spills reloaded inside the sweeps instead of 131 and none; -DSOSRT_HELPERS_AS_CALLS for A/B builds)
#ifdef SOSRT_HELPERS_AS_CALLS
#define SOSRT_HELPER_INLINE __attribute__((noinline))
#else
#define SOSRT_HELPER_INLINE __forceinline__
#endif
// LDX / STX: cache policy of the field rows' loads and stores (0: as the one-order kernels have it -- loads past the L1 with
// `glc`, plain stores; the order-loop kernel passes 16 / 17, see bload_aux)
template <bool ACC, bool SAVED, int LDX = 0, int STX = 0>
__device__ SOSRT_HELPER_INLINE bool finish_flagged_rows(const int* s_nf, int L, int N, int RB, const double* __restrict__ gmu, __amdgpu_buffer_rsrc_t rIn,
                                    __amdgpu_buffer_rsrc_t rI, __amdgpu_buffer_rsrc_t rS, double* s_work, double& rup_v, double& rup_i) {
    const int tid = threadIdx.x, NP = N + 2;
    double* s_rows = s_work;
    int* s_list = reinterpret_cast<int*>(s_work + kNfBatch * NP);
    int* s_kf = s_list + L;
    int* s_cnt = s_kf + kNfBatch;
    if (tid == 0) {
        int c = 0;
        for (int t = L - 1; t >= 0; --t)
            if ((s_nf[t >> 5] >> (t & 31)) & 1) s_list[c++] = t;
        *s_cnt = c;
    }
    __syncthreads();
    const int cnt = *s_cnt;
    const bool act = tid < N, tr = act && tid > 0;
    const int j = act ? tid : 0;
    const double mu = tr ? gmu[N + j] : 1.0;
    const int vo = (N + j) * 8;
    bool missing = false;
    for (int c0 = 0; c0 < cnt; c0 += kNfBatch) {
        const int nb = min(kNfBatch, cnt - c0);
        double xo[kNfBatch], Io[kNfBatch];
#pragma unroll
        for (int i = 0; i < kNfBatch; ++i) {                    // the rows as the sweep left them, past the L1
            const int t = s_list[c0 + min(i, nb - 1)];
            xo[i] = bload_aux<LDX ? LDX : 1>(rIn, vo, t * RB);
            Io[i] = ACC ? bload_aux<LDX ? LDX : 1>(rI, vo, t * RB) : 0.0;
        }
#pragma unroll
        for (int i = 0; i < kNfBatch; ++i)
            if (act) s_rows[i * NP + j] = xo[i];
        if (tid < kNfBatch) s_kf[tid] = 1 << 30;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kNfBatch; ++i) {
            const double* r = s_rows + i * NP;
            if (i < nb && act && j >= 1 && j <= N - 3 && !(fabs((r[j] - r[j + 1]) - (r[j + 1] - r[j + 2])) > 0.0001)) atomicMin(&s_kf[i], j);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kNfBatch; ++i) {
            if (i >= nb) continue;
            const double* r = s_rows + i * NP;
            const int ks = s_kf[i];
            missing |= ks == (1 << 30);
            const int kf = ks == (1 << 30) ? 1 : ks + 1;
            const int t = s_list[c0 + i];
            if (tr && tid < kf) {                               // spec:407-409
                const double x = blend_val(blend_weight(mu, 1.0 / gmu[N + kf]), r[0], r[kf]);
                bstore_aux<STX>(rIn, vo, t * RB, x);
                double It = 0;
                if (ACC) {
                    It = Io[i] + (x - xo[i]);
                    bstore_aux<STX>(rI, vo, t * RB, It);
                }
                if (SAVED) bstore_aux<STX>(rS, vo, t * RB, x);
                if (t == 0) { rup_v = x; rup_i = It; }
            }
        }
        __syncthreads();
    }
    return missing;
}

// The whole upward sweep again, row by row, every direction a candidate of the search: for a column in which the FIRST ROW OF A
// ZONE has no stop among the lanes of the first wave -- its blended value is the state of the zone above (spec:411,417 after
// :391-409), so everything above it changes.  Thread tid < N is upward direction N + tid and carries the recurrence
// from U0 (the reflected surface row); a row goes through LDS for the search and the blend; what the sweep had stored is read
// back past the L1 to correct the running total, I += new - old.  The rows' loads do not depend on the recurrence: they are
// issued a chunk of kRedoRows rows ahead, so a row costs two barriers instead of two memory round trips (round 2: 3 us per row,
// 650 us for a 200-row column that the rest of its launch then waited for).
// s_work: kRedoRows-independent, 2 N + 4 doubles.  Returns true if some row has no stop at all (IndexError, spec:404).
constexpr int kRedoRows = 8;
template <bool ACC, bool SAVED, bool MZ, int LDX = 0, int STX = 0>
__device__ SOSRT_HELPER_INLINE bool redo_upward_sweep(int L, int N, int RB, const ZoneRows<MZ>& zr, const double* s_hd,
                                  const double* __restrict__ gmu, __amdgpu_buffer_rsrc_t rJ, __amdgpu_buffer_rsrc_t rE,
                                  __amdgpu_buffer_rsrc_t rIn, __amdgpu_buffer_rsrc_t rI, __amdgpu_buffer_rsrc_t rS, double U0,
                                  double* s_work, double& rup_v, double& rup_i) {
    const int tid = threadIdx.x;
    double* s_row = s_work;                                             // [2][N + 2] alternating rows
    int* s_kf = reinterpret_cast<int*>(s_work + 2 * (N + 2));          // [2]
    const bool act = tid < N, tr = act && tid > 0;
    const int j = act ? tid : 0;
    const double mu = tr ? gmu[N + j] : 1.0;
    const double prmu = tr ? 1.0 / mu : 0.0;
    const int vo = (N + j) * 8;
    double U = U0, Jnext = 0;
    bool missing = false;
    if (tid < 2) s_kf[tid] = 1 << 30;
    __syncthreads();
    int par = 0;
    for (int t0 = L - 1; t0 >= 0; t0 -= kRedoRows) {
        double Jc[kRedoRows], Ec[kRedoRows], xo[kRedoRows], Io[kRedoRows];
#pragma unroll
        for (int u = 0; u < kRedoRows; ++u) {
            const int t = max(t0 - u, 0);
            Jc[u] = bload_aux<LDX>(rJ, vo, t * RB);
            Ec[u] = bload(rE, vo, t * RB);
            xo[u] = bload_aux<LDX ? LDX : 1>(rIn, vo, t * RB);
            Io[u] = ACC ? bload_aux<LDX ? LDX : 1>(rI, vo, t * RB) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kRedoRows; ++u) {
            const int t = t0 - u;
            if (t < 0) break;                                           // (uniform)
            double* row = s_row + par * (N + 2);
            const double src = rec_src(rec_hr(s_hd[t + 1], prmu), Jnext, Ec[u], Jc[u]);
            const double Un = rec_step(U, Ec[u], zr.ends(t) ? 0.0 : src);   // first row of a zone: attenuate only (H4)
            if (act) row[j] = tid == 0 ? Jc[u] : Un;                    // spec:401
            __syncthreads();
            if (act && j >= 1 && j <= N - 3 && !(fabs((row[j] - row[j + 1]) - (row[j + 1] - row[j + 2])) > 0.0001)) atomicMin(&s_kf[par], j);
            if (tid == 0) s_kf[par ^ 1] = 1 << 30;                      // (the other row's slot: read two barriers ago)
            __syncthreads();
            const int ks = s_kf[par];
            missing |= ks == (1 << 30);
            const int kf = ks == (1 << 30) ? 1 : ks + 1;
            double x = act ? row[j] : 0.0;
            if (tr && tid < kf) x = blend_val(blend_weight(mu, 1.0 / gmu[N + kf]), row[0], row[kf]);
            const bool zone_start = zr.starts(t) != 0;                  // blended row feeds the zone above (H5)
            U = (zone_start && tr) ? x : Un;
            Jnext = Jc[u];
            if (act) {
                bstore_aux<STX>(rIn, vo, t * RB, x);
                double It = 0;
                if (ACC) {
                    It = Io[u] + (x - xo[u]);
                    bstore_aux<STX>(rI, vo, t * RB, It);
                }
                if (SAVED) bstore_aux<STX>(rS, vo, t * RB, x);
                if (t == 0) { rup_v = x; rup_i = It; }
            }
            par ^= 1;
        }
    }
    __syncthreads();
    return missing;
}

// Python's max() over a row (see block_pymax in kernels.hip); first_tid holds element 0.
__device__ __forceinline__ double block_pymax_(double x, bool valid, double* s_red, int first_tid) {
    const int tid = threadIdx.x, nw = blockDim.x >> 6;
    double v = wave_fmax_(valid ? x : __builtin_nan(""));
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    if (tid == first_tid) s_red[nw] = x;
    __syncthreads();
    double r = s_red[0];
    for (int i = 1; i < nw; ++i) r = fmax(r, s_red[i]);
    const double first = s_red[nw];
    return (first != first) ? first : r;
}

}  // namespace
}  // namespace sosrt
