// The tile of the source-function contraction (jn_gemm.hip: the kernels of one order; order_loop.hip: the contraction as one
// role of the launch that keeps a few columns for several orders).
#pragma once
#include "kernels.hpp"
#include "transport_util.hpp"

namespace sosrt {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int A_LD = GEMM_KC + 2;
constexpr int B_LD = GEMM_BN + 16;

// RT = MFMA row tiles per wave: 4 (64 rows) for the plain rows, 2 (32 rows) for the slab rows, whose
// two passes over k would otherwise make their workgroups the critical path of the launch.
// local row of a tile list -> row of the batch (-1 beyond the end)
struct ListRows {               // the host-built row lists over all columns (null list: identity)
    const int* rows;
    int n;
    __device__ int operator()(int lr) const { return lr < n ? (rows ? rows[lr] : lr) : -1; }   // a list may hold -1 (padding)
};
struct ColumnRows {             // the plain or the slab rows of one column
    int base, iu, ns, n;        // b*L, first slab row, slab rows, rows of this kind
    bool slab;
    __device__ int operator()(int lr) const {
        if (lr >= n) return -1;
        return base + (slab ? iu + lr : (lr < iu ? lr : lr + ns));
    }
};

// wmix (slab tiles of one column only): the column's combined matrix ca W_atm + cr W_aer -- one pass over k
// with unit coefficients instead of two passes
//
// SYM (flip-symmetric matrices, W[D-1-k][D-1-m] = W[k][m] -- every phase function of the scattering angle on a
// symmetric direction grid): with a_k = In_1[k], b_k = In_1[D-1-k], u = a + b, v = a - b (k < N),
//     Jn[m] = X + Y,  Jn[D-1-m] = X - Y,   X = sum_k u_k S[k][m],  Y = sum_k v_k A[k][m]   (m < N)
// S = (W[k][m] + W[D-1-k][m]) / 2, A = (W[k][m] - W[D-1-k][m]) / 2 (k_symfold): two N x N products instead of one
// D x D -- half the flops.  The workgroup's 128 columns are 64 values of m, X and Y each; a wave keeps X in its
// first column tile and Y in its second.  The matrices are stored [k][S: 0..Wld/2 | A: Wld/2..Wld].
//
// COH (order_loop.hip: the contraction as one role of a launch that also holds the transport of the same columns, on other
// workgroups): the rows of In_1 were stored write-through by workgroups of this launch and are loaded `sc1` (past this CU's L1,
// bload_aux), the rows of Jn are stored write-through for the transport's loaders; both through buffer descriptors of the
// tile's column (RowOf = ColumnRows).  The arithmetic is the same.
//
// ASTAGE (with COH, DEEP): the tile's rows of In_1 are brought into LDS WHOLE, at once, by LDS-DMA (`sRaw`: 16 RT rows of D + 2
// doubles), and the k-loop takes its A operand from there.  A workgroup of the order-loop launch is alone on its CU and its A
// rows come from memory (they were stored write-through moments ago): staged two k-chunks ahead in registers, every pair of
// chunks waited for a memory round trip of its own -- 12.5 us per tile of 8 chunks, measured -- where one round trip serves.
template <int RT, bool SLAB, bool DEEP = false, bool SYM = false, class RowOf = ListRows, bool COH = false, bool ASTAGE = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, double* sA, double* sB, int* s_any, int tile, int bn0,
                                          RowOf row_of, bool check_active, const double* __restrict__ wmix = nullptr,
                                          double* sRaw = nullptr) {
    static_assert(!ASTAGE || (COH && DEEP), "ASTAGE is a form of the order-loop launch's tile");
    constexpr int BM = 16 * RT;
    double* const sAv = sA + BM * A_LD;          // SYM: the v operand next to the u operand
    const int Nn = g.D >> 1, Nh = g.Wld >> 1;
    // (COH: the tile sits inside the tile / order loops of order_loop.hip; an opaque copy of the thread id keeps the compiler from
    // hoisting its lane constants out of those loops and holding them in registers across every tile)
    int tid_ = threadIdx.x;
    if (COH) asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int D = g.D, Dp = g.Dp, Wld = g.Wld;
    constexpr bool slab = SLAB;
    const int bm0 = tile * BM;

    // the row this thread stages, its coefficients; skip the tile when every column it touches has converged
    const int arow = tid >> 2, akq = (tid & 3) * (GEMM_KC / 4);
    int grow = -1;
    if (arow < BM) grow = row_of(bm0 + arow);
    if (check_active && g.active && g.check_tiles) {
        if (tid == 0) *s_any = 0;
        __syncthreads();
        if ((tid & 3) == 0 && grow >= 0 && g.active[grow / g.L]) *s_any = 1;
        __syncthreads();
        if (!*s_any) return;
    }
    // the tile's row ids for the epilogue (the lists are read once, here)
    __shared__ int s_rowid[16 * (GEMM_RT > 2 ? GEMM_RT : 2)];
    if ((tid & 3) == 0 && arow < BM) s_rowid[arow] = grow;
    const double coef_a = grow >= 0 ? (wmix ? 1.0 : g.ca[grow]) : 0.0;
    const double coef_r = (slab && grow >= 0) ? g.cr[grow] : 0.0;
    const double* __restrict__ Arow = g.A + (size_t)(grow >= 0 ? grow : 0) * D;
    // COH: the column's fields as buffers, this thread's row as a byte offset
    int colrow0 = 0;
    if constexpr (COH) colrow0 = row_of.base;
    const int colbytes = COH ? g.L * D * 8 : 0;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(g.A + (size_t)colrow0 * D, colbytes);
    const __amdgpu_buffer_rsrc_t rC = make_rsrc(g.C + (size_t)colrow0 * D, colbytes);
    const int avo = ((grow >= 0 ? grow : colrow0) - colrow0) * D * 8;
    const int RS = D + 2;                                       // ASTAGE: row stride of sRaw (16-byte aligned rows, 4 banks apart)
    if constexpr (ASTAGE) {
        typedef __attribute__((address_space(3))) void* lds_ptr_t_;
        __syncthreads();                                        // (s_rowid is complete; the previous tile has left sRaw)
        const int rowbytes = D * 8, pieces = (rowbytes + 1023) / 1024;
        for (int r = wave; r < BM; r += 4) {
            const int gr_ = s_rowid[r];
            if (gr_ < 0) continue;                              // (uniform) padding row: its coefficient is zero and its reads are guarded
            const int ro = (gr_ - colrow0) * rowbytes;
            for (int pc = 0; pc < pieces; ++pc)
                if (pc * 1024 + lane * 16 < rowbytes)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t_)(sRaw + (size_t)r * RS + pc * 128), 16, lane * 16, ro + pc * 1024, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#define SOSRT_A2(off_) (ASTAGE ? *reinterpret_cast<const double2*>(sRaw + (size_t)arow * RS + (off_)) \
                               : (COH ? bload2_aux<16>(rA, avo + (off_) * 8, 0) : *reinterpret_cast<const double2*>(Arow + (off_))))

    f64x4 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0, 0, 0, 0};

    constexpr int BQ = GEMM_KC * GEMM_BN / 256 / 2;      // double2 per thread of a W chunk
    constexpr int AQ = GEMM_KC / 8;                       // double2 per thread of an A chunk
    const int bk = tid / (GEMM_BN / (2 * BQ)), bc = (tid % (GEMM_BN / (2 * BQ))) * 2 * BQ;
    const int fr = lane & 15, fk = lane >> 4;
    const int nck = (SYM ? g.Ks : Dp) / GEMM_KC;  // chunks per pass
    const int ntot = slab ? 2 * nck : nck;
    // SYM: this thread's 8 doubles of a W chunk are 8 columns of S (first half of the staging row) or of A
    const int bcol = SYM ? (bc < GEMM_BN / 2 ? (bn0 >> 1) + bc : Nh + (bn0 >> 1) + bc - GEMM_BN / 2) : bn0 + bc;

    // Register staging as plain named values (arrays passed through lambdas end up in scratch).  The A
    // operand (In_1, from HBM) is staged two chunks ahead, the W operand (L2-resident) one chunk ahead.
    struct StageA { double2 a[AQ]; };
    struct StageM { double2 m[AQ]; };                   // the mirrored elements (In_1[D-1-k]; SYM only, else never touched)
    StageA s0;
    StageM m0;
    double2 sb0, sb1, sb2, sb3, sb4, sb5, sb6, sb7;     // named: an array here ends up in scratch
    // global -> registers for chunk c (clamped: every call issues the same loads)
#define SOSRT_GLOAD_A(ST, SM, c_)                                                                            \
    {                                                                                                     \
        const int cc_ = min((c_), ntot - 1);                                                              \
        const int kc_ = (cc_ >= nck ? cc_ - nck : cc_) * GEMM_KC;                                         \
        _Pragma("unroll") for (int q = 0; q < AQ; ++q) {                                                  \
            const int k0_ = kc_ + akq + 2 * q;                                                            \
            ST.a[q] = (grow >= 0 && k0_ + 1 < D) ? SOSRT_A2(k0_)          \
                                                 : make_double2(0, 0);                                    \
            if (SYM) SM.m[q] = (grow >= 0 && k0_ + 1 < D) ? SOSRT_A2(D - 2 - k0_) \
                                                          : make_double2(0, 0);                           \
        }                                                                                                 \
    }
#define SOSRT_GLOAD_B(c_)                                                                                 \
    {                                                                                                     \
        const int cc_ = min((c_), ntot - 1);                                                              \
        const int pass_ = cc_ >= nck ? 1 : 0;                                                             \
        const int kc_ = (cc_ - pass_ * nck) * GEMM_KC;                                                    \
        const double* __restrict__ W_ = wmix ? wmix : (pass_ ? g.Wr : g.Wa);                                              \
        const double* Wp_ = W_ + (size_t)(kc_ + bk) * Wld + bcol;                                         \
        sb0 = *reinterpret_cast<const double2*>(Wp_); sb1 = *reinterpret_cast<const double2*>(Wp_ + 2);    \
        sb2 = *reinterpret_cast<const double2*>(Wp_ + 4); sb3 = *reinterpret_cast<const double2*>(Wp_ + 6); \
        if (BQ > 4) {                                                                                     \
        sb4 = *reinterpret_cast<const double2*>(Wp_ + 8); sb5 = *reinterpret_cast<const double2*>(Wp_ + 10); \
        sb6 = *reinterpret_cast<const double2*>(Wp_ + 12); sb7 = *reinterpret_cast<const double2*>(Wp_ + 14); } \
    }
#define SOSRT_ASTORE(ST, SM, c_)                                                                          \
    {                                                                                                     \
        const double cf_ = (c_) >= nck ? coef_r : coef_a;                                                 \
        if (arow < BM) {                       /* fewer rows than staging threads in the small tiles */      \
        _Pragma("unroll") for (int q = 0; q < AQ; ++q) {                                                  \
            if (SYM) {                                                                                    \
                const int k0_ = ((c_) >= nck ? (c_) - nck : (c_)) * GEMM_KC + akq + 2 * q;                \
                const bool v0_ = k0_ < Nn, v1_ = k0_ + 1 < Nn;                                            \
                *reinterpret_cast<double2*>(&sA[arow * A_LD + akq + 2 * q]) =                             \
                    make_double2(v0_ ? cf_ * (ST.a[q].x + SM.m[q].y) : 0.0, v1_ ? cf_ * (ST.a[q].y + SM.m[q].x) : 0.0); \
                *reinterpret_cast<double2*>(&sAv[arow * A_LD + akq + 2 * q]) =                            \
                    make_double2(v0_ ? cf_ * (ST.a[q].x - SM.m[q].y) : 0.0, v1_ ? cf_ * (ST.a[q].y - SM.m[q].x) : 0.0); \
            } else {                                                                                      \
                *reinterpret_cast<double2*>(&sA[arow * A_LD + akq + 2 * q]) =                             \
                    make_double2(cf_ * ST.a[q].x, cf_ * ST.a[q].y);                                       \
            }                                                                                             \
        }                                                                                                 \
        }                                                                                                 \
    }
#define SOSRT_LSTORE(ST, SM, c_)                                                                          \
    {                                                                                                     \
        SOSRT_ASTORE(ST, SM, c_)                                                                            \
        double2* sbp_ = reinterpret_cast<double2*>(&sB[bk * B_LD + bc]);                                  \
        sbp_[0] = sb0; sbp_[1] = sb1; sbp_[2] = sb2; sbp_[3] = sb3;                                       \
        if (BQ > 4) { sbp_[4] = sb4; sbp_[5] = sb5; sbp_[6] = sb6; sbp_[7] = sb7; }                       \
    }
    auto compute = [&]() {
#pragma unroll
        for (int kk = 0; kk < GEMM_KC; kk += 4) {
            double af[RT], bf[2];
#pragma unroll
            for (int i = 0; i < RT; ++i) af[i] = sA[(i * 16 + fr) * A_LD + kk + fk];
            if (SYM) {
                double av[RT];
#pragma unroll
                for (int i = 0; i < RT; ++i) av[i] = sAv[(i * 16 + fr) * A_LD + kk + fk];
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = sB[(kk + fk) * B_LD + j * (GEMM_BN / 2) + wave * 16 + fr];
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bf[1], acc[i][1], 0, 0, 0);
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = sB[(kk + fk) * B_LD + wave * 32 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    if (!DEEP) {
        SOSRT_GLOAD_A(s0, m0, 0);
        SOSRT_GLOAD_B(0);
        for (int c = 0; c < ntot; ++c) {
            __syncthreads();                 // previous chunk consumed
            SOSRT_LSTORE(s0, m0, c);
            __syncthreads();
            SOSRT_GLOAD_A(s0, m0, c + 1);
            SOSRT_GLOAD_B(c + 1);
            compute();
        }
    } else {
        // Tail launches (few live columns: a workgroup is alone on its CU and nothing else hides the
        // load latency): both operands staged two chunks ahead in two register sets, loop unrolled by
        // two so that a set in flight is never copied.
        struct Stage { double2 a[AQ]; double2 b[BQ]; };
        Stage t0, t1;
        StageM m1;
#define SOSRT_GLOAD2(ST, SM, c_)                                                                          \
    {                                                                                                     \
        const int cc_ = min((c_), ntot - 1);                                                              \
        const int pass_ = cc_ >= nck ? 1 : 0;                                                             \
        const int kc_ = (cc_ - pass_ * nck) * GEMM_KC;                                                    \
        _Pragma("unroll") for (int q = 0; q < AQ; ++q) {                                                  \
            const int k0_ = kc_ + akq + 2 * q;                                                            \
            ST.a[q] = (grow >= 0 && k0_ + 1 < D) ? SOSRT_A2(k0_)          \
                                                 : make_double2(0, 0);                                    \
            if (SYM) SM.m[q] = (grow >= 0 && k0_ + 1 < D) ? SOSRT_A2(D - 2 - k0_) \
                                                          : make_double2(0, 0);                           \
        }                                                                                                 \
        const double* __restrict__ W_ = wmix ? wmix : (pass_ ? g.Wr : g.Wa);                                              \
        const double* Wp_ = W_ + (size_t)(kc_ + bk) * Wld + bcol;                                         \
        _Pragma("unroll") for (int q = 0; q < BQ; ++q) ST.b[q] = *reinterpret_cast<const double2*>(Wp_ + 2 * q); \
    }
#define SOSRT_LSTORE2(ST, SM, c_)                                                                          \
    {                                                                                                     \
        SOSRT_ASTORE(ST, SM, c_)                                                                            \
        double2* sbp_ = reinterpret_cast<double2*>(&sB[bk * B_LD + bc]);                                  \
        _Pragma("unroll") for (int q = 0; q < BQ; ++q) sbp_[q] = ST.b[q];                                 \
    }
        SOSRT_GLOAD2(t0, m0, 0);
        SOSRT_GLOAD2(t1, m1, 1);
        for (int c = 0; c < ntot; c += 2) {
            __syncthreads();
            SOSRT_LSTORE2(t0, m0, c);
            __syncthreads();
            SOSRT_GLOAD2(t0, m0, c + 2);
            compute();
            if (c + 1 < ntot) {
                __syncthreads();
                SOSRT_LSTORE2(t1, m1, c + 1);
                __syncthreads();
                SOSRT_GLOAD2(t1, m1, c + 3);
                compute();
            }
        }
#undef SOSRT_GLOAD2
#undef SOSRT_LSTORE2
    }
#undef SOSRT_GLOAD_A
#undef SOSRT_GLOAD_B
#undef SOSRT_LSTORE
#undef SOSRT_ASTORE
#undef SOSRT_A2
    // epilogue: lane holds column (l & 15), rows 4r + (l >> 4)
#pragma unroll
    for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = s_rowid[i * 16 + 4 * r + fk];       // (written before the first barrier of the k loop)
            if (SYM) {
                const int m = (bn0 >> 1) + wave * 16 + fr;
                if (gr >= 0 && m < Nn) {
                    const double x = acc[i][0][r], y = acc[i][1][r];
                    if constexpr (COH) {
                        const int ro = (gr - colrow0) * D * 8;
                        bstore_aux<17>(rC, ro + m * 8, 0, x + y);
                        bstore_aux<17>(rC, ro + (D - 1 - m) * 8, 0, x - y);
                    } else {
                        g.C[(size_t)gr * D + m] = x + y;
                        g.C[(size_t)gr * D + D - 1 - m] = x - y;
                    }
                }
            } else if (gr >= 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = bn0 + wave * 32 + j * 16 + fr;
                    if (col < D) {
                        if constexpr (COH) bstore_aux<17>(rC, ((gr - colrow0) * D + col) * 8, 0, acc[i][j][r]);
                        else g.C[(size_t)gr * D + col] = acc[i][j][r];
                    }
                }
            }
        }
    }
}


// The tile of the LAST FEW live columns (symmetric form): 16 rows x 64 values of m, no barrier inside the k-loop.
//
// A launch over a handful of columns is one tile's latency, and the staged tile above is eight k-chunks that wait for each other
// (two barriers, an LDS round trip, 16 MFMAs: ~1 us a chunk -- profiles/r04_order_loop_timeline.txt).  With a workgroup alone on
// its CU the register file holds what the chunks were staged for: a lane's fragments of the folded matrix for a block of 64 values
// of k are 2 x 16 doubles, requested straight from L2 into registers, two blocks in flight in two register sets; the tile's 16 rows
// of In_1 come into LDS whole by LDS-DMA (one barrier) and a lane forms u = c (a + b), v = c (a - b) on its way into the MFMA.
// Same products, same order of k, same epilogue as gemm_tile<.., SYM> -- a row's bits do not depend on the tiling that computed it.
// SLAB: two passes (W_atm with ca, then W_aer with cr) for slab rows without a combined matrix; wmix: one pass, unit coefficient.
constexpr int LONE_KB = 16;         // k-steps (of 4) per register block
struct LoneFrag { double s[LONE_KB], a[LONE_KB]; };      // a lane's fragments of [S | A] for one block of 64 values of k
// block `kb` (of the pass's Ks / 64) of the folded matrix W: k = 64 kb + 4 q + (lane >> 4), columns m (of S) and Nh + m (of A).
// Buffer loads: the lane's part of the address is two registers for all 32 requests (the k-step goes into the scalar offset);
// with 64-bit addresses the 8-KB stride between k-steps costs a register pair per request and the fragments spill.
__device__ __forceinline__ void lone_load(LoneFrag& F, __amdgpu_buffer_rsrc_t rW, int kb, int fk, int Wld, int Nh, int mcol) {
    const int vs = (fk * Wld + mcol) * 8, va = vs + Nh * 8;
#pragma unroll
    for (int q = 0; q < LONE_KB; ++q) {
        const int so = 4 * (kb * LONE_KB + q) * Wld * 8;
        F.s[q] = bload(rW, vs, so);
        F.a[q] = bload(rW, va, so);
    }
}
// PRELOADED: the caller requested blocks 0 and 1 of the (single) pass into f0 / f1 before it knew the tile's column -- the plain
// rows' matrix is the same for every column, so its fragments travel while the live flags do.
template <bool SLAB, bool PRELOADED>
__device__ __forceinline__ void gemm_tile_lone(const GemmArgs& g, double* sRaw, int tile, int bn0, ColumnRows row_of,
                                               LoneFrag& f0, LoneFrag& f1, const double* __restrict__ wmix = nullptr) {
    static_assert(!(SLAB && PRELOADED), "two passes start from the column's own coefficients");
    typedef __attribute__((address_space(3))) void* lds_ptr_t_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int D = g.D, Wld = g.Wld, Nn = D >> 1, Nh = Wld >> 1;
    const int RS = D + 2;                                // row stride of sRaw: 16-byte aligned rows, 4 banks apart (D = 0 mod 32)
    const int fr = lane & 15, fk = lane >> 4;
    const int nkb = (g.Ks >> 2) / LONE_KB;               // register blocks per pass (the host takes this tile when Ks is a multiple of 64)
    const int nblk = (SLAB ? 2 : 1) * nkb;
    const int grow = row_of(tile * 16 + fr);             // the row whose A fragment this lane supplies
    const double cf_a = grow >= 0 ? (wmix ? 1.0 : g.ca[grow]) : 0.0;
    const double cf_r = (SLAB && grow >= 0) ? g.cr[grow] : 0.0;
    // the tile's rows of In_1, whole, by LDS-DMA (a wave takes every fourth row)
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(g.A + (size_t)row_of.base * D, g.L * D * 8);
    const int rowbytes = D * 8, pieces = (rowbytes + 1023) / 1024;
    for (int r = wave; r < 16; r += 4) {
        const int gr_ = row_of(tile * 16 + r);
        if (gr_ < 0) continue;                           // (uniform) padding row: never read below
        const int ro = (gr_ - row_of.base) * rowbytes;
        for (int pc = 0; pc < pieces; ++pc)
            if (pc * 1024 + lane * 16 < rowbytes)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t_)(sRaw + (size_t)r * RS + pc * 128), 16, lane * 16, ro + pc * 1024, 0, 0);
    }
    // this lane's fragments of the folded matrix: X against S (first half of a row of [S | A]), Y against A
    const int mcol = (bn0 >> 1) + wave * 16 + fr;
#define SOSRT_LONE_LOAD(F, blk_)                                                                                  \
    {                                                                                                             \
        const int b_ = (blk_);                                                                                    \
        if (b_ < nblk) {                                                                                          \
            const int pass_ = b_ >= nkb ? 1 : 0;                                                                  \
            lone_load(F, pass_ ? rW1 : rW0, b_ - pass_ * nkb, fk, Wld, Nh, mcol);                                 \
        }                                                                                                         \
    }
    const __amdgpu_buffer_rsrc_t rW0 = make_rsrc(wmix ? wmix : g.Wa, g.Dp * Wld * 8);
    const __amdgpu_buffer_rsrc_t rW1 = make_rsrc(SLAB ? g.Wr : g.Wa, g.Dp * Wld * 8);
    if (!PRELOADED) {
        SOSRT_LONE_LOAD(f0, 0);
        SOSRT_LONE_LOAD(f1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f64x4 accx = (f64x4){0, 0, 0, 0}, accy = (f64x4){0, 0, 0, 0};
    const double* __restrict__ arow = sRaw + (size_t)fr * RS;
#define SOSRT_LONE_COMPUTE(F, blk_)                                                                               \
    {                                                                                                             \
        const int b_ = (blk_);                                                                                    \
        const int pass_ = b_ >= nkb ? 1 : 0;                                                                      \
        const double cf_ = pass_ ? cf_r : cf_a;                                                                   \
        const int k0_ = 4 * (b_ - pass_ * nkb) * LONE_KB + fk;                                                    \
        _Pragma("unroll") for (int q = 0; q < LONE_KB; ++q) {                                                     \
            const int k_ = k0_ + 4 * q;                                                                           \
            const bool ok_ = grow >= 0 && k_ < Nn;                                                                \
            const double a_ = arow[k_], m_ = arow[D - 1 - k_];          /* (in bounds whatever k_ < Ks) */        \
            const double u_ = ok_ ? cf_ * (a_ + m_) : 0.0, v_ = ok_ ? cf_ * (a_ - m_) : 0.0;                      \
            accx = __builtin_amdgcn_mfma_f64_16x16x4f64(u_, F.s[q], accx, 0, 0, 0);                               \
            accy = __builtin_amdgcn_mfma_f64_16x16x4f64(v_, F.a[q], accy, 0, 0, 0);                               \
        }                                                                                                         \
    }
    for (int blk = 0; blk < nblk; blk += 2) {
        SOSRT_LONE_COMPUTE(f0, blk);
        SOSRT_LONE_LOAD(f0, blk + 2);
        if (blk + 1 < nblk) {
            SOSRT_LONE_COMPUTE(f1, blk + 1);
            SOSRT_LONE_LOAD(f1, blk + 3);
        }
    }
#undef SOSRT_LONE_LOAD
#undef SOSRT_LONE_COMPUTE
    // epilogue: lane holds m = mcol, rows 4r + (l >> 4)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gr = row_of(tile * 16 + 4 * r + fk);
        if (gr >= 0 && mcol < Nn) {
            const double x = accx[r], y = accy[r];
            g.C[(size_t)gr * D + mcol] = x + y;
            g.C[(size_t)gr * D + D - 1 - mcol] = x - y;
        }
    }
}

constexpr int TAIL_RT_SLAB = 1;     // MFMA row tiles of the live-column tilings: slab rows (16-row tiles)
constexpr int TAIL_RT = 2;          // ... plain rows of the small tiling (32-row tiles)

}  // namespace
}  // namespace sosrt
