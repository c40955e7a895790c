// k_jn_gemm: the source function of one scattering order for every layer of every column,
//     Jn[r][:] = ca[r] (In_1[r][:] @ W_atm)  (+ cr[r] (In_1[r][:] @ W_aer) for aerosol-slab rows)
// (spec:314-323, I1_In:62-74) as an FP64-MFMA contraction.
//
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; the four
// results of lane l are D[i = 4r + (l>>4)][j = l&15], r = 0..3.
//
// Workgroup tile: 64 rows x 128 columns, 4 waves, each wave 4x2 MFMA tiles (64 accumulator
// registers).  Operands are staged through LDS in k-chunks of GEMM_KC = 16 with row strides chosen so
// that the fragment reads (ds_read_b64) are bank-conflict free: A rows 18 doubles, W rows 144 doubles
// (288 = 32 mod 64 dwords).  The global loads of the next chunk are in registers while the current one
// is multiplied (one chunk ahead in the dense tilings, four workgroups per CU hide the rest; two chunks
// ahead in two register sets, loop unrolled by two, in the tail tiling, where a workgroup is alone on
// its CU).
//
// Rows are addressed through row lists, so that one launch covers the plain rows (against W_atm) and the
// slab rows (against the combined matrix ca W_atm + cr W_aer of their coefficient pair, one pass; two
// passes, W_atm then W_aer, only when a batch has more distinct pairs than the cache of combined matrices)
// without either writing the other's rows; the per-row coefficient is applied to the A operand on its way
// into LDS.
//
// Two forms of the product (chosen per handle from the matrices, never from the batch): the full D x D
// product, and -- when the folded matrices are flip-symmetric, as every phase function of the scattering
// angle makes them -- two N x N products on the sum and the difference of a row's two halves (SYM below):
// half the flops.
#include "kernels.hpp"

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
template <int RT, bool SLAB, bool DEEP = false, bool SYM = false, class RowOf = ListRows>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, double* sA, double* sB, int* s_any, int tile, int bn0,
                                          RowOf row_of, bool check_active, const double* __restrict__ wmix = nullptr) {
    constexpr int BM = 16 * RT;
    double* const sAv = sA + BM * A_LD;          // SYM: the v operand next to the u operand
    const int Nn = g.D >> 1, Nh = g.Wld >> 1;
    const int tid = threadIdx.x, lane = tid & 63;
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
            ST.a[q] = (grow >= 0 && k0_ + 1 < D) ? *reinterpret_cast<const double2*>(Arow + k0_)          \
                                                 : make_double2(0, 0);                                    \
            if (SYM) SM.m[q] = (grow >= 0 && k0_ + 1 < D) ? *reinterpret_cast<const double2*>(Arow + D - 2 - k0_) \
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
            ST.a[q] = (grow >= 0 && k0_ + 1 < D) ? *reinterpret_cast<const double2*>(Arow + k0_)          \
                                                 : make_double2(0, 0);                                    \
            if (SYM) SM.m[q] = (grow >= 0 && k0_ + 1 < D) ? *reinterpret_cast<const double2*>(Arow + D - 2 - k0_) \
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
                    g.C[(size_t)gr * D + m] = x + y;
                    g.C[(size_t)gr * D + D - 1 - m] = x - y;
                }
            } else if (gr >= 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = bn0 + wave * 32 + j * 16 + fr;
                    if (col < D) g.C[(size_t)gr * D + col] = acc[i][j][r];
                }
            }
        }
    }
}

template <bool SYM>
__global__ __launch_bounds__(256, GEMM_WPS) void k_jn_gemm(GemmArgs g) {
    publish_live(g);
    __shared__ double sA[(SYM ? 2 : 1) * 16 * (GEMM_RT > 2 ? GEMM_RT : 2) * A_LD];
    __shared__ double sB[GEMM_KC * B_LD];
    __shared__ int s_any;
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  The column tiles of one row
    // tile read the same rows of In_1: numbering them 8 apart puts them on one XCD, a few dispatches apart,
    // so that In_1 comes from HBM once instead of once per column tile.
    const int tiles_main = (g.n_main + 16 * GEMM_RT - 1) / (16 * GEMM_RT);
    const int tiles = tiles_main + (g.n_slab + GEMM_BM / 2 - 1) / (GEMM_BM / 2);
    const int nct = (g.D + GEMM_BN - 1) / GEMM_BN;
    const int id = blockIdx.x;
    const int tile = (id / (8 * nct)) * 8 + (id & 7), bn0 = ((id >> 3) % nct) * GEMM_BN;
    if (tile >= tiles) return;
    if (tile < tiles_main) {
        gemm_tile<GEMM_RT, false, false, SYM>(g, sA, sB, &s_any, tile, bn0, ListRows{g.rows_main, g.n_main}, true);
    } else if (g.Wmix && g.slab_tile_group) {
        // the slab rows are listed group by group (one distinct coefficient pair per 32-row tile): one pass over the
        // group's combined matrix, the same arithmetic as the live-column tilings (a column's result does not
        // depend on which tiling, or which batch, it was computed in)
        const int st = tile - tiles_main;
        gemm_tile<2, false, false, SYM>(g, sA, sB, &s_any, st, bn0, ListRows{g.rows_slab, g.n_slab}, true,
                                        g.Wmix + (size_t)g.slab_tile_group[st] * g.Dp * g.Wld);
    } else {
        gemm_tile<2, true, false, SYM>(g, sA, sB, &s_any, tile - tiles_main, bn0, ListRows{g.rows_slab, g.n_slab}, true);
    }
}

// The same contraction once some columns have converged.  Tiling the row lists would launch a
// workgroup for every tile of every column, and the few live tiles would queue behind thousands of
// workgroups that only find out that their columns have converged.  Here blockIdx.x = (i, tile): the
// workgroup finds the i-th live column itself (a prefix count over the live flags) and tiles that
// column's rows: 16-row tiles for the slab rows, scheduled first because their double pass over k is
// the critical path of the launch, then the tiles of the plain rows.
constexpr int TAIL_RT_SLAB = 1;
template <int RT, bool DEEP, bool SYM>
__device__ __forceinline__ void gemm_live_columns(const GemmArgs& g, double* sA, double* sB) {
    __shared__ int s_w[4];
    __shared__ int s_col;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ts = (g.max_slab + 16 * TAIL_RT_SLAB - 1) / (16 * TAIL_RT_SLAB);
    const int tm = (g.max_main + 16 * RT - 1) / (16 * RT);
    // same XCD-aware numbering as the dense kernel: the column tiles of a row tile 8 apart
    const int nct = (g.D + GEMM_BN - 1) / GEMM_BN;
    const int id = blockIdx.x;
    const int xq = (id / (8 * nct)) * 8 + (id & 7), bn0 = ((id >> 3) % nct) * GEMM_BN;
    const int ci = xq / (ts + tm), tt = xq % (ts + tm);
    if (tid == 0) s_col = -1;
    int before = 0;
    for (int base = 0; base < g.B; base += 256) {
        const bool f = base + tid < g.B && g.active[g.col0 + base + tid] != 0;
        const unsigned long long mk = __ballot(f);
        if (lane == 0) s_w[wave] = __popcll(mk);
        __syncthreads();
        int pre = before, tot = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) pre += s_w[w];
            tot += s_w[w];
        }
        if (f && pre + __popcll(mk & ((2ull << lane) - 1)) == ci + 1) s_col = g.col0 + base + tid;
        before += tot;
        __syncthreads();
        if (before > ci) break;
    }
    const int b = s_col;
    // the transport of this order takes its columns from this list
    if (g.live_list && tt == 0 && bn0 == 0 && tid == 0 && ci < g.live_cap) g.live_list[ci] = b < 0 ? -1 : b - g.col0;
    if (b < 0) return;                                   // fewer live columns than the host's (lagging) count
    const int iu = g.idx_up ? g.idx_up[b] : 0;
    const int ns = g.idx_up ? g.idx_down[b] - iu + 1 : 0;
    if (tt < ts) {
        if (tt * 16 * TAIL_RT_SLAB >= ns) return;
        if (g.Wmix)      // SLAB = false: a single pass, over the column's combined matrix
            gemm_tile<TAIL_RT_SLAB, false, DEEP, SYM>(g, sA, sB, nullptr, tt, bn0, ColumnRows{b * g.L, iu, ns, ns, true},
                                                      false, g.Wmix + (size_t)g.mix_group[b] * g.Dp * g.Wld);
        else
            gemm_tile<TAIL_RT_SLAB, true, DEEP, SYM>(g, sA, sB, nullptr, tt, bn0, ColumnRows{b * g.L, iu, ns, ns, true}, false);
    } else {
        const int t2 = tt - ts;
        if (t2 * 16 * RT >= g.L - ns) return;
        gemm_tile<RT, false, DEEP, SYM>(g, sA, sB, nullptr, t2, bn0, ColumnRows{b * g.L, iu, ns, g.L - ns, false}, false);
    }
}

// many live columns: the tile of the dense kernel
template <bool SYM>
__global__ __launch_bounds__(256, SYM ? 4 : GEMM_WPS) void k_jn_gemm_cols(GemmArgs g) {
    publish_live(g);
    __shared__ double sA[(SYM ? 2 : 1) * 16 * GEMM_RT * A_LD];
    __shared__ double sB[GEMM_KC * B_LD];
    gemm_live_columns<GEMM_RT, false, SYM>(g, sA, sB);
}
// few live columns: 32-row tiles (more workgroups, so more CUs take part) and deeper staging, since
// such a workgroup is alone on its CU
constexpr int TAIL_RT = 2;
// (symmetric form: staged one chunk ahead like the dense tiling -- 94 registers, five workgroups per CU instead of three;
// with 55 to 200 live columns the launch is a few rounds of workgroups and the fuller CUs save a round: the tail of the
// contraction 0.63 -> 0.57 ms per step.  The full product keeps the two-chunk staging it was tuned with.)
constexpr bool kTailDeepSym = false;
constexpr int kTailWpsSym = 4;
template <bool SYM>
__global__ __launch_bounds__(256, SYM ? kTailWpsSym : 2) void k_jn_gemm_tail(GemmArgs g) {
    publish_live(g);
    __shared__ double sA[(SYM ? 2 : 1) * 16 * TAIL_RT * A_LD];
    __shared__ double sB[GEMM_KC * B_LD];
    gemm_live_columns<TAIL_RT, SYM ? kTailDeepSym : true, SYM>(g, sA, sB);
}
// the last few columns (at most kTailDeepCols live): a workgroup is alone on its CU and every k-chunk is a trip to L2 or HBM that
// nothing else hides, so both operands are staged two chunks ahead (the register budget no longer matters: one workgroup per CU)
constexpr int kTailDeepCols = 32;
__global__ __launch_bounds__(256, 2) void k_jn_gemm_tail_deep(GemmArgs g) {
    publish_live(g);
    __shared__ double sA[2 * 16 * TAIL_RT * A_LD];
    __shared__ double sB[GEMM_KC * B_LD];
    gemm_live_columns<TAIL_RT, true, true>(g, sA, sB);
}

}  // namespace

void launch_gemm_tail(hipStream_t s, const GemmArgs& a, int cols, bool small_tiles) {
    const int rt = small_tiles ? TAIL_RT : GEMM_RT;
    const int ts = (a.max_slab + 16 * TAIL_RT_SLAB - 1) / (16 * TAIL_RT_SLAB);
    const int tm = (a.max_main + 16 * rt - 1) / (16 * rt);
    if (cols <= 0 || ts + tm <= 0) return;
    const int nct = (a.D + GEMM_BN - 1) / GEMM_BN;
    dim3 grid((unsigned)((cols * (ts + tm) + 7) / 8 * 8 * nct));
    if (small_tiles) {
        if (a.sym && cols <= kTailDeepCols) hipLaunchKernelGGL(k_jn_gemm_tail_deep, grid, dim3(256), 0, s, a);
        else if (a.sym) hipLaunchKernelGGL(k_jn_gemm_tail<true>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_jn_gemm_tail<false>, grid, dim3(256), 0, s, a);
    } else {
        if (a.sym) hipLaunchKernelGGL(k_jn_gemm_cols<true>, grid, dim3(256), (size_t)a.pad_lds, s, a);
        else hipLaunchKernelGGL(k_jn_gemm_cols<false>, grid, dim3(256), (size_t)a.pad_lds, s, a);
    }
}

// Wmix[g] = ca[g] W_atm + cr[g] W_aer for every distinct slab coefficient pair of the batch
__global__ void k_wmix(size_t n, int ngroups, const double* __restrict__ Wa, const double* __restrict__ Wr,
                       const double* __restrict__ ca, const double* __restrict__ cr, double* __restrict__ Wmix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = Wa[i], r = Wr[i];
    for (int gq = 0; gq < ngroups; ++gq) Wmix[(size_t)gq * n + i] = ca[gq] * a + cr[gq] * r;
}
void launch_wmix(hipStream_t s, size_t n, int ngroups, const double* Wa, const double* Wr, const double* ca, const double* cr,
                 double* Wmix) {
    hipLaunchKernelGGL(k_wmix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, ngroups, Wa, Wr, ca, cr, Wmix);
}

void launch_gemm(hipStream_t s, const GemmArgs& a) {
    const int tiles = (a.n_main + 16 * GEMM_RT - 1) / (16 * GEMM_RT) + (a.n_slab + GEMM_BM / 2 - 1) / (GEMM_BM / 2);
    if (tiles <= 0) return;
    const int nct = (a.D + GEMM_BN - 1) / GEMM_BN;
    dim3 grid((unsigned)((tiles + 7) / 8 * 8 * nct));
    if (a.sym) hipLaunchKernelGGL(k_jn_gemm<true>, grid, dim3(256), (size_t)a.pad_lds, s, a);
    else hipLaunchKernelGGL(k_jn_gemm<false>, grid, dim3(256), (size_t)a.pad_lds, s, a);
}

// Flip-symmetric folding of a contraction matrix (gemm_tile, SYM): for k, m < N, with k' = D-1-k, m' = D-1-m,
//     S[k][m] = (W[k][m] + W[k'][m'] + W[k'][m] + W[k][m']) / 4,   A[k][m] = (W[k][m] + W[k'][m'] - W[k'][m] - W[k][m']) / 4
// (the symmetric part of W; what is dropped is W[k][m] - W[k'][m'], the rounding of the phase-matrix builders, which the
// host has checked to be negligible).  Stored [k][S: 0..Wld/2 | A: Wld/2..Wld], zero elsewhere.
__global__ void k_symfold(int nmat, int N, int D, int Dp, int Wld, const double* __restrict__ W, double* __restrict__ SA) {
    const size_t per = (size_t)Dp * Wld;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * nmat) return;
    const size_t mat = i / per, r = i % per;
    const int k = (int)(r / Wld), c = (int)(r % Wld), Nh = Wld >> 1;
    const int m = c < Nh ? c : c - Nh;
    double out = 0.0;
    if (k < N && m < N) {
        const double* w = W + mat * per;
        const double p = w[(size_t)k * Wld + m], pf = w[(size_t)(D - 1 - k) * Wld + (D - 1 - m)];
        const double q = w[(size_t)(D - 1 - k) * Wld + m], qf = w[(size_t)k * Wld + (D - 1 - m)];
        out = c < Nh ? 0.25 * ((p + pf) + (q + qf)) : 0.25 * ((p + pf) - (q + qf));
    }
    SA[i] = out;
}
void launch_symfold(hipStream_t s, int nmat, int N, int D, int Dp, int Wld, const double* W, double* SA) {
    const size_t n = (size_t)Dp * Wld * nmat;
    hipLaunchKernelGGL(k_symfold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nmat, N, D, Dp, Wld, W, SA);
}

}  // namespace sosrt
