// k_jn_gemm: the source function of one scattering order for every layer of every column,
//     Jn[r][:] = ca[r] (In_1[r][:] @ W_atm)  (+ cr[r] (In_1[r][:] @ W_aer) for aerosol-slab rows)
// (spec:314-323, I1_In:62-74) as an FP64-MFMA contraction.
//
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; the four
// results of lane l are D[i = 4r + (l>>4)][j = l&15], r = 0..3.
//
// Workgroup tile: 64 rows x 128 columns, 4 waves, each wave 4x2 MFMA tiles (64 accumulator
// registers).  Operands are staged through LDS in k-chunks of 32 with row strides chosen so that
// the fragment reads (ds_read_b64) are bank-conflict free: A rows 34 doubles (68 dwords, 68/4 odd),
// W rows 144 doubles (288 = 32 mod 64 dwords).  Two register buffers hold the global loads of the
// next two chunks while the current one is multiplied (the loop is unrolled by two, so a buffer in
// flight is never copied); a chunk is 64 MFMAs per wave = 4096 cycles of matrix pipe, which is
// what hides the L2 / HBM latency of the loads, also when a workgroup is alone on its CU.
//
// Rows are addressed through row lists, so that one launch covers the plain rows (one pass over
// W_atm) and the slab rows (two passes: W_atm then W_aer) without either writing the other's
// rows; the per-row coefficient is applied to the A operand on its way into LDS.
#include "kernels.hpp"

namespace sosrt {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int A_LD = GEMM_KC + 2;
constexpr int B_LD = GEMM_BN + 16;

__global__ __launch_bounds__(256, 2) void k_jn_gemm(GemmArgs g) {
    __shared__ double sA[GEMM_BM * A_LD];
    __shared__ double sB[GEMM_KC * B_LD];
    __shared__ int s_any;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int D = g.D, Dp = g.Dp, Wld = g.Wld;
    const int tiles_main = (g.n_main + GEMM_BM - 1) / GEMM_BM;
    const bool slab = (int)blockIdx.x >= tiles_main;                 // uniform
    const int* __restrict__ rows = slab ? g.rows_slab : g.rows_main;
    const int nrows = slab ? g.n_slab : g.n_main;
    const int bm0 = (slab ? (int)blockIdx.x - tiles_main : (int)blockIdx.x) * GEMM_BM, bn0 = blockIdx.y * GEMM_BN;

    // the row this thread stages, its coefficients; skip the tile when every column it touches has converged
    const int arow = tid >> 2, akq = (tid & 3) * 8;
    int grow = -1;
    if (bm0 + arow < nrows) grow = rows ? rows[bm0 + arow] : bm0 + arow;
    if (g.active) {
        if (tid == 0) s_any = 0;
        __syncthreads();
        if ((tid & 3) == 0 && grow >= 0 && g.active[grow / g.L]) s_any = 1;
        __syncthreads();
        if (!s_any) return;
    }
    const double coef_a = grow >= 0 ? g.ca[grow] : 0.0;
    const double coef_r = (slab && grow >= 0) ? g.cr[grow] : 0.0;
    const double* __restrict__ Arow = g.A + (size_t)(grow >= 0 ? grow : 0) * D;

    f64x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0, 0, 0, 0};

    const int bk = tid >> 3, bc = (tid & 7) * 16;
    const int fr = lane & 15, fk = lane >> 4;
    const int nck = Dp / GEMM_KC;                 // chunks per pass
    const int ntot = slab ? 2 * nck : nck;

    // Register staging buffers as plain named values (arrays passed through lambdas end up in scratch).
    struct Stage { double2 a[4]; double2 b[8]; };
    Stage s0, s1;
    // global -> registers for chunk c (clamped: every call issues the same loads)
#define SOSRT_GLOAD(ST, c_)                                                                               \
    {                                                                                                     \
        const int cc_ = min((c_), ntot - 1);                                                              \
        const int pass_ = cc_ >= nck ? 1 : 0;                                                             \
        const int kc_ = (cc_ - pass_ * nck) * GEMM_KC;                                                    \
        const double* __restrict__ W_ = pass_ ? g.Wr : g.Wa;                                              \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
            const int k0_ = kc_ + akq + 2 * q;                                                            \
            ST.a[q] = (grow >= 0 && k0_ + 1 < D) ? *reinterpret_cast<const double2*>(Arow + k0_)          \
                                                 : make_double2(0, 0);                                    \
        }                                                                                                 \
        const double* Wp_ = W_ + (size_t)(kc_ + bk) * Wld + bn0 + bc;                                     \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) ST.b[q] = *reinterpret_cast<const double2*>(Wp_ + 2 * q); \
    }
#define SOSRT_LSTORE(ST, c_)                                                                              \
    {                                                                                                     \
        const double cf_ = (c_) >= nck ? coef_r : coef_a;                                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                     \
            *reinterpret_cast<double2*>(&sA[arow * A_LD + akq + 2 * q]) =                                 \
                make_double2(cf_ * ST.a[q].x, cf_ * ST.a[q].y);                                           \
        _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                     \
            *reinterpret_cast<double2*>(&sB[bk * B_LD + bc + 2 * q]) = ST.b[q];                           \
    }
    auto compute = [&]() {
#pragma unroll
        for (int kk = 0; kk < GEMM_KC; kk += 4) {
            double af[4], bf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = sA[(i * 16 + fr) * A_LD + kk + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = sB[(kk + fk) * B_LD + wave * 32 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    SOSRT_GLOAD(s0, 0);
    SOSRT_GLOAD(s1, 1);
    for (int c = 0; c < ntot; c += 2) {
        __syncthreads();                 // previous chunk consumed
        SOSRT_LSTORE(s0, c);
        __syncthreads();
        SOSRT_GLOAD(s0, c + 2);
        compute();
        if (c + 1 >= ntot) break;
        __syncthreads();
        SOSRT_LSTORE(s1, c + 1);
        __syncthreads();
        SOSRT_GLOAD(s1, c + 3);
        compute();
    }
#undef SOSRT_GLOAD
#undef SOSRT_LSTORE
    // epilogue: lane holds column (l & 15), rows 4r + (l >> 4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lr = bm0 + i * 16 + 4 * r + fk;
            if (lr < nrows) {
                const int gr = rows ? rows[lr] : lr;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = bn0 + wave * 32 + j * 16 + fr;
                    if (col < D) g.C[(size_t)gr * D + col] = acc[i][j][r];
                }
            }
        }
    }
}

}  // namespace

void launch_gemm(hipStream_t s, const GemmArgs& a) {
    const int tiles = (a.n_main + GEMM_BM - 1) / GEMM_BM + (a.n_slab + GEMM_BM - 1) / GEMM_BM;
    if (tiles <= 0) return;
    dim3 grid(tiles, (a.D + GEMM_BN - 1) / GEMM_BN);
    hipLaunchKernelGGL(k_jn_gemm, grid, dim3(256), 0, s, a);
}

}  // namespace sosrt
