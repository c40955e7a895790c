// k_jn_gemm: the source function of one scattering order for every layer of every column,
//     Jn[r][:] = ca[r] (In_1[r][:] @ W_atm)  (+ cr[r] (In_1[r][:] @ W_aer) for aerosol-slab rows)
// (spec:314-323, I1_In:62-74) as an FP64-MFMA contraction.
//
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; the four
// results of lane l are D[i = 4r + (l>>4)][j = l&15], r = 0..3.
//
// Workgroup tile: 64 rows x 128 columns, 4 waves, each wave 4x2 MFMA tiles (64 accumulator
// registers).  Operands are staged through LDS in k-chunks of GEMM_KC = 16 with row strides chosen for
// the fragment READS (ds_read_b64: A rows 18 doubles, W rows 144 doubles = 32 mod 64 dwords, so the 32
// lanes of a half wave fall into 64 distinct banks).  The kernel is not free of bank conflicts, though:
// the staging STORES are 16 bytes per lane, four lanes per 18-double row, and SQ_LDS_BANK_CONFLICT counts
// 8.9e6 cycles per launch of the symmetric form, 18 % of the launch's CU-cycles, against an MFMA pipe
// busy for 47 % of the SIMD-cycles (profiles/r04_pmc_gemm_sq.txt; round 2's "conflict free" was the reads).
// The global loads of the next chunk are in registers while the current one
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
#include "jn_gemm_tile.hpp"

namespace sosrt {

namespace {

// MFMA row tiles of a slab-row workgroup of the dense tiling: the slab rows are the last workgroups of a launch, and the last
// (partial) round of workgroups costs a whole tile's time whatever its fill -- with 16-row slab tiles (1) instead of 32-row ones (2)
// that round is twice as full and half as long.  (A row's arithmetic does not depend on its tile's height: the live-column tilings
// have always used 16-row slab tiles.)
#ifndef SOSRT_DENSE_SLAB_RT
#define SOSRT_DENSE_SLAB_RT 1
#endif
constexpr int DENSE_SLAB_RT = SOSRT_DENSE_SLAB_RT;
#ifndef SOSRT_LIVE_TILE_MAJOR
#define SOSRT_LIVE_TILE_MAJOR 1
#endif
constexpr int DENSE_SLAB_ROWS = 16 * DENSE_SLAB_RT;

// The live columns of the launch's group, in ascending order, for the transport of this order (live_list[i] = i-th live column
// relative to col0, -1 from the live count up to live_cap): one workgroup.  The live-column tilings write the list as they find
// their columns; the dense tiling has no such step, and a transport launched over ALL columns of a half-converged batch leaves
// its live columns where the batch put them -- two to a CU here, none there: 124 us for 271 live columns of 512 where the same
// kernel over the list takes 75 (profiles/r04_order_table.txt, orders 11-13).
__device__ __forceinline__ void write_live_list(const GemmArgs& g) {
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int before = 0;
    for (int base = 0; base < g.B; base += 256) {
        const bool f = base + tid < g.B && g.active[g.col0 + base + tid] != 0;
        const unsigned long long mk = __ballot(f);
        if (lane == 0) s_cnt[wave] = __popcll(mk);
        __syncthreads();
        int pre = before, tot = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) pre += s_cnt[w];
            tot += s_cnt[w];
        }
        const int at = pre + __popcll(mk & ((1ull << lane) - 1));
        if (f && at < g.live_cap) g.live_list[at] = base + tid;
        before += tot;
        __syncthreads();
    }
    for (int i = before + tid; i < g.live_cap; i += 256) g.live_list[i] = -1;
}

template <bool SYM>
__global__ __launch_bounds__(256, GEMM_WPS) void k_jn_gemm(GemmArgs g) {
    if (g.live_list && blockIdx.x == gridDim.x - 1) {    // (the launch's extra workgroup: launch_gemm)
        write_live_list(g);
        return;
    }
    publish_live(g);
    __shared__ double sA[(SYM ? 2 : 1) * 16 * (GEMM_RT > 2 ? GEMM_RT : 2) * A_LD];
    __shared__ double sB[GEMM_KC * B_LD];
    __shared__ int s_any;
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  The column tiles of one row
    // tile read the same rows of In_1: numbering them 8 apart puts them on one XCD, a few dispatches apart,
    // so that In_1 comes from HBM once instead of once per column tile.
    const int tiles_main = (g.n_main + 16 * GEMM_RT - 1) / (16 * GEMM_RT);
    const int tiles = tiles_main + (g.n_slab + DENSE_SLAB_ROWS - 1) / DENSE_SLAB_ROWS;
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
        // (the host lists a group for every 32 slab rows)
        gemm_tile<DENSE_SLAB_RT, false, false, SYM>(g, sA, sB, &s_any, st, bn0, ListRows{g.rows_slab, g.n_slab}, true,
                                                    g.Wmix + (size_t)g.slab_tile_group[st * DENSE_SLAB_ROWS / 32] * g.Dp * g.Wld);
    } else {
        gemm_tile<DENSE_SLAB_RT, true, false, SYM>(g, sA, sB, &s_any, tile - tiles_main, bn0, ListRows{g.rows_slab, g.n_slab}, true);
    }
}

// The same contraction once some columns have converged.  Tiling the row lists would launch a
// workgroup for every tile of every column, and the few live tiles would queue behind thousands of
// workgroups that only find out that their columns have converged.  Here blockIdx.x = (i, tile): the
// workgroup finds the i-th live column itself (a prefix count over the live flags) and tiles that
// column's rows: 16-row tiles for the slab rows, scheduled first because their double pass over k is
// the critical path of the launch, then the tiles of the plain rows.
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
#if SOSRT_LIVE_TILE_MAJOR
    // Tile-major, the plain rows' tiles of every column first, the slab rows' 16-row tiles last: the last workgroups of a launch
    // decide how long its last, partly filled round of workgroups takes, and with the combined slab matrices (one pass over k) the slab
    // tiles are the short ones.  (Until round 3 a column's tiles were consecutive, slab tiles first -- from the time they made two
    // passes over k.)
    const int cap = g.live_cap > 0 ? g.live_cap : 1;
    const int tq = xq / cap, ci = xq % cap;
    if (tq >= ts + tm) return;                           // (uniform) padding of the grid
    const int tt = tq < tm ? ts + tq : tq - tm;          // tt < ts: slab tile tt; else plain tile tt - ts
#else
    const int ci = xq / (ts + tm), tt = xq % (ts + tm);
#endif
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
    if (g.live_list && xq < (SOSRT_LIVE_TILE_MAJOR ? (g.live_cap > 0 ? g.live_cap : 1) : 1 << 30) &&
        (SOSRT_LIVE_TILE_MAJOR || tt == 0) && bn0 == 0 && tid == 0 && ci < g.live_cap)
        g.live_list[ci] = b < 0 ? -1 : b - g.col0;
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

// the last handful of columns (api.hip: plan_order, SOSRT_PLAN_GEMM_LIVE16_REGS): one tile's latency is the launch's, so the tile
// keeps its share of the folded matrix in registers and has no barrier in its k-loop (jn_gemm_tile.hpp: gemm_tile_lone), and the
// way to the tile is three memory round trips -- live flags; the column's slab rows and matrix group, requested together; operands
// -- with the report to the host on a workgroup of its own (the last one: its release at system scope and the load in front of it
// would otherwise be in front of the first tile).  Tile-major numbering as in gemm_live_columns, 16-row tiles for all rows.
__global__ __launch_bounds__(256, 1) void k_jn_gemm_lone(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double s_lone[];
    __shared__ int s_w[4];
    __shared__ int s_col[4];                             // the tile's column, its first slab row, slab rows, matrix group
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == gridDim.x - 1) {
        if (g.host_pub && tid == 0) publish_live_now(g);
        return;
    }
    const int ts = (g.max_slab + 15) / 16, tm = (g.max_main + 15) / 16;
    const int nct = (g.D + GEMM_BN - 1) / GEMM_BN;
    const int id = blockIdx.x;
    const int xq = (id / (8 * nct)) * 8 + (id & 7), bn0 = ((id >> 3) % nct) * GEMM_BN;
    const int cap = g.live_cap > 0 ? g.live_cap : 1;
    const int tq = xq / cap, ci = xq % cap;
    if (tq >= ts + tm) return;                           // (uniform) padding of the grid
    const int tt = tq < tm ? ts + tq : tq - tm;          // tt < ts: slab tile tt; else plain tile tt - ts (the short slab tiles last)
    // The ci-th live column of the launch and, in the same round trip, what the tile needs to know of it: every thread looks at
    // one candidate column's flag and descriptors.  The requests of the first 256 candidates leave before anything else ...
    int fl = 0, iu_c = 0, id_c = -1, mg_c = 0;
    if (tid < g.B) {
        const int c = g.col0 + tid;
        fl = g.active[c];
        if (g.idx_up) {
            iu_c = g.idx_up[c]; id_c = g.idx_down[c];
            if (g.Wmix) mg_c = g.mix_group[c];
        }
    }
    // ... and behind them, for a plain tile, the first two register blocks of its matrix: W_atm whatever the column
    LoneFrag f0, f1;
    const int nkb = (g.Ks >> 2) / LONE_KB;
    if (tt >= ts) {
        const int mcol = (bn0 >> 1) + (__builtin_amdgcn_readfirstlane(tid) >> 6) * 16 + (lane & 15);
        const __amdgpu_buffer_rsrc_t rW = make_rsrc(g.Wa, g.Dp * g.Wld * 8);
        lone_load(f0, rW, 0, lane >> 4, g.Wld, g.Wld >> 1, mcol);
        if (nkb > 1) lone_load(f1, rW, 1, lane >> 4, g.Wld, g.Wld >> 1, mcol);
    }
    if (tid == 0) s_col[0] = -1;
    int before = 0;
    for (int base = 0;; base += 256) {
        const bool f = fl != 0;
        const unsigned long long mk = __ballot(f);
        if (lane == 0) s_w[wave] = __popcll(mk);
        __syncthreads();
        int pre = before, tot = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) pre += s_w[w];
            tot += s_w[w];
        }
        if (f && pre + __popcll(mk & ((2ull << lane) - 1)) == ci + 1) {
            s_col[0] = g.col0 + base + tid; s_col[1] = iu_c; s_col[2] = id_c - iu_c + 1; s_col[3] = mg_c;
        }
        before += tot;
        __syncthreads();
        if (before > ci || base + 256 >= g.B) break;
        fl = 0;                                          // the next 256 candidates (a group of more than 256 columns)
        if (base + 256 + tid < g.B) {
            const int c = g.col0 + base + 256 + tid;
            fl = g.active[c];
            if (g.idx_up) {
                iu_c = g.idx_up[c]; id_c = g.idx_down[c];
                if (g.Wmix) mg_c = g.mix_group[c];
            }
        }
    }
    const int b = s_col[0];
    // the transport of this order takes its columns from this list
    if (g.live_list && tq == 0 && bn0 == 0 && tid == 0 && ci < g.live_cap) g.live_list[ci] = b < 0 ? -1 : b - g.col0;
    if (b < 0) return;                                   // fewer live columns than the host's (lagging) count
    const int iu = g.idx_up ? s_col[1] : 0, ns = g.idx_up ? s_col[2] : 0, mg = s_col[3];
    if (tt < ts) {
        if (tt * 16 >= ns) return;
        if (g.Wmix) gemm_tile_lone<false, false>(g, s_lone, tt, bn0, ColumnRows{b * g.L, iu, ns, ns, true}, f0, f1, g.Wmix + (size_t)mg * g.Dp * g.Wld);
        else gemm_tile_lone<true, false>(g, s_lone, tt, bn0, ColumnRows{b * g.L, iu, ns, ns, true}, f0, f1);
    } else {
        if ((tt - ts) * 16 >= g.L - ns) return;
        gemm_tile_lone<false, true>(g, s_lone, tt - ts, bn0, ColumnRows{b * g.L, iu, ns, g.L - ns, false}, f0, f1);
    }
}

}  // namespace

void launch_gemm_tail(hipStream_t s, const GemmArgs& a, int cols, bool small_tiles, bool regs) {
    if (regs && a.sym) {
        const int ts = (a.max_slab + 15) / 16, tm = (a.max_main + 15) / 16;
        if (cols <= 0 || ts + tm <= 0) return;
        const int nct = (a.D + GEMM_BN - 1) / GEMM_BN;
        const int lds = 16 * (a.D + 2) * 8;
        static PerDeviceOnce big_lds;                           // (more than 64 KB of dynamic LDS from N = 256 on)
        if (big_lds.first()) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_jn_gemm_lone), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        }
        // (+ 1: the workgroup that reports to the host)
        hipLaunchKernelGGL(k_jn_gemm_lone, dim3((unsigned)((cols * (ts + tm) + 7) / 8 * 8 * nct + 1)), dim3(256), (size_t)lds, s, a);
        return;
    }
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
    const int tiles = (a.n_main + 16 * GEMM_RT - 1) / (16 * GEMM_RT) + (a.n_slab + DENSE_SLAB_ROWS - 1) / DENSE_SLAB_ROWS;
    if (tiles <= 0) return;
    const int nct = (a.D + GEMM_BN - 1) / GEMM_BN;
    // (+ 1 with a live list: the workgroup that writes it)
    dim3 grid((unsigned)((tiles + 7) / 8 * 8 * nct + (a.live_list ? 1 : 0)));
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
