// k_order_loop: the order loop of spec:309-458 for the LAST FEW LIVE COLUMNS of a batch, several orders in ONE launch.
//
// While hundreds of columns are live an order is two launches, the contraction (jn_gemm.hip, FP64 MFMA) and the transport
// (transport_ring.hip / transport_scan.hip, HBM), and the host spins on a pinned word between them (api.hip).  Once few
// columns are left, an order of the whole batch is the latency of ONE column's order -- 12 us of contraction + 27-31 us of
// transport on a lone column, of which ~8 us are the two launches themselves and the rest two kernels'
// ramps and a column's serial chain -- and twenty such orders are a quarter of a 512-column sweep.  Here the workgroups of one
// launch keep their roles for all the remaining orders:
//
//   transport workgroups   ceil(N / 64) per live column (the split form of the chunk-parallel kernel: transport_scan_body.hpp,
//                          FUSED), or one per column where the split form does not apply: order after order of their column;
//   contraction workgroups every other workgroup of the grid: the 32-row (slab: 16-row) tiles of the live columns' source
//                          function (jn_gemm_tile.hpp, COH), dealt round-robin, order after order.
//
// What ties them is per-column words in global memory (agent-scope atomics; every poll bounded):
//   ord_done[c]  orders of the launch the column has finished (its rows of In are final and in memory): a tile of order k
//                waits for k - 1;   col_stop[c]  the column has converged (or failed, or spent its order budget): its tiles
//                are skipped;   jn_done[c]  tiles of the column's source function stored so far: the transport of order k
//                waits for k x (tiles of the column).
// The rows that cross between the roles are stored write-through and loaded `sc1` on both sides (the XCDs' L2s are not
// coherent; MI355X_MICROARCH: stores acknowledged by every storing wave, then a barrier, then ONE lane's counter).  The
// arithmetic of both roles is the code of the one-order kernels, so a column has the same bits whether its last orders ran
// here or there (test_order_loop_kernel_keeps_the_bits).
//
// Residency.  Workgroups that wait for each other must all be on the machine.  The grid is at most one workgroup per CU of the
// share the host reserves for it (api.hip: a per-device budget across the handles of the process), and the launch opens with a
// HANDSHAKE: every workgroup arrives at a counter; the last one declares the launch READY; a workgroup that has waited kHandshakeTicks
// for that declares it NOT RESIDENT instead (another process's kernels hold CUs) -- one compare-and-swap decides, nothing has
// been touched yet, everybody leaves and the host goes on with the one-order kernels.  The last workgroup to leave reports
// {status, tag} to pinned host memory.
#include "jn_gemm_tile.hpp"
#include "transport_scan_body.hpp"

namespace sosrt {

namespace {

constexpr long long kHandshakeTicks = 400000;     // wall_clock64 ticks (100 MHz): 4 ms
#ifndef SOSRT_OL_DEEP
#define SOSRT_OL_DEEP 1
#endif
constexpr bool kOlDeep = SOSRT_OL_DEEP != 0;      // contraction tiles staged two chunks ahead (jn_gemm_tile.hpp, DEEP)

template <bool SPLIT, int NC>
__global__ __launch_bounds__(768) void k_order_loop(OrderLoopArgs p) {
    const Grid& g = p.t.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int nthreads = blockDim.x, nw = nthreads >> 6;
    const int G = gridDim.x;
    const int N = NC ? NC : g.N, L = g.L;
    const int parts = SPLIT ? (N + 63) >> 6 : 1;
    int* const sy = p.sync;
    int* const abort = sy + kOlAbort;
    __shared__ int s_live[kOrderLoopMaxCols];
    __shared__ int s_w[12];
    __shared__ int s_misc[4];

    // ---- handshake: is the whole grid on the machine? ----
    if (tid == 0) {
        const int arrived = __hip_atomic_fetch_add(sy + kOlArrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        int st = 0;
        if (arrived == G) {
            int expect = 0;
            __hip_atomic_compare_exchange_strong(sy + kOlState, &expect, kOlReady, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const long long t0 = wall_clock64();
        for (;;) {
            st = __hip_atomic_load(sy + kOlState, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st) break;
            if (wall_clock64() - t0 > kHandshakeTicks) {
                int expect = 0;
                __hip_atomic_compare_exchange_strong(sy + kOlState, &expect, kOlNotResident, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_s_sleep(16);
        }
        s_misc[0] = st;
    }
    __syncthreads();
    const int state = s_misc[0];

    // ---- the live columns of this column group, in order (every workgroup builds the same list) ----
    int ncol = 0;
    if (state == kOlReady) {
        int before = 0;
        for (int base = 0; base < p.B; base += nthreads) {
            const bool f = base + tid < p.B && p.t.cv.active[base + tid] != 0;
            const unsigned long long mk = __ballot(f);
            if (lane == 0) s_w[wid] = __popcll(mk);
            __syncthreads();
            int pre = before, tot = 0;
            for (int w = 0; w < nw; ++w) {
                if (w < wid) pre += s_w[w];
                tot += s_w[w];
            }
            const int pos = pre + __popcll(mk & ((1ull << lane) - 1));
            if (f && pos < kOrderLoopMaxCols) s_live[pos] = base + tid;
            before += tot;
            __syncthreads();
        }
        ncol = before;
    }
    const int NT = ncol * parts;                       // transport workgroups; the rest contract
    const bool run = state == kOlReady && ncol > 0 && ncol <= kOrderLoopMaxCols && NT < G;
    if (state == kOlReady && ncol > 0 && !run && blockIdx.x == 0 && tid == 0)    // (never: the host sizes the grid from an upper bound of the live count)
        __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    if (run) {
        const int wg = blockIdx.x;
        const int ts = (p.gm.max_slab + 16 * TAIL_RT_SLAB - 1) / (16 * TAIL_RT_SLAB);
        const int tm = (p.gm.max_main + 16 * TAIL_RT - 1) / (16 * TAIL_RT);
        const int nct = (p.gm.D + GEMM_BN - 1) / GEMM_BN;
        // tiles of a column's source function (ColumnRows: its slab rows in 16-row tiles, its plain rows in 32-row tiles)
        auto tiles_of = [&](int bg) {
            const int iu = p.gm.idx_up ? p.gm.idx_up[bg] : 0;
            const int ns = p.gm.idx_up ? p.gm.idx_down[bg] - iu + 1 : 0;
            return ((ns + 16 * TAIL_RT_SLAB - 1) / (16 * TAIL_RT_SLAB) + (L - ns + 16 * TAIL_RT - 1) / (16 * TAIL_RT)) * nct;
        };
#ifndef SOSRT_OL_ONLY
#define SOSRT_OL_ONLY 0
#endif
        if (wg < NT && SOSRT_OL_ONLY != 2) {
            // ------------------------------- transport: one column, order after order -------------------------------
            const int ci = wg / parts, part = wg - ci * parts;
            const int b = s_live[ci];
            int* const cs = sy + kOlCols + ci * kOlColStride;
            const int T = tiles_of(p.col0 + b);
            ScanFused fu;
            fu.jn_done = cs + kOlJnDone; fu.ord_done = cs + kOlOrdDone; fu.col_stop = cs + kOlColStop; fu.abort = abort; fu.orders = sy + kOlOrders;
            fu.log = p.log; fu.wg = wg;
            for (int k = 1; k <= p.kmax; ++k) {
                // An order's body is the one-order kernel's, and is to be compiled like it: nothing of it hoisted out of this
                // loop (per-lane constants, descriptors and table addresses kept live across the whole body cost the sweeps
                // their registers: 168 + 32 spilled against 129).  The column and the pointers are laundered per order.
                TransportArgs a = p.t;
                int bb = b;
                asm volatile("" : "+s"(bb), "+s"(a.g.mu), "+s"(a.desc), "+s"(a.tau), "+s"(a.g.fix), "+s"(a.Etab), "+s"(a.I), "+s"(a.Jn));
                a.In = (k & 1) ? p.bufP : p.bufQ;
                a.order = p.order0 + k - 1;
                fu.k = k; fu.first = k == 1; fu.last = k == p.kmax; fu.jn_need = k * T;
                __syncthreads();                       // (the end of the previous order has read what this one's head resets)
                if (tid == 0) SOSRT_OL_STAMP(p.log, wg, k, 1);
                if (transport_scan_order<true, false, SPLIT, NC, false, true>(a, p.fixcap, bb, part, fu)) break;
            }
        } else if (SOSRT_OL_ONLY != 1) {
            // ------------------------------- contraction: tiles of every live column, order after order -------------------------------
            // (four waves of the workgroup; the others END here -- a wave parked at a later barrier would be counted by every
            // barrier of the tiles)
            if (tid >= 256) return;
            extern __shared__ double sm[];
            double* sA = sm;
            double* sB = sm + 2 * 16 * TAIL_RT * A_LD;
            double* sRaw = sB + GEMM_KC * B_LD;          // [16 TAIL_RT][D + 2]: the tile's rows of In_1, whole (jn_gemm_tile.hpp, ASTAGE)
            const bool astage = p.astage != 0;           // (they fit the launch's LDS: N <= 128 beside the transport role's need)
            const int NW = G - NT, w = wg - NT;
            const int TPC = (ts + tm) * nct;           // schedule slots per column (a column with fewer rows leaves some empty)
            GemmArgs gk = p.gm;
            for (int k = 1; k <= p.kmax; ++k) {
                // every column stopped: done
                if (tid == 0) {
                    int live = 0;
                    for (int c = 0; c < ncol; ++c)
                        live += __hip_atomic_load(sy + kOlCols + c * kOlColStride + kOlColStop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0 : 1;
                    s_misc[1] = live && !__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                if (!s_misc[1]) break;
                gk.A = k == 1 ? p.in0 : (((k - 1) & 1) ? p.gbufP : p.gbufQ);
                for (int slot = w; slot < ncol * TPC; slot += NW) {
                    // a column's tiles go to different workgroups (column index fastest)
                    const int ci = slot % ncol, tq = slot / ncol;
                    const int tt = tq / nct, bn0 = (tq % nct) * GEMM_BN;
                    const int bg = p.col0 + s_live[ci];
                    const int iu = p.gm.idx_up ? p.gm.idx_up[bg] : 0;
                    const int ns = p.gm.idx_up ? p.gm.idx_down[bg] - iu + 1 : 0;
                    if (tt < ts ? tt * 16 * TAIL_RT_SLAB >= ns : (tt - ts) * 16 * TAIL_RT >= L - ns) continue;    // (uniform) no such tile
                    int* const cs = sy + kOlCols + ci * kOlColStride;
                    __syncthreads();                   // (s_misc[2] of the previous tile has been read)
                    if (tid == 0) {
                        // the rows of order k - 1 are final and in memory (k = 1: they were when the launch started) -- or the
                        // column has stopped at an earlier order (col_stop is written before ord_done, so a column seen to
                        // have finished order k - 1 without the flag goes on to order k)
                        int go = 0;
                        for (unsigned it = 0;; ++it) {
                            const int od = __hip_atomic_load(cs + kOlOrdDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const int st = __hip_atomic_load(cs + kOlColStop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (st) break;
                            if (od >= k - 1) {
                                // (read in this order a stop published together with order k - 1 could be missed: look again)
                                go = !__hip_atomic_load(cs + kOlColStop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                            if ((it & 63) == 63 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                            if (it > kFusedSpinLimit) { __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                            __builtin_amdgcn_s_sleep(8);
                        }
                        s_misc[2] = go;
                        SOSRT_OL_STAMP(p.log, wg, k, 10);                                   // tile's rows ready (or column stopped)
                    }
                    __syncthreads();
                    if (!s_misc[2]) continue;
                    if (tt < ts) {
                        const ColumnRows cr{bg * L, iu, ns, ns, true};
                        const double* wm = p.gm.Wmix ? p.gm.Wmix + (size_t)p.gm.mix_group[bg] * p.gm.Dp * p.gm.Wld : nullptr;
                        if (wm && astage) gemm_tile<TAIL_RT_SLAB, false, true, true, ColumnRows, true, true>(gk, sA, sB, nullptr, tt, bn0, cr, false, wm, sRaw);
                        else if (wm) gemm_tile<TAIL_RT_SLAB, false, kOlDeep, true, ColumnRows, true>(gk, sA, sB, nullptr, tt, bn0, cr, false, wm);
                        else gemm_tile<TAIL_RT_SLAB, true, kOlDeep, true, ColumnRows, true>(gk, sA, sB, nullptr, tt, bn0, cr, false);
                    } else {
                        const ColumnRows cr{bg * L, iu, ns, L - ns, false};
                        if (astage) gemm_tile<TAIL_RT, false, true, true, ColumnRows, true, true>(gk, sA, sB, nullptr, tt - ts, bn0, cr, false, nullptr, sRaw);
                        else gemm_tile<TAIL_RT, false, kOlDeep, true, ColumnRows, true>(gk, sA, sB, nullptr, tt - ts, bn0, cr, false);
                    }
                    // the tile's rows (write-through) acknowledged by every wave, then one lane counts it
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) {
                        __hip_atomic_fetch_add(cs + kOlJnDone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        SOSRT_OL_STAMP(p.log, wg, k, 11);                                   // tile stored and counted
                    }
                }
            }
        }
    }
    // ---- the last workgroup to leave reports to the host ----
    __syncthreads();
    if (tid == 0) {
        const int left = __hip_atomic_fetch_add(sy + kOlLeft, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (left == G && p.host_done) {
            const int ab = __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int status = state != kOlReady ? kOlNotResident : (ab ? kOlAborted : kOlReady);
            __hip_atomic_store(p.host_done, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(p.host_done + 1, p.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <bool SPLIT>
unsigned order_loop_block(const Grid& g) {
    const int nwc = SPLIT ? 1 : (g.N + 63) / 64;
    return (unsigned)((nwc * ScanCfg<SPLIT>::SW + NLOAD) * 64);
}
size_t order_loop_gemm_lds() { return (size_t)(2 * 16 * TAIL_RT * A_LD + GEMM_KC * B_LD) * sizeof(double); }
size_t order_loop_raw_lds(const Grid& g) { return (size_t)16 * TAIL_RT * (g.D + 2) * sizeof(double); }

template <bool SPLIT, int NC>
hipError_t launch_order_loop_t(hipStream_t s, int grid, const OrderLoopArgs& p) {
    auto kern = k_order_loop<SPLIT, NC>;
    static PerDeviceOnce big_lds;      
    if (big_lds.first()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScanLdsBytes);
        if (e != hipSuccess) return e;
    }
    size_t shm = scan_lds_bytes<SPLIT>(p.t.g, kRingZones);
    if (shm < order_loop_gemm_lds()) shm = order_loop_gemm_lds();
    OrderLoopArgs q = p;
    // the contraction role's whole-tile staging of In_1, where it fits the CU's LDS (N <= 128; a launch is one workgroup per CU)
    q.astage = order_loop_gemm_lds() + order_loop_raw_lds(p.t.g) <= kScanLdsBytes ? 1 : 0;
    if (q.astage && shm < order_loop_gemm_lds() + order_loop_raw_lds(p.t.g)) shm = order_loop_gemm_lds() + order_loop_raw_lds(p.t.g);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(order_loop_block<SPLIT>(p.t.g)), shm, s, q);
    return hipGetLastError();
}

}  // namespace

// whether the order-loop kernel takes this shape (the handle-level conditions -- symmetric contraction, zone layout, no saved
// orders -- are the caller's): the chunk-parallel transport in its split form (two to four workgroups per column; specular
// surface or none) or, failing that, with one workgroup per column
bool order_loop_ok(const Grid& g, bool split) {
    if (g.nsmall != 0) return false;                    // (the caller passes nsmall = 0 once no |mu| < 0.01 lane keeps its k_smallmu value)
    return split ? (transport_scan_split_ok(g) && transport_scan_fits(g, kRingZones, true))
                 : (transport_scan_ok(g) && transport_scan_fits(g, kRingZones, false));
}

int order_loop_parts(const Grid& g, bool split) { return split ? transport_scan_parts(g) : 1; }

hipError_t launch_order_loop(hipStream_t s, int grid, bool split, const OrderLoopArgs& p) {
    if (split) {
        if (p.t.g.N == 128) return launch_order_loop_t<true, 128>(s, grid, p);
        if (p.t.g.N == 256) return launch_order_loop_t<true, 256>(s, grid, p);
        return launch_order_loop_t<true, 0>(s, grid, p);
    }
    return launch_order_loop_t<false, 0>(s, grid, p);
}

}  // namespace sosrt
