// Device-side data structures and kernel launchers (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include "plan.hpp"

namespace sosrt {

constexpr int kMaxZones = 8;       // zones of a column, top to bottom: clear / slab / clear / slab ... (up to four aerosol slabs)
constexpr int kRingZones = 3;      // zones the register-streaming kernel (fast) handles, and the zones the ring / chunk-parallel kernels keep in scalars

// One independent SOS problem as the kernels see it: a zone table instead of the reference's
// three copies of every formula (spec:113-449).
struct alignas(16) ColDesc {
    int nz;                    // zones, top to bottom
    int r0[kMaxZones];         // first / last row of each zone
    int r1[kMaxZones];
    int mix[kMaxZones];        // 1: aerosol slab (phase mix of spec:149,321)
    int nfix[kMaxZones];       // rewritten downward angles next to mu=0- (spec:342-345,361-364,380-383)
    int fixtab[kMaxZones];     // which extrapolation table
    int surface;               // SOSRT_SURFACE_*
    int geom;
    double mu0, rho, T, tau_bottom;
    double wa;                 // single-scattering albedo of the molecules
    double wr[kMaxZones];      // per zone: single-scattering albedo of its aerosol, slab mixing fractions (spec:50-53,149)
    double fa[kMaxZones];
    double fr[kMaxZones];
    double ca[kMaxZones];      // Jn coefficient on In_1 @ W_atm per zone (spec:321,323)
    double cr[kMaxZones];      // Jn coefficient on In_1 @ W_aer per zone
};

struct FixTab {
    int idx, s0, ns, pad;
    double C[kFixMaxIdx * kFixMaxSrc];
};

// per-column scalars as uploaded by sosrt_set_columns (structure of arrays)
struct ColScalars {
    // zone table of every column ([max_batch][kMaxZones]; filled from idx_up / idx_down by sosrt_set_columns, or given by
    // the caller through sosrt_set_columns_zones): first row of each zone, 1 for an aerosol slab, the slab's aerosol
    // single-scattering albedo and optical-depth step
    const int* nz;
    const int* zr0;
    const int* zmix;
    const double* zwr;
    const double* zdtr;
    const int* idx_up;
    const int* idx_down;
    const double* mu0;
    const double* rho;
    const double* alb_atm;
    const double* alb_aer;
    const double* dtau_atm;
    const double* dtau_aer;
    const double* T;
};

struct Grid {
    int L, N, D;
    int Dp;                    // D rounded up to the GEMM k-chunk
    int Wld;                   // leading dimension of the folded matrices (multiple of the GEMM column tile)
    const double* mu;          // [D]
    const double* Wa;          // [Dp][Wld]
    const double* Wr;          // [Dp][Wld]
    const FixTab* fix;         // [4]
    const int* small_lanes;
    int nsmall;
    const double* wflux_dn;    // [N]
    const double* wflux_up;    // [N]
};

struct Conv {
    int* active;               // [B] 1 while the column still iterates
    int* norders;              // [B] n of spec:307
    int* status;               // [B]
    int* nactive;              // [1]
    double* ratio;             // [B] last value of the spec:309 test
    int* redo;                 // [B] set by k_transport_fast when the column needs the general upward sweep
    double tol;
};

constexpr int GEMM_BM = 64, GEMM_BN = 128, GEMM_KC = 16;
#ifndef SOSRT_GEMM_RT
#define SOSRT_GEMM_RT 4
#endif
#ifndef SOSRT_GEMM_WPS
#define SOSRT_GEMM_WPS 3
#endif
constexpr int GEMM_RT = SOSRT_GEMM_RT;     // MFMA row tiles per wave of the plain-row workgroups
constexpr int GEMM_WPS = SOSRT_GEMM_WPS;   // waves per SIMD the register budget is set for

// Transport rows are processed in chunks of TC: the loads of the next chunks are in flight while
// the current one is computed, the per-row source terms of a chunk are independent
// (instruction-level parallelism for the one wave per SIMD a column gives), only the TC fused
// multiply-adds of the recurrence are sequential.
constexpr int TC = 8;

struct TransportArgs {
    Grid g;
    const double* tau;
    const double* Jn;
    double* In;
    double* I;
    double* saved;
    size_t saved_col_stride;
    const ColDesc* desc;
    Conv cv;
    int order;
    int accumulate;
    const double* Etab;
    const int* erep;           // [B] column whose attenuation table this column uses (same tau profile); null: its own
    unsigned long long* stamps;   // diagnostic builds only: [B][8] cycle stamps of the sweeps (nullable)
    // ring kernel, some columns converged: `live` > 0 launches workgroups for (an upper bound of) the live columns only;
    // workgroup i takes column live_list[i], the list the source-function launch of the same order wrote (the flags
    // themselves change while the transport runs).  Besides skipping the idle workgroups this spreads the live
    // columns evenly over the XCDs (workgroups are dealt round-robin over them, and the slow-converging columns of a
    // sweep tend to share their index modulo 8).
    int live = 0;
    const int* live_list = nullptr;
    int slots = 0;             // ring depth of this launch (0: the default)
    // most zones of any column of the batch (>= kRingZones): sizes the per-zone tables of the ring / chunk-parallel kernels and
    // selects their instantiation (beyond three zones: the one that tests the other boundaries too)
    int nzcap = kRingZones;
    // chunk-parallel kernel, ceil(N / 64) workgroups per column (transport_scan.hip): exchange rows and {arrivals, flags} per column
    int scan_split = 0;
    double* scan_scratch = nullptr;
    int* scan_sync = nullptr;
};
void launch_transport_fast(hipStream_t s, dim3 grid, dim3 block, const TransportArgs& a);
// transport_ring.hip: same sweeps, rows streamed through an LDS ring by loader waves
bool transport_ring_ok(const Grid& g);
bool transport_ring_fits(const Grid& g, int nzcap);            // ... with the per-zone tables of up to nzcap zones
void launch_transport_ring(hipStream_t s, dim3 grid, const TransportArgs& a, int slots);
// transport_scan.hip: the chunks of a sweep dealt to several waves (chunk-local recurrence + carried values)
bool transport_scan_ok(const Grid& g);
bool transport_scan_split_ok(const Grid& g);
bool transport_scan_fits(const Grid& g, int nzcap, bool split);
int transport_scan_parts(const Grid& g);                       // workgroups per column of the split form: ceil(N / 64)
size_t transport_scan_scratch_doubles();                       // per column
void launch_transport_scan(hipStream_t s, dim3 grid, const TransportArgs& a);
// order_loop.hip: the last orders of a few live columns in ONE launch (transport and contraction as roles of its workgroups)
constexpr int kOrderLoopMaxCols = 256;                 // live columns a launch takes at most
// words of the launch in global memory (ints, zeroed before the launch): launch-wide ones on lines of their own, then per live column
constexpr int kOlArrive = 0, kOlState = 32, kOlLeft = 64, kOlAbort = 96, kOlOrders = 100, kOlCols = 128;
constexpr int kOlColStride = 64, kOlOrdDone = 0, kOlColStop = 1, kOlJnDone = 32;
constexpr int kOlReady = 1, kOlNotResident = 2, kOlAborted = 3;            // state of a launch, as reported to the host
inline size_t order_loop_sync_ints(int cols) { return (size_t)kOlCols + (size_t)cols * kOlColStride; }
extern int g_ring_slots, g_ring_debug;                        // tuning (SOSRT_RING_SLOTS)
extern unsigned long long* g_transport_stamps;   // diagnostics (sosrt_debug_stamps)

// what the first kernel of a solve does besides the zone tables (all null: nothing)
struct SolveSetup {
    int* zero_next = nullptr;            // [n_zero] counters of the next solve
    int n_zero = 0;
    int* redo = nullptr;                 // [B] per-column redo flags, cleared
    unsigned long long* hash = nullptr;  // [B] hash of the column's optical-depth profile
    int* scan_sync = nullptr;            // [B][2] exchange words of the chunk-parallel transport's split form, cleared
};
void launch_prepare(hipStream_t s, const Grid& g, int B, int geom, int surface, ColScalars sc, const double* tau,
                    ColDesc* desc, double* rowcoef_a, double* rowcoef_r, int* need_small = nullptr, SolveSetup su = SolveSetup());
void launch_first_order_readme(hipStream_t s, const Grid& g, const double* w, int B, const double* tau, const double* P0a,
                               const double* P0r, const ColDesc* desc, double* I1_out, double* I_out, double* saved,
                               size_t saved_col_stride, Conv cv, int do_conv);
void launch_first_order(hipStream_t s, const Grid& g, int B, const double* tau, const double* P0a, const double* P0r,
                        const ColDesc* desc, double* I1_out, double* I_out, double* saved, size_t saved_col_stride,
                        Conv cv, int do_conv);
// Jn = diag(ca) A Wa for the rows of `rows_main`, and diag(ca) A Wa + diag(cr) A Wr for the rows of
// `rows_slab` (spec:321), in one launch.  Row lists hold global row ids; rows_main == nullptr means
// the identity list 0..n_main-1.
struct GemmArgs {
    const double* A;
    const double* Wa;
    const double* Wr;
    const double* ca;      // per-row coefficients
    const double* cr;
    const int* rows_main;
    int n_main;
    const int* rows_slab;
    int n_slab;
    int D, Dp, Wld, L;
    double* C;
    const int* active;     // nullable: skip tiles whose columns have all converged
    // k_jn_gemm_tail only: tiles are laid over the live columns (the i-th group of workgroups finds the
    // i-th live column itself), not over the row lists
    int B = 0;
    int col0 = 0;                    // the launch covers the columns [col0, col0 + B) of the batch (all per-column arrays are whole-batch)
    const int* idx_up = nullptr;     // [batch] first / last slab row of a column; null: no slab rows
    const int* idx_down = nullptr;
    int max_main = 0, max_slab = 0;  // most plain / slab rows any column has
    const double* Wmix = nullptr;    // [groups][Dp][Wld] ca W_atm + cr W_aer per distinct slab coefficient pair; null: two passes
    const int* mix_group = nullptr;  // [B] group of a column
    int pad_lds = 0;                 // host side: unused dynamic LDS added to the launch (caps the workgroups per CU)
    int* live_list = nullptr;        // live-column tilings: [live_cap] the i-th live column (relative to col0), -1 beyond the live count
    int live_cap = 0;
    const int* slab_tile_group = nullptr;   // dense kernel: group of every 32-row tile of rows_slab (listed group by group, padded with -1)
    // The order loop's view of the batch: the first workgroup of the source-function launch of order
    // n+1 (which starts when order n has finished) writes {live columns after order n, tag} to pinned
    // host memory, where the host spins on the tag -- no copy, no event, no stream drain.
    const int* nactive = nullptr;    // [1]: live columns of this column group
    const int* need_small = nullptr; // [1]: whether any column of the batch needs k_smallmu (set by k_prepare)
    int* host_pub = nullptr;         // pinned [2 slots][4] = {live, tag, needs k_smallmu, -}; slot = tag & 1
    int tag = 0;
    // flip-symmetric contraction (jn_gemm.hip, SYM): Wa / Wr / Wmix then hold the folded matrices [k < Ks][S | A]
    int sym = 0;
    int Ks = 0;                      // N rounded up to the k-chunk
    int check_tiles = 1;             // dense kernel: skip the tiles whose columns have all converged (two barriers and a dependent load per tile)
};

// (publish_live_now: by the calling thread, whichever workgroup it belongs to)
__device__ inline void publish_live_now(const GemmArgs& g) {
    const int live = __hip_atomic_load(g.nactive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int* slot = g.host_pub + 4 * (g.tag & 1);
    __hip_atomic_store(slot, live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot + 2, g.need_small ? g.need_small[0] : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot + 1, g.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void publish_live(const GemmArgs& g) {
    if (g.host_pub && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) publish_live_now(g);
}
// A kernel's opt-in to more than 64 KB of dynamic LDS (hipFuncSetAttribute) is made once per DEVICE: a process may hold handles on
// several GPUs (sosrt_create(device = k)), and the attribute set under one device does not reach the kernel's object on another.
struct PerDeviceOnce {
    bool seen[64] = {};
    bool first() {                                   // true the first time the current device asks
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        if (seen[d]) return false;
        seen[d] = true;
        return true;
    }
};
void launch_gemm(hipStream_t s, const GemmArgs& a);
struct OrderLoopArgs {
    TransportArgs t;           // the column group's view (every per-column pointer offset to its first column): g, tau, Jn, I, desc, cv, Etab, erep, scan_scratch, scan_sync
    GemmArgs gm;               // the batch's view (whole-batch pointers, global column ids): folded matrices, Wmix, mix_group, idx_up / idx_down, ca / cr, C = Jn
    const double* in0;         // operand of the launch's first contraction (whole batch)
    double* gbufP;             // In of the launch's odd / even orders (whole batch) ...
    double* gbufQ;
    double* bufP;              // ... and the same from the group's first column
    double* bufQ;
    int order0;                // scattering order n of the launch's first order
    int kmax;                  // orders the launch runs at most
    int B, col0;               // the group's columns: [col0, col0 + B) of the batch
    int fixcap;
    int* sync;                 // order_loop_sync_ints(cap) zeroed ints
    int* host_done;            // pinned {status, tag}
    int tag;
    unsigned long long* log = nullptr;   // diagnostic builds (-DSOSRT_OL_STAMPS): [1 + 65000] event log, log[0] = count
    int astage = 0;            // (set by the launcher) the contraction role stages its tiles' rows of In_1 whole in LDS
};
bool order_loop_ok(const Grid& g, bool split);
int order_loop_parts(const Grid& g, bool split);
hipError_t launch_order_loop(hipStream_t s, int grid, bool split, const OrderLoopArgs& p);
// jn_gemm_f32.hip: the contraction with float operands and a float accumulator (opt-in, tolerance study)
void launch_gemm_f32(hipStream_t s, const GemmArgs& a, const float* Wa32, const float* Wmix32);
void launch_to_float(hipStream_t s, size_t n, const double* src, float* dst);
void launch_wmix(hipStream_t s, size_t n, int ngroups, const double* Wa, const double* Wr, const double* ca, const double* cr,
                 double* Wmix);
// the flip-symmetric folding [k][S | A] of `nmat` contraction matrices [Dp][Wld]
void launch_symfold(hipStream_t s, int nmat, int N, int D, int Dp, int Wld, const double* W, double* SA);
// some columns have converged (at most `cols` are live, an upper bound): workgroups only for live
// columns; small_tiles: 32-row tiles and deeper staging for the last few
void launch_gemm_tail(hipStream_t s, const GemmArgs& a, int cols, bool small_tiles, bool regs = false);
void launch_smallmu(hipStream_t s, const Grid& g, int B, const double* tau, const double* Jn, double* In,
                    const ColDesc* desc, const int* active);
void launch_transport(hipStream_t s, const Grid& g, int B, const double* tau, const double* Jn, double* In, double* I,
                      double* saved, size_t saved_col_stride, const ColDesc* desc, Conv cv, int order, int accumulate,
                      const double* Etab, int mode, const int* erep = nullptr, int live = 0, const int* live_list = nullptr,
                      int ring_slots = 0, int scan_split = 0, double* scan_scratch = nullptr, int* scan_sync = nullptr,
                      int nzcap = kRingZones);
bool transport_fast_ok(const Plan& plan);
// erep (nullable): tables are built only for columns with erep[b] == b
void launch_attenuation(hipStream_t s, const Grid& g, int B, const double* tau, double* Etab, const int* erep);
// erep[b] = first column with the same optical-depth profile as column b (hash[B] is scratch)
// need_small / host_flag / tag (nullable): publishes k_prepare's verdict on k_smallmu to pinned host memory {flag, tag}
void launch_tau_groups(hipStream_t s, const Grid& g, int B, const double* tau, unsigned long long* hash, int* erep,
                       const int* need_small = nullptr, int* host_flag = nullptr, int tag = 0);
void launch_fluxes(hipStream_t s, const Grid& g, int B, const double* tau, const double* I, const ColDesc* desc,
                   int beam_norm, double* fdn, double* fup);
// epilogue.hip: outputs of the epilogue (each nullable): [B][L] per level, net_toa [B]
struct EpilogueOut {
    double* flux_down;
    double* flux_up;
    double* diffusivity;
    double* heating_rate;
    double* net_toa;
};
// w: np.trapz weights on the whole direction grid [D]; z_profile [L] (nullable: no heating rate)
void launch_epilogue(hipStream_t s, const Grid& g, const double* w, int B, const double* tau, const double* I,
                     const ColDesc* desc, int beam_norm, const double* z_profile, const EpilogueOut& out);
// phase functions on the device; cosphi / wphi: cos(phi) and trapz weights of phi = linspace(0, pi, nphi)
void launch_phase_p0(hipStream_t s, const Grid& g, const double* w, int B, int kind, double gpar, const double* tab_mu,
                     const double* tab_p, int ntab, const double* cosphi, const double* wphi, int nphi,
                     const double* mu0, double* P0);
void launch_phase_matrix(hipStream_t s, const Grid& g, const double* w, int kind, double gpar, const double* tab_mu,
                         const double* tab_p, int ntab, const double* cosphi, const double* wphi, int nphi, double* P);
void launch_limit_rows(hipStream_t s, const Grid& g, int R, int table, const double* rows, double* out);
void launch_asymptotic(hipStream_t s, int R, int stride, const int* len, const double* J, const double* tau,
                       const double* tau_t, const double* mu, double* out);
void launch_finalize(hipStream_t s, int B, Conv cv, int max_orders, int* n_out = nullptr, int* status_out = nullptr);
void launch_bench(hipStream_t s, int which, double* a, double* b, size_t n, int iters);

}  // namespace sosrt
