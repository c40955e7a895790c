// C ABI of libsosrt.so (see include/sosrt.h).  Host logic only: argument checks, device
// buffers, the order loop with lagged convergence polling, HIP-event profiling.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/sosrt.h"
#include "comm.hpp"
#include "kernels.hpp"
#include "plan.hpp"

using namespace sosrt;

namespace {
thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(x)                                                                                        \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) return fail(SOSRT_E_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                           \
    } while (0)

constexpr int kProfPool = 8192;
constexpr int kNPhi = 25;                // phase:81  nb_phi

// HIP-event timing of launch groups.  An interval is a pair of events of the pool; when one bracket
// closes and the next opens with nothing enqueued in between (the order loop: contraction, transport,
// contraction, ...), the closing event is the next opening one, which halves the markers in the stream.
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;          // pool
    std::vector<int> kind, first, last;  // per interval: kernel family, opening / closing event
    size_t used = 0, nint = 0;           // events / intervals used
    int open = -1;                       // opening event of the current bracket
    int adjacent = -1;                   // closing event of the previous bracket, if nothing was enqueued since
};
}  // namespace

struct sosrt_handle {
    int device = -1, L = 0, N = 0, D = 0, max_batch = 0, max_orders = 0;
    int order_budget = 0;                // orders a solve runs at most (sosrt_set_order_budget; <= max_orders, the default)
    int saved_slots = 0;                 // orders per column in I_saved_out (sosrt_set_saved_orders; default max_orders)
    bool gpu = false;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // Column groups of the order loop: a large batch is solved as two halves, the second on an internal stream, so
    // that the MFMA-bound contraction of one half overlaps the HBM-bound transport of the other (SOSRT_GROUPS)
    static constexpr int kMaxGroups = 2;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // (round 4, alternating runs on one box, shards of the EVA sweep: 32 columns 1.54 -> 1.48 ms, 64 columns 1.71 -> 1.66, 128 columns
    // 2.06 -> 1.96, 256 columns 2.98 -> 2.82 -- round 3 had only measured 256 and up with the capped contraction; tools/ab_small_groups.py)
    int ngroups = 1, want_groups = 0, split_min = 48;       // want_groups 0: two groups above split_min columns (set_columns)
    int split_at = -1;                   // SOSRT_GROUP_SPLIT: first column of the second group (default: the middle)
    int prio2 = 0;                       // SOSRT_GROUP_PRIO: the internal stream is created with the highest priority
    // Round 2 capped the contraction at two workgroups per CU with LDS padding (27 008 bytes: 27 656 static + this > 1/3 of 160 KiB)
    // and ran the ring two slots deep, so that a transport workgroup of the other group fits beside them on every CU.  With round 3's
    // kernels the uncapped contraction and a three-slot ring are faster under two groups (alternating runs: 512 columns 4.81 -> 4.75 ms,
    // 1024 columns 8.45 -> 8.15, 384 columns unchanged): the groups share the GPU CU by CU rather than inside a CU.
    int coresident_pad = 0;             // SOSRT_GEMM_PAD_LDS (diagnostic builds)
    int diag_ks_mult = 1;               // SOSRT_GEMM_KS_MULT (diagnostic builds)
    int coresident_slots = 3;           // SOSRT_GROUP_RING_SLOTS: ring depth of the transport under two groups
    double stagger = 1.0;                // a group starts when the previous one is down to this fraction of live columns (SOSRT_STAGGER; 1: together)
    int gb[kMaxGroups + 1] = {0, 0, 0};                    // column range of group g: [gb[g], gb[g+1])
    int main_off[kMaxGroups + 1] = {0, 0, 0};              // its plain rows in d_mainrows
    int slab_off[kMaxGroups + 1] = {0, 0, 0};              // its slab rows in d_slabrows (tile-aligned when grouped by coefficient pair)
    Plan plan;
    bool have_grid = false, have_phase = false, have_aer = false, have_cols = false;
    int B = 0, geom = 0, surface = 0;
    std::vector<double> Wa_h, Wr_h;
    Grid g{};
    // device: grid / phase
    double *d_mu = nullptr, *d_Wa = nullptr, *d_Wr = nullptr, *d_wfdn = nullptr, *d_wfup = nullptr;
    double *d_w = nullptr;               // [D] np.trapz weights on the whole grid
    int first_order_mode = SOSRT_FIRST_ORDER_CODED;   // sosrt_set_first_order
    double *d_phi = nullptr;             // [2][kNPhi] cos(phi), trapz weights of phi = linspace(0, pi, kNPhi) (phase:81-82)
    double *d_z = nullptr;               // [L] altitude grid of the host epilogue
    double *d_tab = nullptr;             // [2][ntab] table of SOSRT_PHASE_TABLE
    int ntab = 0;
    bool resident = false;               // d_tau / d_I hold the inputs / result of the last sosrt_solve (of resident_B columns)
    int resident_B = 0;
    FixTab* d_fix = nullptr;
    int* d_small = nullptr;
    // device: columns
    int *d_idx_up = nullptr, *d_idx_down = nullptr;
    int *d_nz = nullptr, *d_zr0 = nullptr, *d_zmix = nullptr;     // zone tables [max_batch][kMaxZones]
    double *d_zwr = nullptr, *d_zdtr = nullptr;
    int max_nz = 1;                      // most zones of any column (beyond three: the ring / chunk-parallel kernels' zone-table instantiation; general kernel where the register-streaming one would run)
    bool simple_zones = true;            // every column is (clear, slab, clear): the live-column tilings of the contraction apply
    double* d_scal = nullptr;            // 7 arrays of max_batch
    ColDesc* d_desc = nullptr;
    double *d_rca = nullptr, *d_rcr = nullptr;
    int* d_slabrows = nullptr;
    int* d_mainrows = nullptr;
    int nslab = 0, nmain = 0;
    int max_main = 0, max_slab = 0;      // most plain / slab rows of any column
    // slab rows of the live-column tilings: one pass over ca W_atm + cr W_aer per distinct (ca, cr) of the batch
    static constexpr int kMaxMixGroups = 32;
    int mix_groups = 0;                  // 0: disabled (too many distinct pairs, or no slab)
    bool mix_dirty = true;
    double *d_Wmix = nullptr, *d_mixca = nullptr, *d_mixcr = nullptr;
    int* d_mixgroup = nullptr;
    int* d_slabtilegroup = nullptr;      // [tiles] group of every 32-row slab tile of the dense contraction
    int contraction = SOSRT_CONTRACT_F64;
    // flip-symmetric contraction (jn_gemm.hip, SYM): folded copies [k][S | A] of W_atm, W_aer and the combined matrices
    double asymmetry = 0;                // max |W[k][m] - W[D-1-k][D-1-m]| / max |W| of the last sosrt_set_phase
    bool sym_ok = false;                 // asymmetry <= SOSRT_SYMMETRY_TOL
    bool sym_dirty = true, symmix_dirty = true;
    double *d_Wa_s = nullptr, *d_Wr_s = nullptr, *d_Wmix_s = nullptr;
    size_t mixs_capacity = 0;
    float *d_Wa32 = nullptr, *d_Wmix32 = nullptr;   // float copies of W_atm and of the combined slab matrices (SOSRT_CONTRACT_F32)
    size_t mix32_capacity = 0;
    bool w32_dirty = true;
    int* d_livelist = nullptr;           // [max_batch] live columns of the current order, written by the source-function launch
    size_t mix_capacity = 0;
    // device: fields (internal)
    double *d_tau = nullptr, *d_P0a = nullptr, *d_P0r = nullptr;
    double *d_Jn = nullptr, *d_InA = nullptr, *d_InB = nullptr, *d_I = nullptr, *d_E = nullptr;
    int use_etab = 1;
    // convergence
    int *d_active = nullptr, *d_norders = nullptr, *d_status = nullptr, *d_redo = nullptr, *d_erep = nullptr;
    // live columns per group + "some column needs k_smallmu": two sets used by alternate solves, the first kernel of a solve
    // zeroes the other set (no memset launch at the head of a solve); d_nactive points at the set of the current solve
    int *d_nactive_sets = nullptr, *d_nactive = nullptr;
    int nactive_set = 0;
    unsigned long long* d_tauhash = nullptr;
    // 0: general kernel, 1: wave-independent fast kernel (+ repair), 2: LDS-ring kernel, 3 (default): the ring kernel for
    // launches with many live columns (HBM-bound) and the chunk-parallel kernel (transport_scan.hip) for launches with at
    // most scan_cols (latency-bound), 4: the chunk-parallel kernel always.  The ring and the chunk-parallel kernel share
    // their arithmetic (chunk-local recurrence), so the choice follows the live count without touching a column's bits.
    int transport_mode = 3;
    int scan_cols = 200;                 // SOSRT_SCAN_COLS
    bool ring_ok = false, scan_ok = false, scan_split_ok = false;
    int scan_split = 1;                  // SOSRT_SCAN_SPLIT: two workgroups per column when at most half as many columns are live as the device has CUs
    int cu_count = 0;
    double* d_scan_scratch = nullptr;    // [max_batch][transport_scan_scratch_doubles()] exchange rows of the split form
    int* d_scan_sync = nullptr;          // [max_batch][2] {arrivals, flags}, zero between launches
    int gemm_tail_cols = 1 << 30;        // at or below this many live columns (and below the batch) tiles are laid over live columns (SOSRT_GEMM_TAIL)
    double gemm_tail_frac = 0.6;         // ... and at or below this fraction of the group's columns (SOSRT_GEMM_TAIL_FRAC): above it the dense
                                         // tiling, skipping the tiles of converged columns, is the faster one (contraction -3 % per step at 512 ... 4096 columns)
    int gemm_small_cols = 200;           // at or below this many, 32-row tiles (SOSRT_GEMM_SMALL)
    int dense_live_list = 1;             // the dense tiling writes the transport's live list (SOSRT_DENSE_LIVE_LIST=0: A/B)
    int gemm_regs_cols = -1;             // at or below this many (symmetric form), 16-row tiles with the matrix fragments in registers
                                         // (-1: while its workgroups, one per CU, are at most 1.5 rounds of the CUs; 0: never -- SOSRT_GEMM_REGS)
    bool fast_ok = false;
    double* d_ratio = nullptr;
    int* h_pub = nullptr;                // pinned [groups][2 slots][4]: {live count, tag, needs k_smallmu, -} published from the device
    bool need_small = true;              // some column keeps a k_smallmu value (known from the second order on)
    int pub_seq = 0;                     // tags are unique across solves
    int last_max_orders = 0;
    long long last_sum_orders = 0;
    Prof prof[kMaxGroups];               // per column group (= per stream)
    // order-loop kernel (order_loop.hip): the last orders of a few live columns in one launch
    // (OFF by default: measured on MI355X it is bit-identical and slower -- a lone column 54.6 us per order against 47.0 with two
    // launches, 64 columns 137 against 50: the chain sweep -> tile of the next source function -> sweep is the same either way,
    // what the launches cost (~8 us per order) the polls and the write-through hand-offs cost too, and the contraction role has one
    // four-wave team per CU; profiles/r04_order_loop_ab_v0.txt)
    int order_loop = 0;                  // sosrt_set_order_loop / SOSRT_ORDER_LOOP: 0 never (default), 1 where the launch plan says so
    double ol_frac = 0.5;                // ... while the transport workgroups of the live columns are at most this share of the grid
    int* d_olsync = nullptr;             // [kMaxGroups][order_loop_sync_ints(kOrderLoopMaxCols)] words of a launch
    int* h_oldone = nullptr;             // pinned [kMaxGroups][2]: {state, tag} reported by the launch's last workgroup
    unsigned long long* d_ollog = nullptr;   // diagnostic builds (-DSOSRT_OL_STAMPS): event log of an order-loop launch
    int ol_launches = 0, ol_refused = 0; // launches of the last solve; launches that found their grid not resident
    bool ol_group_used[kMaxGroups] = {false, false};      // column groups of the last solve that ran (to the end) in an order-loop launch
    // RCCL communicator of the sharded solve (sosrt_comm_init)
    Rccl::comm_t comm = nullptr;
    int comm_rank = -1, comm_world = 0;
};

namespace {

size_t field_elems(const sosrt_handle* h) { return (size_t)h->max_batch * h->L * h->D; }

template <class T>
int dalloc(T** p, size_t n) {
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) return fail(SOSRT_E_NOMEM, "hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
    return 0;
}

// the pools grow on demand (a long profiled run must not end up with timings of its first steps only)
bool prof_room(Prof& p) {
    if (p.used + 2 > p.ev.size()) {
        const size_t n0 = p.ev.size();
        p.ev.resize(n0 + 4096);
        for (size_t i = n0; i < p.ev.size(); ++i)
            if (hipEventCreateWithFlags(&p.ev[i], hipEventDisableSystemFence) != hipSuccess) { p.ev.resize(i); break; }
        if (p.used + 2 > p.ev.size()) return false;
    }
    if (p.nint >= p.kind.size()) {
        const size_t n = p.kind.size() + 4096;
        p.kind.resize(n, -1); p.first.resize(n, 0); p.last.resize(n, 0);
    }
    return true;
}
hipStream_t group_stream(sosrt_handle* h, int grp) { return grp == 0 ? h->stream : h->stream2; }
void prof_begin(sosrt_handle* h, int kind, int grp = 0) {
    Prof& p = h->prof[grp];
    p.open = -1;
    if (!p.on || !prof_room(p)) return;
    if (p.adjacent >= 0) {
        p.open = p.adjacent;
    } else {
        p.open = (int)p.used++;
        hipEventRecord(p.ev[p.open], group_stream(h, grp));
    }
}
void prof_end(sosrt_handle* h, int kind, int grp = 0) {
    Prof& p = h->prof[grp];
    if (!p.on || p.open < 0) return;
    const int e = (int)p.used++;
    hipEventRecord(p.ev[e], group_stream(h, grp));
    p.kind[p.nint] = kind; p.first[p.nint] = p.open; p.last[p.nint] = e;
    ++p.nint;
    p.adjacent = e;
    p.open = -1;
}
// work enqueued outside a bracket: the next bracket needs its own opening event
void prof_break(sosrt_handle* h) { for (auto& p : h->prof) p.adjacent = -1; }

int need_gpu(sosrt_handle* h) {
    if (!h) return fail(SOSRT_E_INVALID, "null handle");
    if (!h->gpu) return fail(SOSRT_E_STATE, "handle was created host-only (device < 0)");
    return 0;
}

// Live columns after the order whose tag is `tag`, as published by the source-function launch of the
// next order (publish_live in kernels.hpp).  Spins on pinned memory; negative = error code.
int wait_published(sosrt_handle* h, int grp, int tag) {
    volatile int* slot = h->h_pub + 8 * grp + 4 * (tag & 1);
    const auto t0 = std::chrono::steady_clock::now();
    auto next_query = t0 + std::chrono::milliseconds(20);
    for (unsigned it = 1;; ++it) {
        if (__atomic_load_n(&slot[1], __ATOMIC_ACQUIRE) == tag) {
            h->need_small = slot[2] != 0;
            return slot[0];
        }
        // The stream is asked only when the wait is far longer than any order takes (an error has happened, or the GPU is shared):
        // hipStreamQuery puts a marker into the queue, and the kernel behind a marker starts ~6 us late (measured: a query every
        // 0.3 ms of waiting cost every dense order of the headline sweep that gap).
        if ((it & 0x3fff) == 0) {
            const auto now = std::chrono::steady_clock::now();
            if (now >= next_query) {
                next_query = now + std::chrono::milliseconds(20);
                const hipError_t q = hipStreamQuery(group_stream(h, grp));
                if (q != hipSuccess && q != hipErrorNotReady) return fail(SOSRT_E_HIP, "order loop: %s", hipGetErrorString(q));
                if (q == hipSuccess && __atomic_load_n(&slot[1], __ATOMIC_ACQUIRE) != tag)
                    return fail(SOSRT_E_HIP, "order loop: the stream drained without publishing order tag %d", tag);
                if (now - t0 > std::chrono::seconds(120))
                    return fail(SOSRT_E_HIP, "order loop: no progress for 120 s");
            }
        }
        __builtin_ia32_pause();
    }
}

Conv make_conv(sosrt_handle* h, double tol) {
    Conv c;
    c.active = h->d_active; c.norders = h->d_norders; c.status = h->d_status;
    c.nactive = h->d_nactive; c.ratio = h->d_ratio; c.tol = tol; c.redo = h->d_redo;
    return c;
}

int check_ready(sosrt_handle* h, int B, bool need_phase) {
    if (int e = need_gpu(h)) return e;
    prof_break(h);
    if (!h->have_grid) return fail(SOSRT_E_STATE, "sosrt_set_grid has not been called");
    if (need_phase && !h->have_phase) return fail(SOSRT_E_STATE, "sosrt_set_phase has not been called");
    if (!h->have_cols) return fail(SOSRT_E_STATE, "sosrt_set_columns has not been called");
    if (B != h->B) return fail(SOSRT_E_INVALID, "B=%d does not match sosrt_set_columns (B=%d)", B, h->B);
    if (need_phase && h->geom == SOSRT_GEOM_THREE_ZONE && !h->have_aer)
        return fail(SOSRT_E_STATE, "three-zone geometry needs P_aer (sosrt_set_phase)");
    return 0;
}

ColScalars scalars_of(sosrt_handle* h) {
    ColScalars sc;
    const size_t mb = h->max_batch;
    sc.idx_up = h->d_idx_up; sc.idx_down = h->d_idx_down;
    sc.nz = h->d_nz; sc.zr0 = h->d_zr0; sc.zmix = h->d_zmix; sc.zwr = h->d_zwr; sc.zdtr = h->d_zdtr;
    sc.mu0 = h->d_scal + 0 * mb; sc.rho = h->d_scal + 1 * mb; sc.alb_atm = h->d_scal + 2 * mb;
    sc.alb_aer = h->d_scal + 3 * mb; sc.dtau_atm = h->d_scal + 4 * mb; sc.dtau_aer = h->d_scal + 5 * mb;
    sc.T = h->d_scal + 6 * mb;
    return sc;
}

// buffers of the float contraction (SOSRT_CONTRACT_F32)
int ensure_w32(sosrt_handle* h) {
    if (h->contraction != SOSRT_CONTRACT_F32) return 0;
    if (h->nslab > 0 && h->mix_groups == 0)
        return fail(SOSRT_E_INVALID, "the float contraction needs at most %d distinct slab coefficient pairs in a batch", sosrt_handle::kMaxMixGroups);
    const size_t need = (size_t)h->g.Dp * h->g.Wld * (size_t)(h->mix_groups > 0 ? h->mix_groups : 1);
    if (need > h->mix32_capacity) {
        if (h->d_Wmix32) hipFree(h->d_Wmix32);
        h->d_Wmix32 = nullptr; h->mix32_capacity = 0;
        if (int e = dalloc(&h->d_Wmix32, need)) return e;
        h->mix32_capacity = need;
        h->w32_dirty = true;
    }
    return 0;
}

bool use_sym(const sosrt_handle* h) { return h->contraction == SOSRT_CONTRACT_F64 && h->sym_ok; }

// combined slab matrices and, for the symmetric contraction, the folded copies of every matrix (on stream s)
int ensure_matrices(sosrt_handle* h, hipStream_t s) {
    const Grid& g = h->g;
    const size_t per = (size_t)g.Dp * g.Wld;
    const bool mixed = h->mix_groups > 0 && h->mix_dirty;
    if (mixed) {
        prof_break(h);
        launch_wmix(s, per, h->mix_groups, h->d_Wa, h->d_Wr, h->d_mixca, h->d_mixcr, h->d_Wmix);
        h->mix_dirty = false;
        h->symmix_dirty = true;
    }
    if (!use_sym(h)) return 0;
    if (h->sym_dirty) {
        prof_break(h);
        if (!h->d_Wa_s) { if (int e = dalloc(&h->d_Wa_s, per)) return e; }
        if (!h->d_Wr_s) { if (int e = dalloc(&h->d_Wr_s, per)) return e; }
        launch_symfold(s, 1, g.N, g.D, g.Dp, g.Wld, h->d_Wa, h->d_Wa_s);
        launch_symfold(s, 1, g.N, g.D, g.Dp, g.Wld, h->d_Wr, h->d_Wr_s);
        h->sym_dirty = false;
    }
    if (h->mix_groups > 0 && h->symmix_dirty) {
        prof_break(h);
        const size_t need = per * h->mix_groups;
        if (need > h->mixs_capacity) {
            // (the stream may still read the old buffer: hipFree synchronises)
            if (h->d_Wmix_s) hipFree(h->d_Wmix_s);
            h->d_Wmix_s = nullptr; h->mixs_capacity = 0;
            if (int e = dalloc(&h->d_Wmix_s, need)) return e;
            h->mixs_capacity = need;
        }
        launch_symfold(s, h->mix_groups, g.N, g.D, g.Dp, g.Wld, h->d_Wmix, h->d_Wmix_s);
        h->symmix_dirty = false;
    }
    return 0;
}

// Jn for every row of a column group (grp < 0: the whole batch) in one launch: plain rows against W_atm, slab rows
// against the combined matrix of their coefficient pair (or W_atm and W_aer in two passes)
void run_source(sosrt_handle* h, const double* In_1, double* Jn, const int* active, int tail_cols = 0, int pub_tag = 0,
                int grp = -1, bool all_live = false, bool regs_tile = false, int dense_live_cap = 0) {
    const int g0 = grp < 0 ? 0 : grp, g1 = grp < 0 ? h->ngroups : grp + 1;
    const int pg = grp < 0 ? 0 : grp;
    hipStream_t s = group_stream(h, pg);
    GemmArgs ga;
    ga.A = In_1; ga.Wa = h->d_Wa; ga.Wr = h->d_Wr; ga.ca = h->d_rca; ga.cr = h->d_rcr;
    if (h->nslab > 0) {
        ga.rows_main = h->d_mainrows + h->main_off[g0];
        ga.n_main = h->main_off[g1] - h->main_off[g0];
    } else {                                           // no slab rows: the identity list, offset by the group's first row
        ga.rows_main = nullptr;
        ga.n_main = (h->gb[g1] - h->gb[g0]) * h->L;
        ga.A = In_1 + (size_t)h->gb[g0] * h->L * h->D;
        Jn += (size_t)h->gb[g0] * h->L * h->D;
        ga.ca = h->d_rca + (size_t)h->gb[g0] * h->L;
        ga.cr = h->d_rcr + (size_t)h->gb[g0] * h->L;
        if (active) active += h->gb[g0];
    }
    ga.rows_slab = h->d_slabrows + h->slab_off[g0]; ga.n_slab = h->slab_off[g1] - h->slab_off[g0];
    ga.D = h->g.D; ga.Dp = h->g.Dp; ga.Wld = h->g.Wld; ga.L = h->L; ga.C = Jn; ga.active = active;
    // Two column groups: the large tilings of the contraction are capped at two workgroups per CU (unused LDS up to a
    // third of the CU's) so that a transport workgroup of the other group -- 53 KB with a two-slot ring -- runs beside
    // them: the MFMA-bound contraction of one group then overlaps the HBM-bound transport of the other
    if (grp >= 0 && h->ngroups > 1) ga.pad_lds = h->coresident_pad;
    if (pub_tag) {
        ga.nactive = h->d_nactive + pg; ga.need_small = h->d_nactive + sosrt_handle::kMaxGroups;
        ga.host_pub = h->h_pub + 8 * pg; ga.tag = pub_tag;
    }
    // The dense tiling skips the tiles whose columns have all converged -- two barriers and a dependent load per tile.  When the
    // host knows every column of the launch to be live (its count lags by one order: at most the columns that converged in the
    // last order are multiplied once more, and the transport ignores them) the check is dropped: 160 -> 157 us per 512-column launch.
    ga.check_tiles = all_live ? 0 : 1;
    if (ensure_matrices(h, s)) return;                 // (allocation failure: reported by the caller's hipGetLastError / next call)
    if (h->mix_groups > 0) {
        ga.Wmix = h->d_Wmix; ga.mix_group = h->d_mixgroup; ga.slab_tile_group = h->d_slabtilegroup + h->slab_off[g0] / 32;
    }
    if (use_sym(h)) {
        ga.sym = 1; ga.Ks = (h->g.N + GEMM_KC - 1) / GEMM_KC * GEMM_KC;
#ifdef SOSRT_DIAG
        // diagnostic builds (timing only; the results do not change: the extra chunks multiply zeros): SOSRT_GEMM_KS_MULT=2 doubles
        // the chunks per tile at the same prologue / epilogue, which separates the two (tile time = P + chunks * C); read once,
        // at sosrt_create
        if (h->diag_ks_mult > 1 && ga.Ks * h->diag_ks_mult <= h->g.Dp) ga.Ks *= h->diag_ks_mult;
#endif
        ga.Wa = h->d_Wa_s; ga.Wr = h->d_Wr_s;
        if (ga.Wmix) ga.Wmix = h->d_Wmix_s;
    }
    if (h->contraction == SOSRT_CONTRACT_F32) {
        // float operands, float accumulator: the dense tiling over the row lists for every order (tiles of converged
        // columns leave at once)
        if (h->w32_dirty) {
            prof_break(h);
            const size_t per = (size_t)h->g.Dp * h->g.Wld;
            launch_to_float(s, per, h->d_Wa, h->d_Wa32);
            if (h->mix_groups > 0) launch_to_float(s, per * h->mix_groups, h->d_Wmix, h->d_Wmix32);
            h->w32_dirty = false;
        }
        prof_begin(h, SOSRT_K_GEMM, pg);
        launch_gemm_f32(s, ga, h->d_Wa32, h->d_Wmix32);
        prof_end(h, SOSRT_K_GEMM, pg);
        return;
    }
    prof_begin(h, SOSRT_K_GEMM, pg);
    if (tail_cols > 0 && active && h->nslab >= 0) {
        ga.col0 = h->nslab > 0 ? h->gb[g0] : 0; ga.B = h->gb[g1] - h->gb[g0];
        ga.max_main = h->max_main; ga.max_slab = h->max_slab;
        ga.idx_up = h->nslab > 0 ? h->d_idx_up : nullptr; ga.idx_down = h->nslab > 0 ? h->d_idx_down : nullptr;
        ga.live_list = h->d_livelist + h->gb[g0]; ga.live_cap = tail_cols;
        launch_gemm_tail(s, ga, tail_cols, tail_cols <= h->gemm_small_cols, regs_tile);
    } else {
        if (dense_live_cap > 0 && active && h->nslab >= 0) {     // the dense tiling writes the transport's live list too
            ga.col0 = h->nslab > 0 ? h->gb[g0] : 0; ga.B = h->gb[g1] - h->gb[g0];
            ga.live_list = h->d_livelist + h->gb[g0]; ga.live_cap = dense_live_cap;
        }
        launch_gemm(s, ga);
    }
    prof_end(h, SOSRT_K_GEMM, pg);
}

// ---------------------------------------------------------------------------------------------
// Launch plan: which kernels run an order of a column group.  One function, host only -- it reads the handle's shape and
// knobs and touches no device -- so that the policy can be read in one place and tested without a GPU (sosrt_plan_launch).
// ---------------------------------------------------------------------------------------------
struct SolveShape {                      // what a solve fixes for all its orders (from the grid and the batch's zone tables)
    bool ring_like = false;              // the ring / chunk-parallel kernels take the batch (they hold its zone tables)
    bool fast = false;                   // a wave-independent kernel runs (else the general kernel)
    int nzcap = kRingZones;              // most zones of any column, at least three
    int ring_mode = 1;                   // 3: ring-class kernels, 1: the register-streaming kernel
};
SolveShape solve_shape(const sosrt_handle* h, int max_nz) {
    SolveShape sh;
    // The ring / chunk-parallel kernels take columns of any zone count (an instantiation that tests every boundary of the zone
    // table, chosen when the batch holds such a column); the register-streaming kernel knows three zones, so a batch with more
    // goes to the general kernel where that one would run (odd N, N > 256).
    sh.ring_like = h->transport_mode >= 2 && h->ring_ok && (max_nz <= kRingZones || transport_ring_fits(h->g, max_nz));
    sh.fast = h->transport_mode >= 1 && h->fast_ok && (max_nz <= kRingZones || sh.ring_like);
    sh.nzcap = max_nz > kRingZones ? max_nz : kRingZones;
    sh.ring_mode = (h->transport_mode >= 2 && h->ring_ok) ? 3 : 1;
    return sh;
}
struct OrderInputs {                     // what the plan of one order depends on besides the handle
    int nb = 0;                          // columns of the group
    int known = 0;                       // upper bound of its live columns (the host's count lags by one order)
    int surface = SOSRT_SURFACE_NONE;
    bool simple_zones = true;            // every column is (clear, slab, clear), or a single slab
    bool slabs_mixed = true;             // slab rows have their combined matrices (or there are none)
    bool need_small = false;             // some |mu| < 0.01 lane keeps its k_smallmu value
    bool saving = false;                 // the caller wants every order's field (I_saved)
    int orders_left = 1 << 30;           // order budget from this order on
    int cu_share = 0;                    // CUs an order-loop launch of this group may take (0: none)
};
struct LaunchPlan {
    int tail_cols = 0;                   // contraction over the live columns: capacity of the launch (0: dense tiling over the row lists)
    int live_cap = 0;                    // transport over the live list: its capacity (0: over all columns of the group)
    int gemm = SOSRT_PLAN_GEMM_DENSE;
    int transport = SOSRT_PLAN_TRANSPORT_GENERAL;
    int parts = 1;                       // chunk-parallel kernel: workgroups per column
    int repair = 0;                      // register-streaming kernel: the general kernel behind it for searches that leave wave 0
    int order_loop = 0;                  // this and every later order of the group in ONE order-loop launch
    int ol_parts = 0;                    // ... workgroups per column of its transport role
    int ol_grid = 0;                     // ... workgroups of the launch
};
LaunchPlan plan_order(const sosrt_handle* h, const SolveShape& sh, const OrderInputs& in) {
    LaunchPlan pl;
    const Grid& g = h->g;
    // contraction: the tilings over the live columns whenever some column has converged -- and for a small batch from the
    // start: their 32-row tiles put a few columns on more CUs than the dense tiling's 64-row tiles; same bits either way.
    // (the float contraction has the dense tiling only: no live list for the transport either)
    const bool live_tiling = h->contraction != SOSRT_CONTRACT_F32 && in.simple_zones && in.known <= h->gemm_tail_cols &&
                             (in.known <= h->gemm_tail_frac * in.nb || in.nb <= h->gemm_small_cols) &&
                             (in.known < in.nb || in.nb <= h->gemm_small_cols);
    pl.tail_cols = live_tiling ? in.known : 0;
    pl.gemm = !live_tiling ? SOSRT_PLAN_GEMM_DENSE
                           : (pl.tail_cols <= h->gemm_small_cols ? ((use_sym(h) && pl.tail_cols <= 32) ? SOSRT_PLAN_GEMM_LIVE32_DEEP : SOSRT_PLAN_GEMM_LIVE32)
                                                                 : SOSRT_PLAN_GEMM_LIVE64);
    // The last few columns: a tile's latency is the launch's, and the register-resident 16-row tile (jn_gemm_tile.hpp:
    // gemm_tile_lone) has half the staged tile's -- a lone column's launch 12.6 -> 10.0 us at N = 128, 4.7 of which an empty launch
    // takes (profiles/r04_gemm_regs_ab.txt).  Its workgroups are alone on their CUs and each fetches its own share of the matrix:
    // it wins while they make at most about a round and a half (13 columns at L = 200, N = 128; 6 at N = 256), measured break-even
    // at 16 / 8 -- and loses where a lone column is already more than that: L = 800, N = 501, 133 -> 144 us per order (profiles/r04_gemm_regs_ab.txt).  (Its tile's rows of In_1 must fit the LDS, and N rounded up to the k-chunk must be whole register blocks of 64.)
    {
        const int nct = (g.D + GEMM_BN - 1) / GEMM_BN;
        const int auto_cap = (3 * h->cu_count / 2) / (((g.L + 15) / 16 + 1) * nct);
        const int cap = h->gemm_regs_cols >= 0 ? h->gemm_regs_cols : auto_cap;       // (0 at the reference's shipped size: 408 workgroups for a lone column)
        if (live_tiling && use_sym(h) && pl.tail_cols <= cap && 16 * (g.D + 2) * 8 <= 150 * 1024 &&
            ((g.N + GEMM_KC - 1) / GEMM_KC * GEMM_KC) % 64 == 0) pl.gemm = SOSRT_PLAN_GEMM_LIVE16_REGS;
    }
    // The transport takes its columns from the live list whenever some column has converged: the live-column tilings write the
    // list, and so does the dense tiling (one more workgroup) -- the ring-class kernels then run over the live columns, dealt to
    // the CUs one by one, instead of over a batch whose live columns sit where they were put.  (Not the float contraction.)
    pl.live_cap = live_tiling ? pl.tail_cols : ((h->contraction != SOSRT_CONTRACT_F32 && h->dense_live_list && in.known < in.nb) ? in.known : 0);
    // transport
    {
        const int cols_now = pl.live_cap > 0 ? pl.live_cap : in.nb;
        // chunk-parallel kernel: a column on ceil(N / 64) CUs (two at N = 128, four at N = 256) while that many workgroups per
        // live column fit the device at once (the reflection must stay inside a part)
        // (where the shape has no ring kernel -- odd N, N > 256: the split form's WIDE instantiation -- the alternative is the
        // register-streaming kernel, one workgroup per column and 650 us per order at the shipped size against 133 for a round of
        // split workgroups: up to four rounds of them are the faster way)
        const int split_cap = sh.ring_mode == 3 ? h->cu_count : 4 * h->cu_count;
        const bool can_split = h->scan_split && h->scan_split_ok && transport_scan_parts(g) * cols_now <= split_cap &&
                               (in.surface == SOSRT_SURFACE_SPECULAR || in.surface == SOSRT_SURFACE_NONE);
        const bool want_scan = h->transport_mode == 4 || (h->transport_mode == 3 && cols_now <= h->scan_cols);
        // (the split form also takes the shapes no wave-independent kernel does -- the rewritten directions straddle two waves of a
        // half row, e.g. N = 70, 129, 257: sh.fast is false -- as long as the attenuation tables are built)
        const bool split = can_split && transport_scan_fits(g, sh.nzcap, true) && (sh.fast || (h->use_etab && h->transport_mode >= 3));
        const bool scan = want_scan && ((sh.fast && sh.ring_mode == 3 && h->scan_ok && transport_scan_fits(g, sh.nzcap, false)) || split);
        pl.transport = scan ? SOSRT_PLAN_TRANSPORT_SCAN
                            : (!sh.fast ? SOSRT_PLAN_TRANSPORT_GENERAL : (sh.ring_mode == 3 ? SOSRT_PLAN_TRANSPORT_RING : SOSRT_PLAN_TRANSPORT_FAST));
        if (scan && split) pl.parts = transport_scan_parts(g);
        pl.repair = (h->N - 3 > 61 && pl.transport == SOSRT_PLAN_TRANSPORT_FAST) ? 1 : 0;
    }
    // order-loop kernel: the remaining orders in one launch once the live columns' transport workgroups are a small share of the
    // CUs it may take (the other workgroups contract).  It holds the chunk-parallel transport (three zones, no kept k_smallmu
    // lane) and the symmetric contraction's live-column tiles; the default transport policy only (a forced kernel stays forced).
    if (h->order_loop && in.cu_share > 0 && h->transport_mode == 3 && sh.fast && sh.ring_mode == 3 && sh.nzcap <= kRingZones &&
        use_sym(h) && in.simple_zones && in.slabs_mixed && !in.saving && !in.need_small && in.orders_left >= 1 &&
        in.known <= kOrderLoopMaxCols) {
        Grid gt = g;
        gt.nsmall = 0;
        const bool split_ok = h->scan_split && (in.surface == SOSRT_SURFACE_SPECULAR || in.surface == SOSRT_SURFACE_NONE) && order_loop_ok(gt, true);
        const int sp = order_loop_parts(gt, true);
        if (split_ok && sp * in.known <= h->ol_frac * in.cu_share) {
            pl.order_loop = 1; pl.ol_parts = sp;
        } else if (order_loop_ok(gt, false) && in.known <= h->ol_frac * in.cu_share) {
            pl.order_loop = 1; pl.ol_parts = 1;
        }
        if (pl.order_loop) pl.ol_grid = in.cu_share;
    }
    return pl;
}

// CUs of a device that order-loop launches of this process hold (their workgroups wait for each other, so every one of them must
// be resident: the launches of all handles together never ask for more workgroups than the device has CUs)
std::mutex g_ol_mutex;
int g_ol_held[64];
int ol_acquire(int device, int total, int want, int least) {
    if (device < 0 || device >= 64) return 0;
    std::lock_guard<std::mutex> lk(g_ol_mutex);
    int got = total - g_ol_held[device];
    if (got > want) got = want;
    if (got < least || got <= 0) return 0;
    g_ol_held[device] += got;
    return got;
}
void ol_release(int device, int n) {
    if (device < 0 || device >= 64 || n <= 0) return;
    std::lock_guard<std::mutex> lk(g_ol_mutex);
    g_ol_held[device] -= n;
}

}  // namespace

extern "C" {

const char* sosrt_last_error(void) { return g_err.c_str(); }
int sosrt_version(void) { return SOSRT_VERSION; }

int sosrt_plan_fix_count(double tau_ref, int N) { return fix_count(tau_ref, N); }

int sosrt_create(int device, int L, int N, int max_batch, int max_orders, sosrt_t** out) {
    if (!out) return fail(SOSRT_E_INVALID, "out is null");
    *out = nullptr;
    if (L < 2) return fail(SOSRT_E_INVALID, "nb_layers must be >= 2 (got %d)", L);
    if (N < 4) return fail(SOSRT_E_INVALID, "nb_angles must be >= 4 (got %d)", N);
    if (N > 1024) return fail(SOSRT_E_INVALID, "nb_angles must be <= 1024 (got %d)", N);
    if (max_batch < 1 || max_orders < 1) return fail(SOSRT_E_INVALID, "max_batch and max_orders must be >= 1");
    {   // the first-order kernel keeps three values per layer of its column in LDS (the largest per-layer need of any kernel)
        const int nt = (N + 63) / 64 * 64;
        const size_t need = (size_t)(3 * (size_t)L + nt + nt / 64 + 4) * sizeof(double);
        if (need > 64 * 1024)
            return fail(SOSRT_E_INVALID, "nb_layers = %d does not fit: at most %d layers at nb_angles = %d", L,
                        (int)((64 * 1024 / sizeof(double) - nt - nt / 64 - 4) / 3), N);
    }
    sosrt_handle* h = new (std::nothrow) sosrt_handle();
    if (!h) return fail(SOSRT_E_NOMEM, "out of host memory");
    h->device = device; h->L = L; h->N = N; h->D = 2 * N; h->max_batch = max_batch; h->max_orders = max_orders;
    h->order_budget = max_orders;
    h->saved_slots = max_orders;
    h->gpu = device >= 0;
    h->cu_count = 256;                       // (a host-only handle plans for an MI355X; a device handle asks the device below)
    if (const char* ev = getenv("SOSRT_ETAB")) h->use_etab = atoi(ev);
    if (const char* ev = getenv("SOSRT_CONTRACT"))            // "full": the D x D product whatever the symmetry of the matrices
        if (strcmp(ev, "full") == 0) h->contraction = SOSRT_CONTRACT_F64_FULL;
    if (const char* ev = getenv("SOSRT_TRANSPORT"))
        h->transport_mode = strcmp(ev, "general") == 0 ? 0 : (strcmp(ev, "ring") == 0 ? 2 : (strcmp(ev, "scan") == 0 ? 4 : (strcmp(ev, "auto") == 0 ? 3 : 1)));
    if (const char* ev = getenv("SOSRT_SCAN_COLS")) h->scan_cols = atoi(ev);
    if (const char* ev = getenv("SOSRT_SCAN_SPLIT")) h->scan_split = atoi(ev);
    if (const char* ev = getenv("SOSRT_GEMM_TAIL")) h->gemm_tail_cols = atoi(ev);
    if (const char* ev = getenv("SOSRT_GEMM_TAIL_FRAC")) h->gemm_tail_frac = atof(ev);
    if (const char* ev = getenv("SOSRT_GEMM_SMALL")) h->gemm_small_cols = atoi(ev);
    if (const char* ev = getenv("SOSRT_DENSE_LIVE_LIST")) h->dense_live_list = atoi(ev) != 0;
    if (const char* ev = getenv("SOSRT_GEMM_REGS")) h->gemm_regs_cols = atoi(ev);           // (A/B: 0 = the staged tilings for every live count)
    if (const char* ev = getenv("SOSRT_GROUPS")) h->want_groups = atoi(ev) >= 2 ? 2 : (atoi(ev) == 1 ? 1 : 0);      // column groups of the order loop (0: auto)
    if (const char* ev = getenv("SOSRT_SPLIT_MIN")) h->split_min = atoi(ev);                  // smallest batch that is split
#ifdef SOSRT_DIAG   // measurement knobs of DESIGN section 5 (items 1, 5, 8): diagnostic builds only (-DSOSRT_DIAG), never in the product library
    if (const char* ev = getenv("SOSRT_STAGGER")) h->stagger = atof(ev);
    if (const char* ev = getenv("SOSRT_GROUP_SPLIT")) h->split_at = atoi(ev);
    if (const char* ev = getenv("SOSRT_GROUP_PRIO")) h->prio2 = atoi(ev);
    if (const char* ev = getenv("SOSRT_GEMM_PAD_LDS")) h->coresident_pad = atoi(ev);
    if (const char* ev = getenv("SOSRT_GEMM_KS_MULT")) h->diag_ks_mult = atoi(ev);
#endif
    if (const char* ev = getenv("SOSRT_GROUP_RING_SLOTS")) h->coresident_slots = atoi(ev);
    if (const char* ev = getenv("SOSRT_ORDER_LOOP")) h->order_loop = atoi(ev) != 0;           // (A/B: 0 = every order as two launches)
    if (const char* ev = getenv("SOSRT_RING_SLOTS")) g_ring_slots = atoi(ev);
#ifdef SOSRT_RING_DEBUG   // diagnostic builds only: the switches make the ring kernel skip work, its results are wrong
    if (const char* ev = getenv("SOSRT_RING_DEBUG")) g_ring_debug = atoi(ev);
#endif
    Grid& g = h->g;
    g.L = L; g.N = N; g.D = 2 * N;
    g.Dp = (g.D + GEMM_KC - 1) / GEMM_KC * GEMM_KC;
    g.Wld = (g.D + GEMM_BN - 1) / GEMM_BN * GEMM_BN;
    if (h->gpu) {
        int e = 0;
        auto body = [&]() -> int {
            HIPCHK(hipSetDevice(device));
            // the handle's own stream is a BLOCKING stream: it orders against the legacy default stream (handle NULL), which is
            // where a caller without an explicit stream -- torch's default stream is that one -- fills and reads its buffers
            HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamDefault));
            h->stream = h->own_stream;
            // (the second group's stream is created when a batch first takes two groups: HIP maps streams onto a few hardware
            // queues, and a handle that never splits should not take one from the caller's other streams)
            HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
            const size_t mb = max_batch, fe = field_elems(h);
            if ((e = dalloc(&h->d_mu, g.D))) return e;
            if ((e = dalloc(&h->d_Wa, (size_t)g.Dp * g.Wld))) return e;
            if ((e = dalloc(&h->d_Wr, (size_t)g.Dp * g.Wld))) return e;
            if ((e = dalloc(&h->d_wfdn, N))) return e;
            if ((e = dalloc(&h->d_wfup, N))) return e;
            if ((e = dalloc(&h->d_w, g.D))) return e;
            if ((e = dalloc(&h->d_phi, 2 * kNPhi))) return e;
            if ((e = dalloc(&h->d_z, L))) return e;
            if ((e = dalloc(&h->d_fix, 4))) return e;
            if ((e = dalloc(&h->d_small, N))) return e;
            if ((e = dalloc(&h->d_idx_up, mb))) return e;
            if ((e = dalloc(&h->d_idx_down, mb))) return e;
            if ((e = dalloc(&h->d_nz, mb))) return e;
            if ((e = dalloc(&h->d_zr0, mb * kMaxZones))) return e;
            if ((e = dalloc(&h->d_zmix, mb * kMaxZones))) return e;
            if ((e = dalloc(&h->d_zwr, mb * kMaxZones))) return e;
            if ((e = dalloc(&h->d_zdtr, mb * kMaxZones))) return e;
            if ((e = dalloc(&h->d_scal, 7 * mb))) return e;
            if ((e = dalloc(&h->d_desc, mb))) return e;
            if ((e = dalloc(&h->d_rca, mb * L))) return e;
            if ((e = dalloc(&h->d_rcr, mb * L))) return e;
            if ((e = dalloc(&h->d_slabrows, mb * L + 64 * (size_t)sosrt_handle::kMaxMixGroups * sosrt_handle::kMaxGroups))) return e;   // + padding
            if ((e = dalloc(&h->d_slabtilegroup, mb * L / 32 + 2 * sosrt_handle::kMaxMixGroups * sosrt_handle::kMaxGroups + 2))) return e;
            if ((e = dalloc(&h->d_mainrows, mb * L))) return e;
            if ((e = dalloc(&h->d_tau, mb * L))) return e;
            if ((e = dalloc(&h->d_P0a, mb * g.D))) return e;
            if ((e = dalloc(&h->d_P0r, mb * g.D))) return e;
            if ((e = dalloc(&h->d_Jn, fe))) return e;
            if ((e = dalloc(&h->d_InA, fe))) return e;
            if ((e = dalloc(&h->d_InB, fe))) return e;
            if ((e = dalloc(&h->d_I, fe))) return e;
            if ((e = dalloc(&h->d_E, fe))) return e;
            if ((e = dalloc(&h->d_active, mb))) return e;
            if ((e = dalloc(&h->d_norders, mb))) return e;
            if ((e = dalloc(&h->d_status, mb))) return e;
            if ((e = dalloc(&h->d_nactive_sets, 2 * (sosrt_handle::kMaxGroups + 1)))) return e;   // live columns per group; any column needs k_smallmu
            HIPCHK(hipMemset(h->d_nactive_sets, 0, 2 * (sosrt_handle::kMaxGroups + 1) * sizeof(int)));
            h->d_nactive = h->d_nactive_sets;
            if ((e = dalloc(&h->d_redo, mb))) return e;
            if ((e = dalloc(&h->d_erep, mb))) return e;
            if ((e = dalloc(&h->d_mixgroup, mb))) return e;
            if ((e = dalloc(&h->d_livelist, mb))) return e;
            if ((e = dalloc(&h->d_mixca, sosrt_handle::kMaxMixGroups))) return e;
            if ((e = dalloc(&h->d_mixcr, sosrt_handle::kMaxMixGroups))) return e;
            if ((e = dalloc(&h->d_tauhash, mb))) return e;
            if ((e = dalloc(&h->d_ratio, mb))) return e;
            if ((e = dalloc(&h->d_scan_scratch, mb * transport_scan_scratch_doubles()))) return e;
            if ((e = dalloc(&h->d_scan_sync, 2 * mb))) return e;
            if ((e = dalloc(&h->d_olsync, sosrt_handle::kMaxGroups * order_loop_sync_ints(kOrderLoopMaxCols)))) return e;
            HIPCHK(hipHostMalloc((void**)&h->h_oldone, 2 * sosrt_handle::kMaxGroups * sizeof(int), hipHostMallocCoherent));
            memset(h->h_oldone, 0, 2 * sosrt_handle::kMaxGroups * sizeof(int));
            HIPCHK(hipMemset(h->d_scan_sync, 0, 2 * mb * sizeof(int)));
            HIPCHK(hipDeviceGetAttribute(&h->cu_count, hipDeviceAttributeMultiprocessorCount, device));
            // + 2 ints at the end: {needs k_smallmu, tag} published at the start of a solve
            HIPCHK(hipHostMalloc((void**)&h->h_pub, (8 * sosrt_handle::kMaxGroups + 2) * sizeof(int), hipHostMallocCoherent));
            memset(h->h_pub, 0, (8 * sosrt_handle::kMaxGroups + 2) * sizeof(int));

            HIPCHK(hipMemset(h->d_Wa, 0, (size_t)g.Dp * g.Wld * sizeof(double)));
            HIPCHK(hipMemset(h->d_Wr, 0, (size_t)g.Dp * g.Wld * sizeof(double)));
            HIPCHK(hipMemset(h->d_status, 0, mb * sizeof(int)));
            HIPCHK(hipMemset(h->d_redo, 0, mb * sizeof(int)));
            HIPCHK(hipMemset(h->d_tau, 0, mb * L * sizeof(double)));
            return 0;
        };
        if ((e = body())) { sosrt_destroy(h); return e; }
        g.mu = h->d_mu; g.Wa = h->d_Wa; g.Wr = h->d_Wr; g.fix = h->d_fix; g.small_lanes = h->d_small;
        g.wflux_dn = h->d_wfdn; g.wflux_up = h->d_wfup; g.nsmall = 0;
    }
    *out = h;
    return 0;
}

int sosrt_destroy(sosrt_t* h) {
    if (!h) return 0;
    if (h->gpu) {
        hipSetDevice(h->device);
        if (h->comm) { rccl().CommDestroy(h->comm); h->comm = nullptr; }
        if (h->own_stream) hipStreamSynchronize(h->own_stream);
        void* ptrs[] = {h->d_mu, h->d_Wa, h->d_Wr, h->d_wfdn, h->d_wfup, h->d_fix, h->d_small, h->d_idx_up,
                        h->d_idx_down, h->d_scal, h->d_desc, h->d_rca, h->d_rcr, h->d_slabrows, h->d_mainrows, h->d_tau, h->d_P0a,
                        h->d_P0r, h->d_Jn, h->d_InA, h->d_InB, h->d_I, h->d_E, h->d_active, h->d_norders, h->d_status,
                        h->d_nactive_sets, h->d_ratio, h->d_redo, h->d_erep, h->d_tauhash, h->d_Wmix, h->d_mixca, h->d_mixcr,
                        h->d_mixgroup, h->d_Wa_s, h->d_Wr_s, h->d_Wmix_s, h->d_scan_scratch, h->d_scan_sync, h->d_w, h->d_phi, h->d_z, h->d_tab, h->d_slabtilegroup, h->d_livelist, h->d_Wa32, h->d_Wmix32, h->d_nz, h->d_zr0, h->d_zmix, h->d_zwr, h->d_zdtr, h->d_olsync, h->d_ollog};
        for (void* p : ptrs)
            if (p) hipFree(p);
        if (h->h_pub) hipHostFree(h->h_pub);
        if (h->h_oldone) hipHostFree(h->h_oldone);

        for (auto& p : h->prof)
            for (auto& e : p.ev) hipEventDestroy(e);
        if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
        if (h->ev_fork) hipEventDestroy(h->ev_fork);
        if (h->ev_join) hipEventDestroy(h->ev_join);
        if (h->own_stream) hipStreamDestroy(h->own_stream);
    }
    delete h;
    return 0;
}

int sosrt_set_stream(sosrt_t* h, void* s) {
    if (int e = need_gpu(h)) return e;
    h->stream = (hipStream_t)s;              // NULL is the legacy default stream, as everywhere in HIP
    return 0;
}

int sosrt_use_own_stream(sosrt_t* h) {
    if (int e = need_gpu(h)) return e;
    h->stream = h->own_stream;
    return 0;
}

int sosrt_set_saved_orders(sosrt_t* h, int slots) {
    if (!h) return fail(SOSRT_E_INVALID, "null handle");
    if (slots < 1 || slots > h->max_orders) return fail(SOSRT_E_INVALID, "slots must be in 1..max_orders=%d (got %d)", h->max_orders, slots);
    h->saved_slots = slots;
    return 0;
}

int sosrt_set_order_budget(sosrt_t* h, int max_orders) {
    if (!h) return fail(SOSRT_E_INVALID, "null handle");
    if (max_orders < 1 || max_orders > h->max_orders)
        return fail(SOSRT_E_INVALID, "the order budget must be in 1..max_orders=%d of sosrt_create (got %d)", h->max_orders, max_orders);
    h->order_budget = max_orders;
    return 0;
}

int sosrt_set_first_order(sosrt_t* h, int mode) {
    if (int e = need_gpu(h)) return e;
    if (mode != SOSRT_FIRST_ORDER_CODED && mode != SOSRT_FIRST_ORDER_README) return fail(SOSRT_E_INVALID, "unknown first-order mode %d", mode);
    if (mode == SOSRT_FIRST_ORDER_README && h->geom != SOSRT_GEOM_THREE_ZONE)
        return fail(SOSRT_E_INVALID, "the README's Lambertian first order needs the three-zone geometry (it has a surface)");
    h->first_order_mode = mode;
    return 0;
}

int sosrt_set_contraction(sosrt_t* h, int mode) {
    if (int e = need_gpu(h)) return e;
    if (mode != SOSRT_CONTRACT_F64 && mode != SOSRT_CONTRACT_F32 && mode != SOSRT_CONTRACT_F64_FULL) return fail(SOSRT_E_INVALID, "unknown contraction mode %d", mode);
    if (mode == SOSRT_CONTRACT_F32) {
        HIPCHK(hipSetDevice(h->device));
        const size_t per = (size_t)h->g.Dp * h->g.Wld;
        if (!h->d_Wa32) { if (int e = dalloc(&h->d_Wa32, per)) return e; }
        h->w32_dirty = true;
    }
    h->contraction = mode;
    return 0;
}

int sosrt_set_order_loop(sosrt_t* h, int mode) {
    if (!h) return fail(SOSRT_E_INVALID, "null handle");
    if (mode < 0 || mode > 2) return fail(SOSRT_E_INVALID, "order-loop mode must be 0, 1 or 2 (got %d)", mode);
    h->order_loop = mode;
    return 0;
}

int sosrt_order_loop_stats(sosrt_t* h, int* launches, int* refused, long long* column_orders) {
    if (!h) return fail(SOSRT_E_INVALID, "null handle");
    if (launches) *launches = h->ol_launches;
    if (refused) *refused = h->ol_refused;
    if (column_orders) {
        *column_orders = 0;
        if (h->gpu && h->ol_launches > 0) {             // every launch of the last solve left its count in its group's words
            HIPCHK(hipSetDevice(h->device));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (h->stream2) HIPCHK(hipStreamSynchronize(h->stream2));
            for (int k = 0; k < sosrt_handle::kMaxGroups; ++k) {
                if (!h->ol_group_used[k]) continue;
                int v = 0;
                HIPCHK(hipMemcpy(&v, h->d_olsync + (size_t)k * order_loop_sync_ints(kOrderLoopMaxCols) + kOlOrders, sizeof v, hipMemcpyDeviceToHost));
                *column_orders += v;
            }
        }
    }
    return 0;
}

int sosrt_plan_launch(sosrt_t* h, int batch, int live, int surface, int zones, int cus, int* out) {
    if (!h || !out) return fail(SOSRT_E_INVALID, "null argument");
    if (!h->have_grid) return fail(SOSRT_E_STATE, "sosrt_set_grid has not been called");
    if (batch < 1 || live < 0 || live > batch) return fail(SOSRT_E_INVALID, "need 0 <= live <= batch, batch >= 1");
    if (zones < 1 || zones > kMaxZones) return fail(SOSRT_E_INVALID, "zones must be in 1..%d", kMaxZones);
    // as sosrt_set_columns and the order loop of sosrt_solve_dev see a batch of `batch` (clear, slab, clear)-like columns with up to
    // `zones` zones: the column groups, then the plan of an order of the first group with `live` columns of it live
    int want = h->want_groups;
    if (want == 0) want = batch > h->split_min ? 2 : 1;
    const int ng = (want >= 2 && batch >= h->split_min && batch >= 2) ? 2 : 1;
    const int nb = ng == 2 ? batch / 2 : batch;
    const int saved_cus = h->cu_count;
    if (cus > 0) h->cu_count = cus;
    const SolveShape sh = solve_shape(h, zones);
    OrderInputs oi;
    oi.nb = nb; oi.known = live < nb ? live : nb; oi.surface = surface;
    oi.simple_zones = zones == 3 || zones == 1;
    oi.cu_share = h->cu_count / ng;
    const bool saved_sym = h->sym_ok;
    if (!h->have_phase) h->sym_ok = true;            // (no matrices yet: plan for flip-symmetric ones, what every phase function of the scattering angle gives)
    const LaunchPlan pl = plan_order(h, sh, oi);
    h->sym_ok = saved_sym;
    h->cu_count = saved_cus;
    out[0] = ng; out[1] = pl.gemm; out[2] = pl.tail_cols; out[3] = pl.transport; out[4] = pl.parts; out[5] = pl.repair;
    out[6] = pl.order_loop; out[7] = pl.ol_parts; out[8] = pl.ol_grid;
    return 0;
}

int sosrt_phase_asymmetry(sosrt_t* h, double* asymmetry, int* uses_symmetry) {
    if (!h) return fail(SOSRT_E_INVALID, "null handle");
    if (!h->have_phase) return fail(SOSRT_E_STATE, "sosrt_set_phase has not been called");
    if (asymmetry) *asymmetry = h->asymmetry;
    if (uses_symmetry) *uses_symmetry = use_sym(h) ? 1 : 0;
    return 0;
}

int sosrt_synchronize(sosrt_t* h) {
    if (int e = need_gpu(h)) return e;
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int sosrt_set_grid(sosrt_t* h, const double* mu) {
    if (!h || !mu) return fail(SOSRT_E_INVALID, "null argument");
    for (int k = 0; k < h->D; ++k)
        if (!std::isfinite(mu[k])) return fail(SOSRT_E_INVALID, "mu[%d] is not finite", k);
    for (int k = 0; k < h->N; ++k) {
        if (!(mu[k] <= 0)) return fail(SOSRT_E_INVALID, "mu[%d]=%g: the first nb_angles directions must be <= 0", k, mu[k]);
        if (!(mu[h->N + k] >= 0)) return fail(SOSRT_E_INVALID, "mu[%d]=%g: the last nb_angles directions must be >= 0", h->N + k, mu[h->N + k]);
    }
    h->plan.set_grid(h->N, mu);
    for (int b = 0; b < 4; ++b)
        if (h->plan.fix[b].idx > kFixMaxIdx) return fail(SOSRT_E_INVALID, "nb_angles too large for the extrapolation tables");
    h->have_grid = true;
    h->have_phase = false;
    h->fast_ok = transport_fast_ok(h->plan);
    if (h->gpu) {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipMemcpy(h->d_mu, mu, h->D * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_wfdn, h->plan.wflux_dn.data(), h->N * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_wfup, h->plan.wflux_up.data(), h->N * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_w, h->plan.w.data(), h->D * sizeof(double), hipMemcpyHostToDevice));
        {   // phi = np.linspace(0, pi, 25) (phase:81-82): cos(phi0 - phi) and the trapezoid weights of np.trapz(., phi)
            double phi[kNPhi], tab[2 * kNPhi];
            const double step = 3.141592653589793 / (kNPhi - 1);
            for (int q = 0; q < kNPhi; ++q) phi[q] = q * step;
            phi[kNPhi - 1] = 3.141592653589793;
            for (int q = 0; q < kNPhi; ++q) {
                tab[q] = std::cos(0 - phi[q]);
                tab[kNPhi + q] = ((q > 0 ? phi[q] - phi[q - 1] : 0.0) + (q + 1 < kNPhi ? phi[q + 1] - phi[q] : 0.0)) / 2;
            }
            HIPCHK(hipMemcpy(h->d_phi, tab, sizeof tab, hipMemcpyHostToDevice));
        }
        std::vector<FixTab> ft(4);
        for (int b = 0; b < 4; ++b) {
            const FixTable& t = h->plan.fix[b];
            memset(&ft[b], 0, sizeof(FixTab));
            ft[b].idx = t.idx; ft[b].s0 = t.s0; ft[b].ns = t.ns;
            for (size_t i = 0; i < t.C.size(); ++i) ft[b].C[i] = t.C[i];
        }
        HIPCHK(hipMemcpy(h->d_fix, ft.data(), 4 * sizeof(FixTab), hipMemcpyHostToDevice));
        h->g.nsmall = (int)h->plan.small_lanes.size();
        if (h->g.nsmall)
            HIPCHK(hipMemcpy(h->d_small, h->plan.small_lanes.data(), h->g.nsmall * sizeof(int), hipMemcpyHostToDevice));
    }
    // which kernels take this shape (pure functions of the grid: a host-only handle answers sosrt_plan_launch with them)
    h->g.nsmall = (int)h->plan.small_lanes.size();
    h->ring_ok = h->fast_ok && transport_ring_ok(h->g);
    h->scan_ok = h->ring_ok && transport_scan_ok(h->g);
    // (N in (128, 256]: the chunk-parallel kernel has this form only; it does not need the ring kernel's shape -- odd N, N up to
    // 512 and L up to 1024 take its WIDE instantiation: the reference's shipped N = 501, L = 800)
    // (nor the condition of the wave-independent kernels that the rewritten mu -> 0- directions and their sources sit in the last
    // wave of a half row: part 0 of the split form holds the downward directions N-64 .. N-1 in ONE wave whatever N is)
    h->scan_split_ok = transport_scan_split_ok(h->g);
    return 0;
}

int sosrt_set_phase(sosrt_t* h, const double* P_atm, const double* P_aer) {
    if (!h || !P_atm) return fail(SOSRT_E_INVALID, "null argument");
    if (!h->have_grid) return fail(SOSRT_E_STATE, "sosrt_set_grid has not been called");
    h->plan.fold(P_atm, h->Wa_h);
    h->have_aer = P_aer != nullptr;
    if (P_aer) h->plan.fold(P_aer, h->Wr_h);
    else h->Wr_h.clear();
    h->have_phase = true;
    h->mix_dirty = true;
    h->w32_dirty = true;
    h->sym_dirty = true; h->symmix_dirty = true;
    {   // flip symmetry of the folded matrices (see sosrt.h, sosrt_set_contraction)
        const int D = h->D;
        h->asymmetry = 0;
        for (const std::vector<double>* W : {&h->Wa_h, &h->Wr_h}) {
            if (W->empty()) continue;
            double wmax = 0, amax = 0;
            for (int k = 0; k < D; ++k)
                for (int m = 0; m < D; ++m) {
                    const double x = (*W)[(size_t)k * D + m], y = (*W)[(size_t)(D - 1 - k) * D + (D - 1 - m)];
                    const double ax = std::fabs(x), d = std::fabs(x - y);
                    if (!(ax <= wmax)) wmax = ax;          // a NaN ends up here and switches the symmetric form off
                    if (!(d <= amax)) amax = d;
                }
            const double r = wmax > 0 ? amax / wmax : 0.0;
            if (!(r <= h->asymmetry)) h->asymmetry = r;
        }
        // (a compile-time constant: nothing in the environment can put the symmetric form on a matrix without the symmetry)
        h->sym_ok = h->asymmetry <= SOSRT_SYMMETRY_TOL;
    }
    if (h->gpu) {
        HIPCHK(hipSetDevice(h->device));
        const Grid& g = h->g;
        HIPCHK(hipMemcpy2D(h->d_Wa, g.Wld * sizeof(double), h->Wa_h.data(), g.D * sizeof(double), g.D * sizeof(double),
                           g.D, hipMemcpyHostToDevice));
        if (P_aer)
            HIPCHK(hipMemcpy2D(h->d_Wr, g.Wld * sizeof(double), h->Wr_h.data(), g.D * sizeof(double),
                               g.D * sizeof(double), g.D, hipMemcpyHostToDevice));
    }
    return 0;
}

// Common part of sosrt_set_columns / sosrt_set_columns_zones: zone tables [B][kMaxZones] (host), per-column scalars.
static int set_columns_impl(sosrt_handle* h, int B, int geometry, int surface, const std::vector<int>& nz,
                            const std::vector<int>& zr0, const std::vector<int>& zmix, const std::vector<double>& zwr,
                            const std::vector<double>& zdtr, const double* mu0, const double* grd_alb, const double* alb_atm,
                            const double* dtau_atm, const double* tauStar_tot) {
    const size_t mb = h->max_batch;
    const int L = h->L;
    h->resident = false;                     // the descriptors of the resident field's columns are about to change
    std::vector<double> sc(7 * mb, 0.0);
    std::vector<int> slab, plain, iup(B, 0), idn(B, 0);
    auto zone_end = [&](int b, int z) { return z + 1 < nz[b] ? zr0[b * kMaxZones + z + 1] - 1 : L - 1; };
    // column groups of the order loop: two contiguous halves for a large batch
    // (auto: two groups for a batch of more than SPLIT_MIN columns (48; 256 in round 3).  Round 3, EVA / wildfire sweeps with one and two groups
    // alternating on one box: 288 x (200, 128) 3.65 -> 3.33 ms, 320 x 3.89 -> 3.50, 512 x 5.03 -> 4.75, 1024 x 9.2 -> 8.15,
    // 2048 x 16.6 -> 15.5, 4096 x 32.4 -> 30.3; 512 x (200, 64) 3.02 -> 2.94, 1024 x (200, 64) 5.10 -> 4.69; 512 x (200, 256)
    // 13.1 -> 12.65, 512 x (400, 256) 13.5 -> 12.8, 4096 x (400, 256) 106 -> 103.7; 256 x (200, 128) unchanged.  The gain is the
    // MFMA-bound contraction of one half running beside the HBM-bound transport of the other; the HBM bytes of a solve are the same
    // either way: 17.6 vs 18.0 GB by PMC)
    int want = h->want_groups;
    if (want == 0) want = B > h->split_min ? 2 : 1;
    h->ngroups = (want >= 2 && B >= h->split_min && B >= 2) ? 2 : 1;
    if (h->ngroups > 1 && !h->stream2) {
        HIPCHK(hipSetDevice(h->device));
        if (h->prio2) {
            int least = 0, greatest = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, h->prio2 > 0 ? greatest : least));
        } else {
            HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        }
    }
    h->gb[0] = 0; h->gb[1] = h->ngroups == 2 ? ((h->split_at > 0 && h->split_at < B) ? h->split_at : B / 2) : B; h->gb[2] = B;
    for (int k = 0; k <= sosrt_handle::kMaxGroups; ++k) { h->main_off[k] = 0; h->slab_off[k] = 0; }
    h->max_nz = 1;
    h->simple_zones = true;                  // every column is (clear, slab, clear): the live-column tilings apply
    if (geometry == SOSRT_GEOM_THREE_ZONE) {
        for (int b = 0; b < B; ++b) {
            h->max_nz = nz[b] > h->max_nz ? nz[b] : h->max_nz;
            const int* m = &zmix[b * kMaxZones];
            if (!(nz[b] == 3 && m[0] == 0 && m[1] == 1 && m[2] == 0)) h->simple_zones = false;
            if (nz[b] == 3) { iup[b] = zr0[b * kMaxZones + 1]; idn[b] = zr0[b * kMaxZones + 2] - 1; }
        }
        for (int k = 0; k < h->ngroups; ++k) {
            for (int b = h->gb[k]; b < h->gb[k + 1]; ++b)
                for (int z = 0; z < nz[b]; ++z)
                    for (int t = zr0[b * kMaxZones + z]; t <= zone_end(b, z); ++t) (zmix[b * kMaxZones + z] ? slab : plain).push_back(b * L + t);
            h->main_off[k + 1] = (int)plain.size();
            h->slab_off[k + 1] = (int)slab.size();
        }
    }
    for (int b = 0; b < B; ++b) {
        if (!(mu0[b] > 0)) return fail(SOSRT_E_INVALID, "column %d: mu0 must be > 0", b);
        sc[0 * mb + b] = mu0[b];
        sc[1 * mb + b] = grd_alb ? grd_alb[b] : 0.0;
        sc[2 * mb + b] = alb_atm[b];
        sc[4 * mb + b] = dtau_atm ? dtau_atm[b] : 1.0;
        sc[6 * mb + b] = tauStar_tot[b];
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->d_scal, sc.data(), 7 * mb * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (geometry == SOSRT_GEOM_THREE_ZONE) {
        HIPCHK(hipMemcpyAsync(h->d_nz, nz.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_zr0, zr0.data(), (size_t)B * kMaxZones * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_zmix, zmix.data(), (size_t)B * kMaxZones * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_zwr, zwr.data(), (size_t)B * kMaxZones * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_zdtr, zdtr.data(), (size_t)B * kMaxZones * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_idx_up, iup.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_idx_down, idn.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (!slab.empty())
            HIPCHK(hipMemcpyAsync(h->d_slabrows, slab.data(), slab.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (!plain.empty())
            HIPCHK(hipMemcpyAsync(h->d_mainrows, plain.data(), plain.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));   // the staging vectors go out of scope
    h->nslab = (int)slab.size();
    h->nmain = (int)plain.size();
    h->max_main = L; h->max_slab = 0;
    if (geometry == SOSRT_GEOM_THREE_ZONE) {
        h->max_main = 0;
        for (int b = 0; b < B; ++b) {
            int ns = 0;
            for (int z = 0; z < nz[b]; ++z)
                if (zmix[b * kMaxZones + z]) ns += zone_end(b, z) - zr0[b * kMaxZones + z] + 1;
            h->max_slab = ns > h->max_slab ? ns : h->max_slab;
            h->max_main = L - ns > h->max_main ? L - ns : h->max_main;
        }
    }
    h->mix_groups = 0;
    h->mix_dirty = true;
    h->w32_dirty = true;
    if (geometry == SOSRT_GEOM_THREE_ZONE && h->nslab > 0) {
        // distinct slab coefficient pairs (spec:321: (w_atm/4) f_atm on W_atm, (w_aer/4) f_aer on W_aer), per (column, slab)
        std::vector<double> gca, gcr;
        std::vector<int> gid((size_t)B * kMaxZones, -1), gcol(B, 0);
        bool ok = true;
        for (int b = 0; b < B && ok; ++b) {
            for (int z = 0; z < nz[b] && ok; ++z) {
                if (!zmix[b * kMaxZones + z]) continue;
                const double da = dtau_atm[b], dr = zdtr[b * kMaxZones + z];
                const double ca = (alb_atm[b] / 4) * (da / (da + dr)), cr = (zwr[b * kMaxZones + z] / 4) * (dr / (da + dr));
                int k = 0;
                while (k < (int)gca.size() && !(gca[k] == ca && gcr[k] == cr)) ++k;
                if (k == (int)gca.size()) {
                    if (k == sosrt_handle::kMaxMixGroups) { ok = false; break; }
                    gca.push_back(ca); gcr.push_back(cr);
                }
                gid[b * kMaxZones + z] = k;
                gcol[b] = k;                         // the live-column tilings: one slab per column
            }
        }
        if (ok) {
            const size_t per = (size_t)h->g.Dp * h->g.Wld, need = per * gca.size();
            if (need > h->mix_capacity) {
                if (h->d_Wmix) hipFree(h->d_Wmix);
                h->d_Wmix = nullptr; h->mix_capacity = 0;
                if (hipMalloc((void**)&h->d_Wmix, need * sizeof(double)) == hipSuccess) h->mix_capacity = need;
                else (void)hipGetLastError();
            }
            if (h->mix_capacity >= need) {
                HIPCHK(hipMemcpy(h->d_mixca, gca.data(), gca.size() * sizeof(double), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(h->d_mixcr, gcr.data(), gcr.size() * sizeof(double), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(h->d_mixgroup, gcol.data(), B * sizeof(int), hipMemcpyHostToDevice));
                h->mix_groups = (int)gca.size();
                // slab rows of the dense contraction listed, per column group of the order loop, coefficient pair by
                // coefficient pair, every pair padded to whole 32-row tiles
                std::vector<int> grouped, tilegroup;
                for (int cg = 0; cg < h->ngroups; ++cg) {
                    for (int k = 0; k < h->mix_groups; ++k) {
                        for (int b = h->gb[cg]; b < h->gb[cg + 1]; ++b)
                            for (int z = 0; z < nz[b]; ++z)
                                if (gid[b * kMaxZones + z] == k)
                                    for (int t = zr0[b * kMaxZones + z]; t <= zone_end(b, z); ++t) grouped.push_back(b * L + t);
                        while (grouped.size() % 64) grouped.push_back(-1);   // whole 64-row tiles (two 32-row tiles of the same pair)
                        while (tilegroup.size() < grouped.size() / 32) tilegroup.push_back(k);
                    }
                    h->slab_off[cg + 1] = (int)grouped.size();
                }
                HIPCHK(hipMemcpy(h->d_slabrows, grouped.data(), grouped.size() * sizeof(int), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(h->d_slabtilegroup, tilegroup.data(), tilegroup.size() * sizeof(int), hipMemcpyHostToDevice));
                h->nslab = (int)grouped.size();
            }
        }
    }
    h->B = B; h->geom = geometry; h->surface = surface;
    h->have_cols = true;
    return 0;
}

int sosrt_set_columns(sosrt_t* h, int B, int geometry, int surface, const int* idx_up, const int* idx_down,
                      const double* mu0, const double* grd_alb, const double* alb_atm, const double* alb_aer,
                      const double* dtau_atm, const double* dtau_aer, const double* tauStar_tot) {
    if (int e = need_gpu(h)) return e;
    if (B < 1 || B > h->max_batch) return fail(SOSRT_E_INVALID, "B=%d outside 1..max_batch=%d", B, h->max_batch);
    if (!mu0 || !alb_atm || !tauStar_tot) return fail(SOSRT_E_INVALID, "mu0, alb_atm and tauStar_tot are required");
    std::vector<int> nz(B, 1), zr0((size_t)B * kMaxZones, 0), zmix((size_t)B * kMaxZones, 0);
    std::vector<double> zwr((size_t)B * kMaxZones, 0.0), zdtr((size_t)B * kMaxZones, 0.0);
    if (geometry == SOSRT_GEOM_THREE_ZONE) {
        if (!idx_up || !idx_down || !grd_alb || !alb_aer || !dtau_atm || !dtau_aer)
            return fail(SOSRT_E_INVALID, "three-zone geometry needs idx_up, idx_down, grd_alb, alb_aer, dtau_atm, dtau_aer");
        if (surface != SOSRT_SURFACE_SPECULAR && surface != SOSRT_SURFACE_LAMBERTIAN && surface != SOSRT_SURFACE_LAMBERTIAN_README)
            return fail(SOSRT_E_INVALID, "three-zone geometry needs a specular or lambertian surface");
        for (int b = 0; b < B; ++b) {
            if (idx_up[b] < 1 || idx_down[b] < idx_up[b] || idx_down[b] > h->L - 2)
                return fail(SOSRT_E_INVALID, "column %d: need 1 <= idx_up <= idx_down <= nb_layers-2 (got %d, %d)", b,
                            idx_up[b], idx_down[b]);
            // above / inside / below the aerosol slab (spec:113-449)
            nz[b] = 3;
            zr0[b * kMaxZones + 1] = idx_up[b]; zr0[b * kMaxZones + 2] = idx_down[b] + 1;
            zmix[b * kMaxZones + 1] = 1;
            zwr[b * kMaxZones + 1] = alb_aer[b];
            zdtr[b * kMaxZones + 1] = dtau_aer[b];
        }
    } else if (geometry == SOSRT_GEOM_SINGLE_SLAB) {
        surface = SOSRT_SURFACE_NONE;
    } else {
        return fail(SOSRT_E_INVALID, "unknown geometry %d", geometry);
    }
    return set_columns_impl(h, B, geometry, surface, nz, zr0, zmix, zwr, zdtr, mu0, grd_alb, alb_atm, dtau_atm, tauStar_tot);
}

int sosrt_set_columns_zones(sosrt_t* h, int B, int surface, int nzmax, const int* nz_in, const int* zone_r0, const int* zone_mix,
                            const double* mu0, const double* grd_alb, const double* alb_atm, const double* dtau_atm,
                            const double* zone_alb_aer, const double* zone_dtau_aer, const double* tauStar_tot) {
    if (int e = need_gpu(h)) return e;
    if (B < 1 || B > h->max_batch) return fail(SOSRT_E_INVALID, "B=%d outside 1..max_batch=%d", B, h->max_batch);
    if (!nz_in || !zone_r0 || !zone_mix || !mu0 || !grd_alb || !alb_atm || !dtau_atm || !zone_alb_aer || !zone_dtau_aer || !tauStar_tot)
        return fail(SOSRT_E_INVALID, "null argument");
    if (nzmax < 1 || nzmax > kMaxZones) return fail(SOSRT_E_INVALID, "nzmax must be in 1..%d (got %d)", kMaxZones, nzmax);
    if (surface != SOSRT_SURFACE_SPECULAR && surface != SOSRT_SURFACE_LAMBERTIAN && surface != SOSRT_SURFACE_LAMBERTIAN_README)
        return fail(SOSRT_E_INVALID, "a zone table needs a specular or lambertian surface");
    std::vector<int> nz(B), zr0((size_t)B * kMaxZones, 0), zmix((size_t)B * kMaxZones, 0);
    std::vector<double> zwr((size_t)B * kMaxZones, 0.0), zdtr((size_t)B * kMaxZones, 0.0);
    for (int b = 0; b < B; ++b) {
        nz[b] = nz_in[b];
        if (nz[b] < 1 || nz[b] > nzmax) return fail(SOSRT_E_INVALID, "column %d: %d zones, expected 1..%d", b, nz[b], nzmax);
        for (int z = 0; z < nz[b]; ++z) {
            const int r0 = zone_r0[b * nzmax + z], mix = zone_mix[b * nzmax + z] != 0;
            if (z == 0 ? r0 != 0 : !(r0 > zr0[b * kMaxZones + z - 1] && r0 < h->L))
                return fail(SOSRT_E_INVALID, "column %d: zone %d starts at row %d (zones start at 0 and ascend, below nb_layers)", b, z, r0);
            if (z > 0 && mix && zmix[b * kMaxZones + z - 1]) return fail(SOSRT_E_INVALID, "column %d: two adjacent aerosol zones (%d, %d): merge them", b, z - 1, z);
            // the reference's slab lies strictly inside the column (idx_up >= 1, idx_down <= L-2, spec:40): its formulas
            // read the rows either side of a slab
            if (mix && (z == 0 || z == nz[b] - 1)) return fail(SOSRT_E_INVALID, "column %d: an aerosol zone must have a clear zone above and below it", b);
            zr0[b * kMaxZones + z] = r0; zmix[b * kMaxZones + z] = mix;
            zwr[b * kMaxZones + z] = mix ? zone_alb_aer[b * nzmax + z] : 0.0;
            zdtr[b * kMaxZones + z] = mix ? zone_dtau_aer[b * nzmax + z] : 0.0;
            if (mix && !(zdtr[b * kMaxZones + z] >= 0)) return fail(SOSRT_E_INVALID, "column %d zone %d: dtau_aer must be >= 0", b, z);
        }
    }
    return set_columns_impl(h, B, SOSRT_GEOM_THREE_ZONE, surface, nz, zr0, zmix, zwr, zdtr, mu0, grd_alb, alb_atm, dtau_atm, tauStar_tot);
}

// ---------------------------------------------------------------------------------------------
// step level
// ---------------------------------------------------------------------------------------------
int sosrt_first_order(sosrt_t* h, int B, const double* tau, const double* P0_atm, const double* P0_aer,
                      double* I1_out) {
    if (int e = check_ready(h, B, false)) return e;
    if (!tau || !P0_atm || !I1_out) return fail(SOSRT_E_INVALID, "null argument");
    if (h->geom == SOSRT_GEOM_THREE_ZONE && !P0_aer) return fail(SOSRT_E_INVALID, "three-zone geometry needs P0_aer");
    HIPCHK(hipSetDevice(h->device));
    h->resident = false;                     // d_tau is overwritten
    const size_t n = (size_t)B * h->L * h->D;
    HIPCHK(hipMemcpyAsync(h->d_tau, tau, (size_t)B * h->L * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_P0a, P0_atm, (size_t)B * h->D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (P0_aer) HIPCHK(hipMemcpyAsync(h->d_P0r, P0_aer, (size_t)B * h->D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_prepare(h->stream, h->g, B, h->geom, h->surface, scalars_of(h), h->d_tau, h->d_desc, h->d_rca, h->d_rcr);
    prof_begin(h, SOSRT_K_FIRST);
    if (h->first_order_mode == SOSRT_FIRST_ORDER_README)
        launch_first_order_readme(h->stream, h->g, h->d_w, B, h->d_tau, h->d_P0a, P0_aer ? h->d_P0r : nullptr, h->d_desc, h->d_InA,
                                  nullptr, nullptr, 0, make_conv(h, 0), 0);
    else
        launch_first_order(h->stream, h->g, B, h->d_tau, h->d_P0a, P0_aer ? h->d_P0r : nullptr, h->d_desc, h->d_InA, nullptr,
                           nullptr, 0, make_conv(h, 0), 0);
    prof_end(h, SOSRT_K_FIRST);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(I1_out, h->d_InA, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int sosrt_source(sosrt_t* h, int B, const double* In_1, double* Jn_out) {
    if (int e = check_ready(h, B, true)) return e;
    if (!In_1 || !Jn_out) return fail(SOSRT_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->device));
    if (int e = ensure_w32(h)) return e;
    const size_t n = (size_t)B * h->L * h->D;
    HIPCHK(hipMemcpyAsync(h->d_InA, In_1, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // the row coefficients depend on tau only through the zone bounds; prepare needs a tau buffer
    // for the a4b buckets, which the source function does not use
    launch_prepare(h->stream, h->g, B, h->geom, h->surface, scalars_of(h), h->d_tau, h->d_desc, h->d_rca, h->d_rcr);
    run_source(h, h->d_InA, h->d_Jn, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(Jn_out, h->d_Jn, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int sosrt_transport(sosrt_t* h, int B, const double* tau, const double* Jn, double* In_out, int* status_out) {
    if (int e = check_ready(h, B, false)) return e;
    if (!tau || !Jn || !In_out) return fail(SOSRT_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->device));
    h->resident = false;                     // d_tau is overwritten
    const size_t n = (size_t)B * h->L * h->D;
    HIPCHK(hipMemcpyAsync(h->d_tau, tau, (size_t)B * h->L * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_Jn, Jn, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_prepare(h->stream, h->g, B, h->geom, h->surface, scalars_of(h), h->d_tau, h->d_desc, h->d_rca, h->d_rcr);
    HIPCHK(hipMemsetAsync(h->d_InB, 0, n * sizeof(double), h->stream));
    prof_begin(h, SOSRT_K_SMALLMU);
    launch_smallmu(h->stream, h->g, B, h->d_tau, h->d_Jn, h->d_InB, h->d_desc, nullptr);
    prof_end(h, SOSRT_K_SMALLMU);
    prof_begin(h, SOSRT_K_TRANSPORT);
    const bool ring_like = h->transport_mode >= 2 && h->ring_ok && (h->max_nz <= kRingZones || transport_ring_fits(h->g, h->max_nz));
    const int nzcap = h->max_nz > kRingZones ? h->max_nz : kRingZones;
    if (h->transport_mode >= 1 && h->fast_ok && (h->max_nz <= kRingZones || ring_like)) {
        launch_attenuation(h->stream, h->g, B, h->d_tau, h->d_E, nullptr);
        HIPCHK(hipMemsetAsync(h->d_redo, 0, B * sizeof(int), h->stream));
        launch_transport(h->stream, h->g, B, h->d_tau, h->d_Jn, h->d_InB, nullptr, nullptr, 0, h->d_desc, make_conv(h, 0), 0, 0, h->d_E,
                         (h->transport_mode == 4 && h->scan_ok && transport_scan_fits(h->g, nzcap, false)) ? 4 : (ring_like ? 3 : 1),
                         nullptr, 0, nullptr, 0, 0, nullptr, nullptr, nzcap);
        if (h->N - 3 > 61 && !(h->transport_mode >= 2 && h->ring_ok))
            launch_transport(h->stream, h->g, B, h->d_tau, h->d_Jn, h->d_InB, nullptr, nullptr, 0, h->d_desc, make_conv(h, 0), 0, 0, h->d_E, 2);
    } else {
        launch_transport(h->stream, h->g, B, h->d_tau, h->d_Jn, h->d_InB, nullptr, nullptr, 0, h->d_desc, make_conv(h, 0), 0, 0, nullptr, 0);
    }
    prof_end(h, SOSRT_K_TRANSPORT);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(In_out, h->d_InB, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (status_out) HIPCHK(hipMemcpyAsync(status_out, h->d_status, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// column level
// ---------------------------------------------------------------------------------------------
// spec:309 with In = ones when the first order is supplied by the caller
__global__ void k_init_from_I1(Grid g, const double* __restrict__ I1, double* __restrict__ In1, double* __restrict__ I,
                               double* __restrict__ saved, size_t saved_col_stride, Conv cv) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t n = (size_t)g.L * g.D;
    const double* src = I1 + (size_t)b * n;
    for (size_t i = tid; i < n; i += blockDim.x) {
        const double v = src[i];
        In1[(size_t)b * n + i] = v;
        I[(size_t)b * n + i] = v;
        if (saved) saved[(size_t)b * saved_col_stride + i] = v;
    }
    if (tid == 0) {
        // the reference's test with In = ones: python max over the rows, in order
        double a = 1.0 / src[g.N];
        for (int m = g.N + 1; m < g.D; ++m) { const double x = 1.0 / src[m]; if (x > a) a = x; }
        const double* last = src + (size_t)(g.L - 1) * g.D;
        double bb = 1.0 / last[0];
        for (int m = 1; m < g.N; ++m) { const double x = 1.0 / last[m]; if (x > bb) bb = x; }
        const double r = (bb > a) ? bb : a;
        cv.ratio[b] = r; cv.norders[b] = 1; cv.status[b] = SOSRT_COL_OK;
        const int go = (r >= cv.tol) ? 1 : 0;
        cv.active[b] = go;
        if (go) atomicAdd(cv.nactive, 1);
    }
}

int sosrt_solve_dev(sosrt_t* h, int B, const double* d_tau, const double* d_P0_atm, const double* d_P0_aer, double tol,
                    const double* d_I1_in, double* d_I_out, double* d_I_saved_out, int* d_n_orders_out,
                    int* d_status_out) {
    if (int e = check_ready(h, B, true)) return e;
    if (!d_tau || !d_I_out) return fail(SOSRT_E_INVALID, "null argument");
    if (!d_I1_in && !d_P0_atm) return fail(SOSRT_E_INVALID, "P0_atm is required unless I1 is supplied");
    if (!d_I1_in && h->geom == SOSRT_GEOM_THREE_ZONE && !d_P0_aer) return fail(SOSRT_E_INVALID, "three-zone geometry needs P0_aer");
    if (h->max_orders >= 65536) return fail(SOSRT_E_INVALID, "max_orders must be < 65536");
    HIPCHK(hipSetDevice(h->device));
    if (int e = ensure_w32(h)) return e;
    hipStream_t s = h->stream;
    const Grid& g = h->g;
    const size_t LD = (size_t)h->L * h->D;
    const size_t saved_stride = (size_t)h->saved_slots * LD;
    const int NG = h->ngroups;

    prof_break(h);
    // per-sweep setup on the caller's stream: zone tables, shared attenuation tables, combined slab matrices
    // this solve's counters were zeroed by the previous solve's first kernel (or at sosrt_create); its own first kernel zeroes
    // the other set, clears the redo flags and hashes the optical-depth profiles -- one launch instead of four
    h->nactive_set ^= 1;
    h->d_nactive = h->d_nactive_sets + h->nactive_set * (sosrt_handle::kMaxGroups + 1);
    SolveSetup su;
    su.zero_next = h->d_nactive_sets + (h->nactive_set ^ 1) * (sosrt_handle::kMaxGroups + 1);
    su.n_zero = sosrt_handle::kMaxGroups + 1;
    su.redo = h->d_redo;
    su.hash = h->d_tauhash;
    su.scan_sync = h->d_scan_sync;
    launch_prepare(s, g, B, h->geom, h->surface, scalars_of(h), d_tau, h->d_desc, h->d_rca, h->d_rcr,
                   h->d_nactive + sosrt_handle::kMaxGroups, su);
    h->need_small = true;
    const int small_tag = ((++h->pub_seq) & 0x3fffffff) | 0x40000000;       // never equals an order tag
    bool small_published = false;
    const SolveShape shape = solve_shape(h, h->max_nz);
    const bool fast = shape.fast;
    const int nzcap = shape.nzcap;
    if (h->use_etab || fast) {
        // one attenuation table per distinct optical-depth profile
        launch_tau_groups(s, g, B, d_tau, h->d_tauhash, h->d_erep, h->d_nactive + sosrt_handle::kMaxGroups,
                          h->h_pub + 8 * sosrt_handle::kMaxGroups, small_tag);
        launch_attenuation(s, g, B, d_tau, h->d_E, h->d_erep);
        small_published = true;
    }
    if (int e = ensure_matrices(h, s)) return e;
    bool forked = false;
    // an error return after the fork still joins the internal stream back onto the caller's
    int ol_held[sosrt_handle::kMaxGroups] = {0, 0};          // CUs this solve's order-loop launches hold (ol_acquire)
    auto bail = [&](int code) {
        if (forked && hipEventRecord(h->ev_join, h->stream2) == hipSuccess) (void)hipStreamWaitEvent(s, h->ev_join, 0);
        for (int k = 0; k < NG; ++k)
            if (ol_held[k]) {                                 // (a launch still running keeps its CUs until its stream has drained)
                (void)hipStreamSynchronize(group_stream(h, k));
                ol_release(h->device, ol_held[k]);
                ol_held[k] = 0;
            }
        return code;
    };

    // Order loop (spec:309-458), per column group.  Converged columns are masked on the device (every kernel of an
    // order returns at once for them).  r_k = number of live columns of the group after order k is written to a
    // pinned slot by the first workgroup of order k+1's source-function launch; before launching order k+1 the
    // host checks r_{k-1}, which is there as soon as order k has started, so a stream never drains inside the loop
    // and at most one launch group runs on a fully converged group.  With two groups the host feeds them in turn:
    // each stream always holds the next order of its group, and the GPU overlaps the contraction of one group
    // (MFMA-bound) with the transport of the other (HBM-bound).
    struct GroupState {
        int b0 = 0, nb = 0, n = 1, known = 0;
        bool done = false, started = false;
        bool ol_pending = false, ol_off = false;             // an order-loop launch is running; one was refused: no more in this solve
        int ol_tag = 0;
        unsigned ol_polls = 0;
        double *In_1 = nullptr, *In = nullptr;
        Conv cv;
    } gs[sosrt_handle::kMaxGroups];
    h->ol_launches = 0; h->ol_refused = 0;
    for (bool& u : h->ol_group_used) u = false;
    bool used_order_loop = false;
    const int tagbase = ((++h->pub_seq) & 0x3fff) << 16;      // tag of order n = tagbase + n
    for (int k = 0; k < NG; ++k) {
        GroupState& q = gs[k];
        q.b0 = h->gb[k]; q.nb = h->gb[k + 1] - h->gb[k]; q.known = q.nb;
        q.In_1 = h->d_InA; q.In = h->d_InB;                   // whole-batch buffers; every kernel gets its group's offset
        // The coded first order writes I = I1 only: the second order's contraction reads its operand there (the same numbers),
        // and the first order is bound by its stores (one 8 L D array instead of two: 91 -> 50 us for 512 columns)
        if (!d_I1_in && h->first_order_mode != SOSRT_FIRST_ORDER_README) q.In_1 = d_I_out;
        q.cv = make_conv(h, tol);
        q.cv.active += q.b0; q.cv.norders += q.b0; q.cv.status += q.b0; q.cv.ratio += q.b0; q.cv.redo += q.b0;
        q.cv.nactive = h->d_nactive + k;
    }
    auto start_group = [&](int k) {
        GroupState& q = gs[k];
        q.started = true;
        hipStream_t sg = group_stream(h, k);
        const size_t fo = (size_t)q.b0 * LD;
        prof_begin(h, SOSRT_K_FIRST, k);
        if (d_I1_in)
            hipLaunchKernelGGL(k_init_from_I1, dim3(q.nb), dim3(256), 0, sg, g, d_I1_in + fo, q.In_1 + fo, d_I_out + fo,
                               d_I_saved_out ? d_I_saved_out + (size_t)q.b0 * saved_stride : nullptr, saved_stride, q.cv);
        else if (h->first_order_mode == SOSRT_FIRST_ORDER_README)
            launch_first_order_readme(sg, g, h->d_w, q.nb, d_tau + (size_t)q.b0 * h->L, d_P0_atm + (size_t)q.b0 * g.D,
                                      d_P0_aer ? d_P0_aer + (size_t)q.b0 * g.D : nullptr, h->d_desc + q.b0, q.In_1 + fo,
                                      d_I_out + fo, d_I_saved_out ? d_I_saved_out + (size_t)q.b0 * saved_stride : nullptr,
                                      saved_stride, q.cv, 1);
        else
            launch_first_order(sg, g, q.nb, d_tau + (size_t)q.b0 * h->L, d_P0_atm + (size_t)q.b0 * g.D,
                               d_P0_aer ? d_P0_aer + (size_t)q.b0 * g.D : nullptr, h->d_desc + q.b0, d_I_out + fo, nullptr,
                               d_I_saved_out ? d_I_saved_out + (size_t)q.b0 * saved_stride : nullptr, saved_stride, q.cv, 1);
        prof_end(h, SOSRT_K_FIRST, k);
    };
    start_group(0);
    if (NG > 1) {
        // The second column group runs on the internal stream from here on.  The fork comes BEHIND the first group's first order:
        // the two calls cost the host ~25 us, which the GPU -- 20 us of setup kernels ahead of the host at this point -- would
        // otherwise wait for; the second group starts half a cycle after the first anyway.
        HIPCHK(hipEventRecord(h->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        forked = true;
    }
    if (small_published && g.nsmall > 0) {
        // k_prepare's verdict on k_smallmu, published by the second kernel of the solve: by now it has long run
        volatile int* slot = h->h_pub + 8 * sosrt_handle::kMaxGroups;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 1; __atomic_load_n(&slot[1], __ATOMIC_ACQUIRE) != small_tag; ++it) {
            if ((it & 0x3fff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                return bail(fail(SOSRT_E_HIP, "order loop: no progress for 120 s"));
            __builtin_ia32_pause();
        }
        h->need_small = slot[0] != 0;
    }
    int live_groups = NG, n_max = 1;
    while (live_groups > 0) {
        bool progressed = false;
        for (int k = 0; k < NG; ++k) {
            GroupState& q = gs[k];
            if (q.done) continue;
            if (q.ol_pending) {
                // the group's remaining orders run in one launch; its last workgroup reports {state, tag} to pinned memory
                volatile int* dw = h->h_oldone + 2 * k;
                if (__atomic_load_n(&dw[1], __ATOMIC_ACQUIRE) != q.ol_tag) {
                    if ((++q.ol_polls & 0x3fff) != 0) continue;
                    const hipError_t qe = hipStreamQuery(group_stream(h, k));
                    if (qe != hipSuccess && qe != hipErrorNotReady) return bail(fail(SOSRT_E_HIP, "order loop: %s", hipGetErrorString(qe)));
                    if (qe == hipSuccess && __atomic_load_n(&dw[1], __ATOMIC_ACQUIRE) != q.ol_tag)
                        return bail(fail(SOSRT_E_HIP, "order loop: the order-loop launch ended without reporting"));
                    continue;
                }
                const int state = dw[0];
                q.ol_pending = false;
                ol_release(h->device, ol_held[k]);
                ol_held[k] = 0;
                progressed = true;
#ifdef SOSRT_OL_STAMPS
                if (h->d_ollog && getenv("SOSRT_OL_LOG")) {
                    std::vector<unsigned long long> lg(65001);
                    (void)hipStreamSynchronize(group_stream(h, k));
                    (void)hipMemcpy(lg.data(), h->d_ollog, lg.size() * 8, hipMemcpyDeviceToHost);
                    if (FILE* f = fopen(getenv("SOSRT_OL_LOG"), "a")) {
                        fprintf(f, "# launch columns<=%d order0=%d events=%llu\n", q.known, q.n + 1, lg[0]);
                        for (unsigned long long i = 0; i < lg[0] && i < 65000; ++i)
                            fprintf(f, "%llu %llu %llu %llu\n", lg[1 + i] >> 48, (lg[1 + i] >> 40) & 0xff, (lg[1 + i] >> 32) & 0xff, lg[1 + i] & 0xffffffffull);
                        fclose(f);
                    }
                }
#endif
                if (state == kOlReady) { q.done = true; --live_groups; h->ol_group_used[k] = true; continue; }
                if (state == kOlAborted) {
                    // never expected: say where the launch stood (the words of the launch, first columns)
                    std::vector<int> w(order_loop_sync_ints(q.known < 6 ? q.known : 6));
                    (void)hipStreamSynchronize(group_stream(h, k));
                    (void)hipMemcpy(w.data(), h->d_olsync + (size_t)k * order_loop_sync_ints(kOrderLoopMaxCols), w.size() * sizeof(int), hipMemcpyDeviceToHost);
                    std::string cols;
                    for (int c = 0; c < (q.known < 6 ? q.known : 6); ++c) {
                        char buf[96];
                        const int* cs = w.data() + kOlCols + c * kOlColStride;
                        snprintf(buf, sizeof buf, " [%d: orders %d stop %d tiles %d]", c, cs[kOlOrdDone], cs[kOlColStop], cs[kOlJnDone]);
                        cols += buf;
                    }
                    return bail(fail(SOSRT_E_HIP, "order loop: a workgroup of the order-loop launch gave up waiting (arrived %d, left %d, order %d on, %d columns:%s)",
                                     w[kOlArrive], w[kOlLeft], q.n + 1, q.known, cols.c_str()));
                }
                // not resident (another process's kernels held CUs): nothing was touched; the one-order kernels take over
                q.ol_off = true;
                ++h->ol_refused;
            }
            if (!q.started) {
                // Staggered start: the dense orders of this group (they fill the GPU) run beside the long tail of the
                // previous one (a few workgroups per order, latency-bound), instead of both being dense, then both in
                // their tails, together.
                const GroupState& p = gs[k - 1];
                if (!(p.done || (p.n >= 2 && p.known <= h->stagger * p.nb))) continue;
                start_group(k);
            }
            progressed = true;
            if (q.n >= h->order_budget) { q.done = true; --live_groups; continue; }
            if (q.n >= 2) {
                const int live = wait_published(h, k, tagbase + q.n - 1);
                if (live < 0) return bail(live);
                if (live == 0) { q.done = true; --live_groups; continue; }
                q.known = live;
            }
            hipStream_t sg = group_stream(h, k);
            const size_t fo = (size_t)q.b0 * LD;
            const double* tau_g = d_tau + (size_t)q.b0 * h->L;
            const int* erep_g = h->d_erep + q.b0;     // values are whole-batch column ids; the table base is not offset
            // ---- the plan of this order (plan_order: one place for the whole policy) ----
            OrderInputs oi;
            oi.nb = q.nb; oi.known = q.known; oi.surface = h->surface;
            oi.simple_zones = h->simple_zones; oi.slabs_mixed = h->nslab == 0 || h->mix_groups > 0;
            oi.need_small = g.nsmall > 0 && h->need_small; oi.saving = d_I_saved_out != nullptr;
            oi.orders_left = h->order_budget - q.n;
            oi.cu_share = q.ol_off ? 0 : h->cu_count / NG;
            LaunchPlan pl = plan_order(h, shape, oi);
            if (pl.order_loop) {
                // its workgroups wait for each other: they must all fit the CUs no other order-loop launch of this process holds
                const int got = ol_acquire(h->device, h->cu_count, pl.ol_grid, (int)(pl.ol_parts * q.known / h->ol_frac));
                if (got == 0) { oi.cu_share = 0; pl = plan_order(h, shape, oi); }
                else { pl.ol_grid = got; ol_held[k] = got; }
            }
            // (mode 2, for the tests of the refusal path: a grid of twice the CUs can never be resident -- the handshake times out,
            // nothing has been touched, and the order is run as two launches)
            if (pl.order_loop && h->order_loop == 2) pl.ol_grid = 2 * h->cu_count;
            if (pl.order_loop) {
                OrderLoopArgs oa;
                Grid gt = g;
                gt.nsmall = 0;
                oa.t = TransportArgs{gt, tau_g, h->d_Jn + fo, nullptr, d_I_out + fo, nullptr, 0, h->d_desc + q.b0, q.cv, 0, 1, h->d_E, erep_g, nullptr};
                oa.t.nzcap = kRingZones;
                oa.t.scan_split = pl.ol_parts > 1 ? 1 : 0;
                oa.t.scan_scratch = h->d_scan_scratch + (size_t)q.b0 * transport_scan_scratch_doubles();
                oa.t.scan_sync = h->d_scan_sync + 2 * q.b0;
                GemmArgs& ga = oa.gm;
                ga.A = nullptr; ga.Wa = h->d_Wa_s; ga.Wr = h->d_Wr_s; ga.ca = h->d_rca; ga.cr = h->d_rcr;
                ga.D = g.D; ga.Dp = g.Dp; ga.Wld = g.Wld; ga.L = h->L; ga.C = h->d_Jn;
                ga.sym = 1; ga.Ks = (g.N + GEMM_KC - 1) / GEMM_KC * GEMM_KC;
                ga.max_main = h->max_main; ga.max_slab = h->max_slab;
                ga.idx_up = h->nslab > 0 ? h->d_idx_up : nullptr; ga.idx_down = h->nslab > 0 ? h->d_idx_down : nullptr;
                if (h->mix_groups > 0) { ga.Wmix = h->d_Wmix_s; ga.mix_group = h->d_mixgroup; }
                oa.in0 = q.In_1;
                oa.gbufP = q.In;
                oa.gbufQ = (q.In_1 == d_I_out) ? h->d_InA : q.In_1;   // (after the second order: the buffer the first order left unused)
                oa.bufP = oa.gbufP + fo; oa.bufQ = oa.gbufQ + fo;
                oa.order0 = q.n + 1; oa.kmax = h->order_budget - q.n;
                oa.B = q.nb; oa.col0 = q.b0;
                oa.fixcap = (int)(0.06 * g.N) + 1;
                oa.sync = h->d_olsync + (size_t)k * order_loop_sync_ints(kOrderLoopMaxCols);
                oa.host_done = h->h_oldone + 2 * k;
                oa.tag = q.ol_tag = ((++h->pub_seq) & 0x3fffffff) | 0x40000000;
#ifdef SOSRT_OL_STAMPS   // diagnostic builds: SOSRT_OL_LOG=<file> receives the launch's event log (tools/ol_timeline.py)
                if (getenv("SOSRT_OL_LOG")) {
                    if (!h->d_ollog) { if (int e = dalloc(&h->d_ollog, 65001)) return bail(e); }
                    HIPCHK(hipMemsetAsync(h->d_ollog, 0, 8, sg));
                    oa.log = h->d_ollog;
                }
#endif
                prof_break(h);
                HIPCHK(hipMemsetAsync(oa.sync, 0, order_loop_sync_ints(q.known) * sizeof(int), sg));
                prof_begin(h, SOSRT_K_ORDER_LOOP, k);
                const hipError_t le = launch_order_loop(sg, pl.ol_grid, pl.ol_parts > 1, oa);
                prof_end(h, SOSRT_K_ORDER_LOOP, k);
                if (le != hipSuccess) return bail(fail(SOSRT_E_HIP, "order-loop launch failed: %s", hipGetErrorString(le)));
                q.ol_pending = true;
                used_order_loop = true;
                ++h->ol_launches;
                continue;
            }
            const int n = ++q.n;
            n_max = n > n_max ? n : n_max;
            // this launch also publishes the group's live count after order n-1
            const int tail_cols = pl.tail_cols;
            run_source(h, q.In_1, h->d_Jn, h->d_active, tail_cols, tagbase + n - 1, k, q.known == q.nb, pl.gemm == SOSRT_PLAN_GEMM_LIVE16_REGS,
                       tail_cols > 0 ? 0 : pl.live_cap);
            if (g.nsmall > 0 && h->need_small) {      // skipped once the device has reported that every such lane is rewritten anyway
                prof_begin(h, SOSRT_K_SMALLMU, k);
                launch_smallmu(sg, g, q.nb, tau_g, h->d_Jn + fo, q.In + fo, h->d_desc + q.b0, q.cv.active);
                prof_end(h, SOSRT_K_SMALLMU, k);
            }
            prof_begin(h, SOSRT_K_TRANSPORT, k);
            double* sv_n = (d_I_saved_out && n <= h->saved_slots) ? d_I_saved_out + (size_t)q.b0 * saved_stride + (size_t)(n - 1) * LD : nullptr;
            if (pl.transport != SOSRT_PLAN_TRANSPORT_GENERAL) {
                // once the device has reported that no |mu| < 0.01 lane keeps its k_smallmu value, the ring kernel
                // need not stage those rows either
                Grid gt = g;
                if (pl.transport >= SOSRT_PLAN_TRANSPORT_RING && !h->need_small) gt.nsmall = 0;
                launch_transport(sg, gt, q.nb, tau_g, h->d_Jn + fo, q.In + fo, d_I_out + fo, sv_n, saved_stride, h->d_desc + q.b0, q.cv, n, 1,
                                 h->d_E, pl.transport, erep_g, pl.live_cap, h->d_livelist + q.b0, NG > 1 ? h->coresident_slots : 0,
                                 pl.parts > 1 ? 1 : 0, h->d_scan_scratch + (size_t)q.b0 * transport_scan_scratch_doubles(),
                                 h->d_scan_sync + 2 * q.b0, nzcap);
                if (pl.repair)                           // register-streaming kernel: a search that leaves wave 0 is redone by the
                    launch_transport(sg, g, q.nb, tau_g, h->d_Jn + fo, q.In + fo, d_I_out + fo, sv_n, saved_stride, h->d_desc + q.b0, q.cv, n, 1,
                                     h->d_E, 2, erep_g);         // general kernel (flag cv.redo); the ring kernel redoes it itself
            } else {
                launch_transport(sg, g, q.nb, tau_g, h->d_Jn + fo, q.In + fo, d_I_out + fo, sv_n, saved_stride, h->d_desc + q.b0, q.cv, n, 1,
                                 h->use_etab ? h->d_E : nullptr, 0, erep_g);
            }
            prof_end(h, SOSRT_K_TRANSPORT, k);
            if (q.In_1 == d_I_out) { q.In_1 = q.In; q.In = h->d_InA; }      // (after the second order: the buffer the first order left unused)
            else { double* tmp = q.In_1; q.In_1 = q.In; q.In = tmp; }
        }
        if (!progressed) __builtin_ia32_pause();      // (every live group is inside its order-loop launch)
    }
    if (NG > 1) {                                    // back onto the caller's stream
        HIPCHK(hipEventRecord(h->ev_join, h->stream2));
        HIPCHK(hipStreamWaitEvent(s, h->ev_join, 0));
    }
    prof_break(h);
    launch_finalize(s, B, make_conv(h, tol), h->order_budget, d_n_orders_out, d_status_out);
    HIPCHK(hipGetLastError());
    h->last_max_orders = used_order_loop ? -1 : n_max;      // (orders run inside an order-loop launch: read back with the counts)
    h->last_sum_orders = -1;
    return 0;
}

int sosrt_solve(sosrt_t* h, int B, const double* tau, const double* P0_atm, const double* P0_aer, double tol,
                const double* I1_in, double* I_out, double* I_saved_out, int* n_orders_out, int* status_out) {
    if (int e = check_ready(h, B, true)) return e;
    if (!tau) return fail(SOSRT_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    h->resident = false;
    const size_t LD = (size_t)h->L * h->D, n = (size_t)B * LD;
    HIPCHK(hipMemcpyAsync(h->d_tau, tau, (size_t)B * h->L * sizeof(double), hipMemcpyHostToDevice, s));
    if (P0_atm) HIPCHK(hipMemcpyAsync(h->d_P0a, P0_atm, (size_t)B * h->D * sizeof(double), hipMemcpyHostToDevice, s));
    if (P0_aer) HIPCHK(hipMemcpyAsync(h->d_P0r, P0_aer, (size_t)B * h->D * sizeof(double), hipMemcpyHostToDevice, s));
    double* d_I1 = nullptr;
    double* d_saved = nullptr;
    int rc = 0;
    auto body = [&]() -> int {
        if (I1_in) {
            if (int e = dalloc(&d_I1, n)) return e;
            HIPCHK(hipMemcpyAsync(d_I1, I1_in, n * sizeof(double), hipMemcpyHostToDevice, s));
        }
        if (I_saved_out) {
            if (int e = dalloc(&d_saved, (size_t)B * h->saved_slots * LD)) return e;
            HIPCHK(hipMemsetAsync(d_saved, 0, (size_t)B * h->saved_slots * LD * sizeof(double), s));
        }
        if (int e = sosrt_solve_dev(h, B, h->d_tau, P0_atm ? h->d_P0a : nullptr, P0_aer ? h->d_P0r : nullptr, tol, d_I1,
                                    h->d_I, d_saved, nullptr, nullptr))
            return e;
        if (I_out) HIPCHK(hipMemcpyAsync(I_out, h->d_I, n * sizeof(double), hipMemcpyDeviceToHost, s));
        if (I_saved_out)
            HIPCHK(hipMemcpyAsync(I_saved_out, d_saved, (size_t)B * h->saved_slots * LD * sizeof(double), hipMemcpyDeviceToHost, s));
        std::vector<int> no(B);
        HIPCHK(hipMemcpyAsync(no.data(), h->d_norders, B * sizeof(int), hipMemcpyDeviceToHost, s));
        if (status_out) HIPCHK(hipMemcpyAsync(status_out, h->d_status, B * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        long long sum = 0;
        int mx = 1;
        for (int b = 0; b < B; ++b) { sum += no[b] - 1; mx = no[b] > mx ? no[b] : mx; if (n_orders_out) n_orders_out[b] = no[b]; }
        h->last_sum_orders = sum;
        if (h->last_max_orders < 0) h->last_max_orders = mx;
        h->resident = true; h->resident_B = B;
        return 0;
    };
    rc = body();
    if (d_I1) hipFree(d_I1);
    if (d_saved) hipFree(d_saved);
    return rc;
}

int sosrt_last_solve_stats(sosrt_t* h, int* max_orders_run, long long* sum_orders) {
    if (int e = need_gpu(h)) return e;
    if ((h->last_sum_orders < 0 || h->last_max_orders < 0) && h->B > 0) {
        std::vector<int> no(h->B);
        HIPCHK(hipMemcpyAsync(no.data(), h->d_norders, h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        long long sum = 0;
        int mx = 1;
        for (int v : no) { sum += v - 1; mx = v > mx ? v : mx; }
        h->last_sum_orders = sum;
        if (h->last_max_orders < 0) h->last_max_orders = mx;
    }
    if (max_orders_run) *max_orders_run = h->last_max_orders;
    if (sum_orders) *sum_orders = h->last_sum_orders;
    return 0;
}

int sosrt_fluxes(sosrt_t* h, int B, const double* tau, const double* I, int beam_norm, double* flux_down,
                 double* flux_up) {
    if (int e = check_ready(h, B, false)) return e;
    if (!tau || !I || !flux_down || !flux_up) return fail(SOSRT_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->device));
    h->resident = false;                     // d_tau and d_I are overwritten
    hipStream_t s = h->stream;
    const size_t n = (size_t)B * h->L * h->D, r = (size_t)B * h->L;
    HIPCHK(hipMemcpyAsync(h->d_tau, tau, r * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_I, I, n * sizeof(double), hipMemcpyHostToDevice, s));
    launch_prepare(s, h->g, B, h->geom, h->surface, scalars_of(h), h->d_tau, h->d_desc, h->d_rca, h->d_rcr);
    launch_fluxes(s, h->g, B, h->d_tau, h->d_I, h->d_desc, beam_norm, h->d_rca, h->d_rcr);   // row-sized scratch
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(flux_down, h->d_rca, r * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(flux_up, h->d_rcr, r * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// epilogue on a resident field
// ---------------------------------------------------------------------------------------------
int sosrt_epilogue_dev(sosrt_t* h, int B, const double* d_tau, const double* d_I, int beam_norm, const double* d_z_profile,
                       double* d_flux_down, double* d_flux_up, double* d_diffusivity, double* d_heating_rate,
                       double* d_net_toa) {
    if (int e = check_ready(h, B, false)) return e;
    if (!d_tau || !d_I) return fail(SOSRT_E_INVALID, "null argument");
    if (d_heating_rate && !d_z_profile) return fail(SOSRT_E_INVALID, "the heating rate needs z_profile");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    launch_prepare(s, h->g, B, h->geom, h->surface, scalars_of(h), d_tau, h->d_desc, h->d_rca, h->d_rcr);
    EpilogueOut out{d_flux_down, d_flux_up, d_diffusivity, d_heating_rate, d_net_toa};
    launch_epilogue(s, h->g, h->d_w, B, d_tau, d_I, h->d_desc, beam_norm, d_z_profile, out);
    HIPCHK(hipGetLastError());
    return 0;
}

int sosrt_epilogue(sosrt_t* h, int B, int beam_norm, const double* z_profile, double* flux_down, double* flux_up,
                   double* diffusivity, double* heating_rate, double* net_toa) {
    if (int e = check_ready(h, B, false)) return e;
    if (!h->resident) return fail(SOSRT_E_STATE, "no resident field: call sosrt_solve first (and nothing that overwrites it since)");
    if (B != h->resident_B) return fail(SOSRT_E_INVALID, "B=%d but the resident field is of %d columns", B, h->resident_B);
    if (heating_rate && !z_profile) return fail(SOSRT_E_INVALID, "the heating rate needs z_profile");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t r = (size_t)B * h->L;
    if (4 * r + B > field_elems(h)) return fail(SOSRT_E_INVALID, "batch too large for the scratch buffer");
    if (z_profile) HIPCHK(hipMemcpyAsync(h->d_z, z_profile, h->L * sizeof(double), hipMemcpyHostToDevice, s));
    double* o = h->d_Jn;                                     // scratch: the source function of the last order is dead
    if (int e = sosrt_epilogue_dev(h, B, h->d_tau, h->d_I, beam_norm, z_profile ? h->d_z : nullptr, flux_down ? o : nullptr,
                                   flux_up ? o + r : nullptr, diffusivity ? o + 2 * r : nullptr,
                                   heating_rate ? o + 3 * r : nullptr, net_toa ? o + 4 * r : nullptr))
        return e;
    if (flux_down) HIPCHK(hipMemcpyAsync(flux_down, o, r * sizeof(double), hipMemcpyDeviceToHost, s));
    if (flux_up) HIPCHK(hipMemcpyAsync(flux_up, o + r, r * sizeof(double), hipMemcpyDeviceToHost, s));
    if (diffusivity) HIPCHK(hipMemcpyAsync(diffusivity, o + 2 * r, r * sizeof(double), hipMemcpyDeviceToHost, s));
    if (heating_rate) HIPCHK(hipMemcpyAsync(heating_rate, o + 3 * r, r * sizeof(double), hipMemcpyDeviceToHost, s));
    if (net_toa) HIPCHK(hipMemcpyAsync(net_toa, o + 4 * r, B * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// phase functions on the device
// ---------------------------------------------------------------------------------------------
int sosrt_phase_table(sosrt_t* h, const double* tab_mu, const double* tab_p, int ntab) {
    if (int e = need_gpu(h)) return e;
    if (!tab_mu || !tab_p || ntab < 2) return fail(SOSRT_E_INVALID, "a table needs at least two points");
    for (int i = 1; i < ntab; ++i)
        if (!(tab_mu[i] > tab_mu[i - 1])) return fail(SOSRT_E_INVALID, "tab_mu must be strictly ascending (index %d)", i);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_tab) { hipFree(h->d_tab); h->d_tab = nullptr; h->ntab = 0; }
    if (int e = dalloc(&h->d_tab, 2 * (size_t)ntab)) return e;
    HIPCHK(hipMemcpy(h->d_tab, tab_mu, ntab * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_tab + ntab, tab_p, ntab * sizeof(double), hipMemcpyHostToDevice));
    h->ntab = ntab;
    return 0;
}

static int phase_check(sosrt_handle* h, int kind, double g) {
    if (int e = need_gpu(h)) return e;
    if (!h->have_grid) return fail(SOSRT_E_STATE, "sosrt_set_grid has not been called");
    if (kind < SOSRT_PHASE_ISO || kind > SOSRT_PHASE_TABLE) return fail(SOSRT_E_INVALID, "unknown phase-function kind %d", kind);
    if (kind == SOSRT_PHASE_TABLE && !h->d_tab) return fail(SOSRT_E_STATE, "sosrt_phase_table has not been called");
    if (kind == SOSRT_PHASE_HG && !(std::fabs(g) < 1)) return fail(SOSRT_E_INVALID, "|g| must be < 1 (got %g)", g);
    return 0;
}

int sosrt_phase_p0_dev(sosrt_t* h, int B, int kind, double g, const double* d_mu0, double* d_P0_out) {
    if (int e = phase_check(h, kind, g)) return e;
    if (B < 1 || !d_mu0 || !d_P0_out) return fail(SOSRT_E_INVALID, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    launch_phase_p0(h->stream, h->g, h->d_w, B, kind, g, h->d_tab, h->d_tab ? h->d_tab + h->ntab : nullptr, h->ntab, h->d_phi,
                    h->d_phi + kNPhi, kNPhi, d_mu0, d_P0_out);
    HIPCHK(hipGetLastError());
    return 0;
}

int sosrt_phase_p0(sosrt_t* h, int B, int kind, double g, const double* mu0, double* P0_out) {
    if (int e = phase_check(h, kind, g)) return e;
    if (B < 1 || B > h->max_batch || !mu0 || !P0_out) return fail(SOSRT_E_INVALID, "bad argument (B must be 1..max_batch)");
    for (int b = 0; b < B; ++b)
        if (!(mu0[b] > 0 && mu0[b] <= 1)) return fail(SOSRT_E_INVALID, "column %d: mu0 must be in (0, 1]", b);
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->d_ratio, mu0, B * sizeof(double), hipMemcpyHostToDevice, s));
    if (int e = sosrt_phase_p0_dev(h, B, kind, g, h->d_ratio, h->d_P0a)) return e;
    HIPCHK(hipMemcpyAsync(P0_out, h->d_P0a, (size_t)B * h->D * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int sosrt_phase_matrix(sosrt_t* h, int kind, double g, double* P_out) {
    if (int e = phase_check(h, kind, g)) return e;
    if (!P_out) return fail(SOSRT_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    double* dP = nullptr;
    const size_t n = (size_t)h->D * h->D;
    if (int e = dalloc(&dP, n)) return e;
    int rc = 0;
    auto body = [&]() -> int {
        launch_phase_matrix(s, h->g, h->d_w, kind, g, h->d_tab, h->d_tab ? h->d_tab + h->ntab : nullptr, h->ntab, h->d_phi,
                            h->d_phi + kNPhi, kNPhi, dP);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(P_out, dP, n * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    };
    rc = body();
    hipFree(dP);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU gather over RCCL
// ---------------------------------------------------------------------------------------------
#define NCCLCHK(x)                                                                                          \
    do {                                                                                                    \
        const int r_ = (x);                                                                                 \
        if (r_ != 0) return fail(SOSRT_E_HIP, "%s failed: %s", #x, rccl().GetErrorString(r_));              \
    } while (0)

int sosrt_comm_unique_id(void* id_out) {
    if (!id_out) return fail(SOSRT_E_INVALID, "null argument");
    if (const char* e = rccl().load()) return fail(SOSRT_E_STATE, "%s", e);
    Rccl::UniqueId id;
    NCCLCHK(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int sosrt_comm_init(sosrt_t* h, int rank, int world, const void* unique_id) {
    if (int e = need_gpu(h)) return e;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(SOSRT_E_INVALID, "bad rank / world / id");
    if (h->comm) return fail(SOSRT_E_STATE, "the handle already has a communicator");
    if (const char* e = rccl().load()) return fail(SOSRT_E_STATE, "%s", e);
    HIPCHK(hipSetDevice(h->device));
    Rccl::UniqueId id;
    memcpy(&id, unique_id, sizeof id);
    NCCLCHK(rccl().CommInitRank(&h->comm, world, id, rank));
    h->comm_rank = rank; h->comm_world = world;
    return 0;
}

int sosrt_gather(sosrt_t* h, int root, const long long* counts, const double* d_send, double* d_recv) {
    if (int e = need_gpu(h)) return e;
    if (!h->comm) return fail(SOSRT_E_STATE, "sosrt_comm_init has not been called");
    if (!counts || root < 0 || root >= h->comm_world) return fail(SOSRT_E_INVALID, "bad root / counts");
    for (int r = 0; r < h->comm_world; ++r)
        if (counts[r] < 0) return fail(SOSRT_E_INVALID, "counts[%d] < 0", r);
    const int me = h->comm_rank;
    if (counts[me] > 0 && !d_send) return fail(SOSRT_E_INVALID, "d_send is null");
    if (me == root && !d_recv) return fail(SOSRT_E_INVALID, "d_recv is null on the root");
    HIPCHK(hipSetDevice(h->device));
    const int kF64 = 8;                                      // ncclFloat64
    // every exit below goes through GroupEnd: a group left open would swallow the communicator's next calls
    int rc = 0, nrc = 0;
    hipError_t hrc = hipSuccess;
    NCCLCHK(rccl().GroupStart());
    if (me == root) {
        size_t off = 0;
        for (int r = 0; r < h->comm_world && !nrc && hrc == hipSuccess; ++r) {
            if (counts[r] > 0) {
                if (r == me) {
                    if (d_recv + off != d_send)
                        hrc = hipMemcpyAsync(d_recv + off, d_send, (size_t)counts[r] * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
                } else {
                    nrc = rccl().Recv(d_recv + off, (size_t)counts[r], kF64, r, h->comm, h->stream);
                }
            }
            off += (size_t)counts[r];
        }
    } else if (counts[me] > 0) {
        nrc = rccl().Send(d_send, (size_t)counts[me], kF64, root, h->comm, h->stream);
    }
    const int erc = rccl().GroupEnd();
    if (hrc != hipSuccess) rc = fail(SOSRT_E_HIP, "sosrt_gather: copy of the root's own block failed: %s", hipGetErrorString(hrc));
    else if (nrc) rc = fail(SOSRT_E_HIP, "sosrt_gather: ncclSend/ncclRecv failed: %s", rccl().GetErrorString(nrc));
    else if (erc) rc = fail(SOSRT_E_HIP, "sosrt_gather: ncclGroupEnd failed: %s", rccl().GetErrorString(erc));
    return rc;
}

int sosrt_comm_destroy(sosrt_t* h) {
    if (int e = need_gpu(h)) return e;
    if (h->comm) {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        NCCLCHK(rccl().CommDestroy(h->comm));
        h->comm = nullptr; h->comm_rank = -1; h->comm_world = 0;
    }
    return 0;
}

int sosrt_limit_mu_down(sosrt_t* h, int R, int idx, const double* rows, double* out) {
    if (int e = need_gpu(h)) return e;
    if (!h->have_grid) return fail(SOSRT_E_STATE, "sosrt_set_grid has not been called");
    if (R < 1 || !rows || !out) return fail(SOSRT_E_INVALID, "bad argument");
    int table = -1;
    for (int b = 0; b < 4; ++b)
        if (h->plan.fix[b].idx == idx) table = b;
    if (table < 0) return fail(SOSRT_E_INVALID, "idx=%d is not one of the reference's int(c*nb_angles) values for nb_angles=%d", idx, h->N);
    if (idx == 0) return 0;
    if ((size_t)R * h->N > field_elems(h)) return fail(SOSRT_E_INVALID, "too many rows");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->d_Jn, rows, (size_t)R * h->N * sizeof(double), hipMemcpyHostToDevice, s));
    launch_limit_rows(s, h->g, R, table, h->d_Jn, h->d_InB);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, h->d_InB, (size_t)R * idx * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int sosrt_asymptotic_down(sosrt_t* h, int R, int stride, const int* len, const double* J, const double* tau,
                          const double* tau_t, const double* mu, double* out) {
    if (int e = need_gpu(h)) return e;
    if (R < 1 || stride < 1 || !len || !J || !tau || !tau_t || !mu || !out) return fail(SOSRT_E_INVALID, "bad argument");
    const size_t n = (size_t)R * stride;
    if (2 * n + 4 * (size_t)R > field_elems(h)) return fail(SOSRT_E_INVALID, "too many rows");
    for (int r = 0; r < R; ++r)
        if (len[r] < 0 || len[r] > stride) return fail(SOSRT_E_INVALID, "len[%d] out of range", r);
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    double* dJ = h->d_Jn; double* dT = h->d_Jn + n;
    double* dtt = h->d_InB; double* dmu = h->d_InB + R; double* dout = h->d_InB + 2 * (size_t)R;
    int* dlen = (int*)(h->d_InA);
    HIPCHK(hipMemcpyAsync(dJ, J, n * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dT, tau, n * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dtt, tau_t, R * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dmu, mu, R * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dlen, len, R * sizeof(int), hipMemcpyHostToDevice, s));
    launch_asymptotic(s, R, stride, dlen, dJ, dT, dtt, dmu, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout, R * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// plan introspection (host only)
// ---------------------------------------------------------------------------------------------
int sosrt_plan_weights(sosrt_t* h, double* w_out) {
    if (!h || !w_out) return fail(SOSRT_E_INVALID, "null argument");
    if (!h->have_grid) return fail(SOSRT_E_STATE, "sosrt_set_grid has not been called");
    memcpy(w_out, h->plan.w.data(), h->D * sizeof(double));
    return 0;
}

int sosrt_plan_fold(sosrt_t* h, int which, double* W_out) {
    if (!h || !W_out) return fail(SOSRT_E_INVALID, "null argument");
    if (!h->have_phase) return fail(SOSRT_E_STATE, "sosrt_set_phase has not been called");
    const std::vector<double>& W = which ? h->Wr_h : h->Wa_h;
    if (W.empty()) return fail(SOSRT_E_STATE, "that phase matrix was not set");
    memcpy(W_out, W.data(), W.size() * sizeof(double));
    return 0;
}

int sosrt_plan_fix_table(sosrt_t* h, int idx, int* s0, int* ns, double* C_out) {
    if (!h || !s0 || !ns) return fail(SOSRT_E_INVALID, "null argument");
    if (!h->have_grid) return fail(SOSRT_E_STATE, "sosrt_set_grid has not been called");
    if (idx < 0 || idx > kFixMaxIdx || idx + 5 > h->N) return fail(SOSRT_E_INVALID, "idx out of range");
    FixTable t = h->plan.make_fix_table(idx);
    *s0 = t.s0; *ns = t.ns;
    if (C_out && !t.C.empty()) memcpy(C_out, t.C.data(), t.C.size() * sizeof(double));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// diagnostics: cycle stamps of k_transport_fast ([B][2 waves][8] clock64 values; pass NULL to stop)
// ---------------------------------------------------------------------------------------------
int sosrt_debug_stamps(sosrt_t* h, unsigned long long* d_stamps) {
    if (int e = need_gpu(h)) return e;
    sosrt::g_transport_stamps = d_stamps;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// machine peaks
// ---------------------------------------------------------------------------------------------
int sosrt_microbench(sosrt_t* h, int which, double* result) {
    if (int e = need_gpu(h)) return e;
    if (!result || which < 0 || (which > 2 && which < 10) || which > 49) return fail(SOSRT_E_INVALID, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const size_t n = (size_t)1 << 27;                 // 1 GiB of doubles per buffer for the copy test
    double *a = nullptr, *b = nullptr;
    int rc = 0;
    auto body = [&]() -> int {
        const size_t na = which == 1 ? n : 16;
        if (int e = dalloc(&a, na)) return e;
        if (which == 1) { if (int e = dalloc(&b, na)) return e; HIPCHK(hipMemsetAsync(a, 0, na * sizeof(double), s)); }
        const int iters = 20000;
        double best = 0;
        for (int rep = 0; rep < 4; ++rep) {
            HIPCHK(hipEventRecord(e0, s));
            launch_bench(s, which, a, b, na, iters);
            HIPCHK(hipEventRecord(e1, s));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            double v;
            if (which >= 10) {
                const int code = which - 10, nacc = 2 << (code / 10), bpc = (code % 10) ? (code % 10) : 1;
                v = 256.0 * bpc * 4 * (double)iters * nacc * 2048.0 / (ms * 1e-3) / 1e12;
            } else if (which == 0) v = 2048.0 * 4 * (double)iters * 8 * 2048.0 / (ms * 1e-3) / 1e12;        // TFLOP/s
            else if (which == 1) v = 2.0 * na * sizeof(double) / (ms * 1e-3) / 1e9;                  // GB/s read+write
            else v = 2048.0 * 256 * (double)iters * 16 * 2.0 / (ms * 1e-3) / 1e12;                   // TFLOP/s
            if (rep > 0 && v > best) best = v;
        }
        *result = best;
        return 0;
    };
    rc = body();
    if (a) hipFree(a);
    if (b) hipFree(b);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------------------------
int sosrt_profile_enable(sosrt_t* h, int on) {
    if (int e = need_gpu(h)) return e;
    for (Prof& p : h->prof) {
        if (on && p.ev.empty()) {
            HIPCHK(hipSetDevice(h->device));
            p.ev.resize(2 * kProfPool);
            p.kind.assign(kProfPool, -1); p.first.assign(kProfPool, 0); p.last.assign(kProfPool, 0);
            // timing markers only: no system-scope fence (cache write-back / invalidate) at each of them
            for (auto& e : p.ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
        }
        p.on = on != 0;
    }
    return 0;
}

int sosrt_profile_reset(sosrt_t* h) {
    if (int e = need_gpu(h)) return e;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->stream2) HIPCHK(hipStreamSynchronize(h->stream2));
    for (Prof& p : h->prof) { p.used = 0; p.nint = 0; p.adjacent = -1; p.open = -1; }
    return 0;
}

int sosrt_profile_get(sosrt_t* h, int kernel, double* total_ms, long long* launches, double* work) {
    if (int e = need_gpu(h)) return e;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->stream2) HIPCHK(hipStreamSynchronize(h->stream2));
    double tot = 0;
    long long cnt = 0;
    for (Prof& p : h->prof)
        for (size_t i = 0; i < p.nint; ++i) {
            if (p.kind[i] != kernel) continue;
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, p.ev[p.first[i]], p.ev[p.last[i]]));
            tot += ms;
            ++cnt;
        }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = cnt;
    if (work) *work = 0;
    return 0;
}

}  // extern "C"
