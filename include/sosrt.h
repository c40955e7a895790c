/*
 * sosrt.h -- C ABI of libsosrt.so, the MI355X (gfx950) implementation of the
 * Successive-Orders-of-Scattering hot path of
 * Guillaume-SOULIER/SOS-Radiative-Transfer (reference snapshot 2025-09-05).
 *
 * The reference has no FFI layer: its boundary for this path is a set of plain
 * Python functions on caller-owned float64 NumPy arrays.  Each entry point
 * below names the reference interface it replaces (file:line, relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SOSRT_E_* code on error;
 *     sosrt_last_error() returns a message for the calling thread.
 *   - all floating-point data is IEEE double, row-major, index order
 *     [column][order][layer t][direction m], m fastest; directions
 *     are ordered mu = -1..0 (m = 0..N-1, downward) then 0..+1 (m = N..2N-1).
 *   - "host" entry points take host pointers and copy in/out; "_dev" entry
 *     points take device pointers (hipMalloc'd, resident) and only enqueue
 *     work on the handle's stream.
 *   - a handle is bound to one device and one stream and is not thread-safe;
 *     distinct handles may be used concurrently.
 *   - the caller owns every buffer passed in.
 */
#ifndef SOSRT_H
#define SOSRT_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sosrt_handle sosrt_t;

#define SOSRT_OK              0
#define SOSRT_E_INVALID      -1   /* bad argument / shape (Python layer raises ValueError)   */
#define SOSRT_E_HIP          -2   /* HIP runtime error                                        */
#define SOSRT_E_STATE        -3   /* call order (grid / phase / columns not set)              */
#define SOSRT_E_NOMEM        -4

/* per-column status written by transport / solve */
#define SOSRT_COL_OK           0
#define SOSRT_COL_INDEXERROR   1  /* upward mu->0+ search ran off the grid: the reference raises IndexError (spec:404, I1_In:103) */
#define SOSRT_COL_MAXORDERS    2  /* not converged within max_orders                                                          */
#define SOSRT_COL_INTERNAL     3  /* the transport kernel gave up waiting on itself (never expected; the column's field is unusable) */

/* geometry of a column */
#define SOSRT_GEOM_THREE_ZONE  0  /* above / inside / below the aerosol slab: SOS_Aer_main_specular.py:104-458 */
#define SOSRT_GEOM_SINGLE_SLAB 1  /* one homogeneous slab, black surface:      SOS_Aer_I1_In.py:13-130           */

/* surface model for orders n >= 2 */
#define SOSRT_SURFACE_NONE       0  /* single slab (I1_In:86-98)                                       */
#define SOSRT_SURFACE_SPECULAR   1  /* spec:397/399                                                    */
#define SOSRT_SURFACE_LAMBERTIAN 2  /* lam:399/401, coded sign (SURVEY hazard H2): the reflected radiance comes
                                       out negative, because the code integrates over a descending mu array      */
#define SOSRT_SURFACE_LAMBERTIAN_README 3  /* the same term with the sign of README.md:215 (positive); not what the
                                              reference's file computes -- non-default, parity unpinned            */

const char* sosrt_last_error(void);
/* 100 * major + minor of the ABI this library was built from.  101 (round 4): sosrt_set_stream(h, NULL) names the legacy default
 * stream (100 read NULL as "the handle's own stream"); the per-handle launch plan (sosrt_plan_launch) and the order loop that
 * runs several orders per launch (sosrt_set_order_loop) were added (sosrt_plan_launch may answer SOSRT_PLAN_GEMM_LIVE16_REGS since the
 * second half of that round).  A binding checks sosrt_version() >= the SOSRT_VERSION it
 * was written against. */
#define SOSRT_VERSION 101
int sosrt_version(void);

/* ---- handle ------------------------------------------------------------------------------- */
/* L = nb_layers, N = nb_angles per hemisphere (spec:33,57).  Buffers are sized for max_batch
 * columns; max_orders bounds the order loop of spec:309.  device < 0 makes a host-only handle
 * (plan queries only, no GPU is touched).  4 <= N <= 1024; 2 <= L, and three values per layer of a column must fit the
 * 64 KiB of LDS the first-order kernel asks for (L <= 2686 at N = 128; the reference ships L = 800): SOSRT_E_INVALID
 * otherwise, with the largest L in sosrt_last_error(). */
int sosrt_create(int device, int L, int N, int max_batch, int max_orders, sosrt_t** out);
int sosrt_destroy(sosrt_t* h);
/* Streams.  A new handle enqueues on a stream of its own, created as a BLOCKING stream (hipStreamDefault): it orders against
 * the legacy default stream, so a caller that fills its buffers on the default stream (handle NULL -- torch's default stream
 * is that one) and then calls a `_dev` entry point gets the order it wrote, without an explicit synchronise.
 * sosrt_set_stream runs the handle on the caller's hipStream_t instead; NULL names the legacy default stream itself, as it
 * does everywhere in HIP (round 2 read NULL as "the handle's own stream", and that stream was non-blocking: a torch caller
 * passing torch.cuda.current_stream().cuda_stream == 0 got work that raced its own fills -- gpurun_out/split_full.txt).
 * sosrt_use_own_stream goes back to the handle's stream. */
int sosrt_set_stream(sosrt_t* h, void* hip_stream);
int sosrt_use_own_stream(sosrt_t* h);
int sosrt_synchronize(sosrt_t* h);
/* I_saved_out of the solves that follow holds `slots` orders per column ([B][slots][L][2N], 1 <= slots <=
 * max_orders; orders beyond are computed but not stored).  Default: max_orders.  The reference's list
 * I_saved (spec:304-305,458) has exactly n entries: solve once without it to learn n, then once with slots = max n. */
int sosrt_set_saved_orders(sosrt_t* h, int slots);
/* The solves that follow run at most `max_orders` orders (1 <= max_orders <= the max_orders of sosrt_create, which is the default):
 * a column still iterating then has status SOSRT_COL_MAXORDERS.  (The reference's loop, spec:309, has no bound.)  Lets one handle
 * serve callers with different budgets -- the Python layer's handle cache does. */
int sosrt_set_order_budget(sosrt_t* h, int max_orders);

/* ---- per-sweep setup ------------------------------------------------------------------------ */
/* direction grid mu[2N] (spec:59-61).  Builds the trapezoid weights of np.trapz(.., mu) used by
 * Jn (I1_In:73), the small-mu lane list (gva:5-7) and the extrapolation tables that replace
 * improved_limit_mu_down (In_limit:113-141). */
int sosrt_set_grid(sosrt_t* h, const double* mu);
/* phase matrices P(mu, mu') [2N x 2N] (outputs of phase_func, phase:12); P_aer may be NULL for the
 * single-slab geometry.  Folded on the host into W[k][m] = w_k P[m][2N-1-k] (I1_In:73, spec:321). */
int sosrt_set_phase(sosrt_t* h, const double* P_atm, const double* P_aer);

/* First order of the solve.  CODED (default): spec:104-292 -- what both mains of the reference compute, with the specularly
 * reflected beam (SOS_Aer_main_lambertian.py's first-order blocks are the same formulas; its lines 274-276 crash).
 * README: the Lambertian first order of the reference's README.md:126-171 -- direct beam + the beam reflected isotropically by
 * the ground (int_0^1 mu'/(mu'-mu) ... dmu' by the trapezoid rule on the upward directions, the removable singularity at
 * mu' = mu taken analytically) + isotropic reflection of the downward first order.  PARITY UNPINNED: no runnable reference
 * code exists for it.  Meant for SOSRT_SURFACE_LAMBERTIAN_README; three-zone geometry only. */
#define SOSRT_FIRST_ORDER_CODED 0
#define SOSRT_FIRST_ORDER_README 1
int sosrt_set_first_order(sosrt_t* h, int mode);

/* arithmetic of the source-function contraction (BASELINE configs[4]: "fp64 -> fp32 mixed with tolerance study").
 * SOSRT_CONTRACT_F64 (default): v_mfma_f64_16x16x4_f64 -- the only mode that meets the 1e-10 parity bar.
 * SOSRT_CONTRACT_F32: operands rounded to float, v_mfma_f32_16x16x4_f32 with a float accumulator; transport, running
 * total and convergence test stay fp64.  About 3e-7 of the field maximum away from the fp64 result (measured on the
 * device: profiles/r02_mixed_precision_gpu.txt) -- an opt-in for callers with that tolerance, never the default.
 * Needs at most 32 distinct slab coefficient pairs in the batch.
 *
 * Within SOSRT_CONTRACT_F64 the library uses the flip symmetry of the folded matrices when they have it: every phase
 * function of the scattering angle on a grid with mu[2N-1-k] = -mu[k] gives W[2N-1-k][2N-1-m] = W[k][m], and then
 *   Jn[m] +- Jn[2N-1-m] = sum_{k<N} (In_1[k] +- In_1[2N-1-k]) (W[k][m] +- W[2N-1-k][m])
 * -- two N x N products instead of one 2N x 2N, half the flops, same v_mfma_f64 arithmetic.  sosrt_set_phase measures
 * max |W[k][m] - W[2N-1-k][2N-1-m]| / max |W| (sosrt_phase_asymmetry); at or below SOSRT_SYMMETRY_TOL (the rounding of
 * the phase-matrix builders: 1e-14 for the reference's) the symmetric form is used on the symmetric part of W, so Jn moves
 * by at most that fraction of max |W| sum |In_1| -- four orders of magnitude inside the 1e-10 parity bar; above it (any
 * matrix without the symmetry) the full product runs.  SOSRT_CONTRACT_F64_FULL forces the full product. */
#define SOSRT_CONTRACT_F64 0
#define SOSRT_CONTRACT_F32 1
#define SOSRT_CONTRACT_F64_FULL 2
#define SOSRT_SYMMETRY_TOL 1e-12
int sosrt_set_contraction(sosrt_t* h, int mode);

/* The order loop of spec:309-458 runs an order as two launches (source function, transport) and the host learns the live count
 * between them.  Once few columns of a batch (or column group) are left -- their transport workgroups at most half the CUs --
 * the REMAINING orders run in ONE launch whose workgroups keep their roles (csrc/order_loop.hip): the chunk-parallel transport
 * of each live column and, on all other workgroups, the tiles of the live columns' source functions, tied by per-column
 * counters in device memory; the host only waits for the launch's report.  Same arithmetic, same bits.  mode 0 (default): never --
 * measured on MI355X the launch is bit-identical but SLOWER than the two launches per order it replaces (the dependency chain
 * sweep -> source-function tile -> sweep is the same; DESIGN section 5 item 9, profiles/r04_order_loop_ab_v0.txt); mode 1: where
 * the launch plan says so (sosrt_plan_launch), kept for the measurements and as the scaffold of a finer-grained pipeline; mode 2
 * (tests): as 1 with a grid of twice the device's CUs, which can never be resident -- every launch is refused and handed back.  The launch checks its own residency first and hands the orders back
 * to the two-launch loop when another process's kernels keep its grid from being resident (sosrt_order_loop_stats: launches of
 * the last solve, how many of them were refused that way, and the (column, order) pairs that ran inside them). */
int sosrt_set_order_loop(sosrt_t* h, int mode);
int sosrt_order_loop_stats(sosrt_t* h, int* launches, int* refused, long long* column_orders /* nullable: (column, order) pairs run inside them; synchronises */);
/* asymmetry of the folded matrices of the last sosrt_set_phase (see above); *uses_symmetry: what the next solve will do */
int sosrt_phase_asymmetry(sosrt_t* h, double* asymmetry, int* uses_symmetry);

/* per-column scalars (the locals of spec:23-53).  Arrays have B entries.
 *   THREE_ZONE : idx_up, idx_down (spec:40), mu0, grd_alb, alb_atm, alb_aer, dtau_atm, dtau_aer
 *                (spec:50-53), tauStar_tot (spec:36).
 *   SINGLE_SLAB: idx_* ignored (may be NULL); alb_atm = alb, tauStar_tot = tauStar of
 *                I1_NumInt / In_NumInt (I1_In:13,77); grd_alb, alb_aer, dtau_* ignored (may be NULL). */
int sosrt_set_columns(sosrt_t* h, int B, int geometry, int surface,
                      const int* idx_up, const int* idx_down,
                      const double* mu0, const double* grd_alb,
                      const double* alb_atm, const double* alb_aer,
                      const double* dtau_atm, const double* dtau_aer,
                      const double* tauStar_tot);

/* The same with a caller's ZONE TABLE instead of one slab: zones top to bottom, zone z of column b starts at
 * row zone_r0[b][z] (zone_r0[b][0] = 0, ascending) and ends before the next one; zone_mix[b][z] = 1 marks an aerosol zone
 * with single-scattering albedo zone_alb_aer[b][z] and optical-depth step zone_dtau_aer[b][z] (the dtau_aer of spec:52,
 * i.e. the slab's aerosol optical depth / its number of rows), 0 a clear zone (its two aerosol entries are ignored).
 * Aerosol zones are separated and bounded by clear ones; at most SOSRT_MAX_ZONES zones (four slabs).  Arrays are
 * [B][nzmax] row-major; nz[b] <= nzmax zones are read for column b.  Every formula of the path is evaluated per zone
 * exactly as the reference writes it for its three (spec:113-449); the extrapolation bucket (spec:342,361,380) of a clear
 * zone below a slab follows that slab's last row.  (clear, slab, clear) columns give bit-for-bit the results of
 * sosrt_set_columns.  The optical-depth grid tau of the solve must be consistent with the table (taup:21-27 per slab). */
#define SOSRT_MAX_ZONES 8
int sosrt_set_columns_zones(sosrt_t* h, int B, int surface, int nzmax, const int* nz, const int* zone_r0, const int* zone_mix,
                            const double* mu0, const double* grd_alb, const double* alb_atm, const double* dtau_atm,
                            const double* zone_alb_aer, const double* zone_dtau_aer, const double* tauStar_tot);

/* ---- step level (host pointers): parity surface of SOS_Aer_I1_In.py ------------------------- */
/* I1_NumInt (I1_In:13) / three-zone first order (spec:104-292).
 * tau [B][L], P0_atm / P0_aer [B][2N] (P0_aer may be NULL for SINGLE_SLAB), I1 out [B][L][2N] */
int sosrt_first_order(sosrt_t* h, int B, const double* tau, const double* P0_atm, const double* P0_aer,
                      double* I1_out);
/* Jn_NumInt (I1_In:62) / spec:314-323.  In_1 [B][L][2N] -> Jn [B][L][2N] */
int sosrt_source(sosrt_t* h, int B, const double* In_1, double* Jn_out);
/* In_NumInt (I1_In:77) / spec:326-449 (+ lam:399/401).  Jn [B][L][2N] -> In [B][L][2N];
 * status_out [B] receives SOSRT_COL_* (may be NULL). */
int sosrt_transport(sosrt_t* h, int B, const double* tau, const double* Jn, double* In_out, int* status_out);

/* ---- column level: the order loop of spec:301-458 ------------------------------------------- */
/* Iterates I1 -> [Jn -> In] until max(In/I at TOA-up, In/I at surface-down) < tol (spec:309) per
 * column.  Outputs: I_out [B][L][2N]; I_saved_out [B][slots][L][2N] or NULL (spec:304-305,458; slots =
 * max_orders unless sosrt_set_saved_orders was called);
 * n_orders_out [B] (the final n of spec:307-310); status_out [B] or NULL; I1_in (nullable,
 * [B][L][2N]) replaces the computed first order (used to pin the Lambertian n>=2 path).
 * I_out may be NULL: the field then stays on the device for sosrt_epilogue. */
int sosrt_solve(sosrt_t* h, int B, const double* tau, const double* P0_atm, const double* P0_aer,
                double tol, const double* I1_in,
                double* I_out, double* I_saved_out, int* n_orders_out, int* status_out);
/* same with device pointers; asynchronous on the handle's stream except for the convergence
 * polls.  n_orders_out / status_out are device int arrays (nullable). */
int sosrt_solve_dev(sosrt_t* h, int B, const double* d_tau, const double* d_P0_atm, const double* d_P0_aer,
                    double tol, const double* d_I1_in,
                    double* d_I_out, double* d_I_saved_out, int* d_n_orders_out, int* d_status_out);
/* number of order iterations (max over the batch) and sum over columns of the last solve */
int sosrt_last_solve_stats(sosrt_t* h, int* max_orders_run, long long* sum_orders);

/* ---- fused epilogue (graphe:157-158, crit:377-382): fluxes from a radiance field ------------ */
/* flux_down/up [B][L]; beam_norm 0: F0/(4 pi) (crit:380), 1: F0 (graphe:157). host pointers. */
int sosrt_fluxes(sosrt_t* h, int B, const double* tau, const double* I, int beam_norm,
                 double* flux_down, double* flux_up);

/* ---- epilogue on a RESIDENT field: what the reference's callers consume (a few kB per column instead of the
 * 0.4-1.6 MB field).  flux_down / flux_up as sosrt_fluxes (graphe:157-158, crit:380-381); diffusivity
 * -trapz(I mu, mu) / trapz(I, mu) per level (graphe:10); heating_rate per level (graphe:74-91, F0/(4 pi) beam
 * terms, last level copied, the two levels at the slab boundaries overwritten as in 'erase_pics'; needs
 * z_profile[L], the altitude grid np.linspace(z0, 0, L) of spec:39); net_toa [B] = -flux_down[0] - flux_up[0]
 * with the F0/(4 pi) beam terms (crit:382).  The net flux of graphe:41 is flux_down + flux_up with beam_norm 1.
 * Every output is nullable and skipped when NULL.
 * _dev: all pointers are device pointers (d_z_profile [L]); enqueued on the handle's stream.
 * sosrt_epilogue: works on the field and the optical depths the last sosrt_solve left in the handle (sosrt_solve
 * accepts I_out == NULL for that), host z_profile and host outputs. */
int sosrt_epilogue_dev(sosrt_t* h, int B, const double* d_tau, const double* d_I, int beam_norm,
                       const double* d_z_profile, double* d_flux_down, double* d_flux_up, double* d_diffusivity,
                       double* d_heating_rate, double* d_net_toa);
int sosrt_epilogue(sosrt_t* h, int B, int beam_norm, const double* z_profile, double* flux_down, double* flux_up,
                   double* diffusivity, double* heating_rate, double* net_toa);

/* ---- inputs of the path built on the device: azimuth-averaged phase functions (phase:68-292) --------------- */
#define SOSRT_PHASE_ISO       0   /* phase:68  isotropic                                                      */
#define SOSRT_PHASE_RAYLEIGH  1   /* phase:79  rayleigh                                                       */
#define SOSRT_PHASE_HG        2   /* phase:141 henyey_greenstein, asymmetry g                                 */
#define SOSRT_PHASE_TABLE     3   /* phase:238 fwc: a tabulated p(cos Theta), linear interpolation phase:198  */
/* table of SOSRT_PHASE_TABLE (host arrays, tab_mu ascending; fwc:3,173 is the reference's table) */
int sosrt_phase_table(sosrt_t* h, const double* tab_mu, const double* tab_p, int ntab);
/* P0(mu, mu0[b]) for B columns (phase:86-103): 25-point azimuth trapezoid, normalised to trapz(P0, mu) = 2.
 * _dev: d_mu0 [B] and d_P0_out [B][2N] are device pointers (a mu0 sweep builds its P0 where the solve reads it). */
int sosrt_phase_p0_dev(sosrt_t* h, int B, int kind, double g, const double* d_mu0, double* d_P0_out);
int sosrt_phase_p0(sosrt_t* h, int B, int kind, double g, const double* mu0, double* P0_out);
/* P(mu, mu') [2N][2N] row-major with the column normalisation trapz(P[:, n], mu) = 4 (phase:107-131); host output */
int sosrt_phase_matrix(sosrt_t* h, int kind, double g, double* P_out);

/* ---- multi-GPU: one process per GPU, columns sharded, ONE collective at the end ------------------
 * Nothing in SOS_Aer_main_specular.py:104-458 couples columns, so the order loop never communicates; these entry
 * points only assemble the results of the ranks on `root` over RCCL (xGMI inside a node).  RCCL is bound at run time
 * (dlopen): the library has no link-time dependency on it.
 *   sosrt_comm_unique_id  rank 0 fills id_out[128] (ncclGetUniqueId); the caller hands it to the other ranks by any
 *                         means (file, MPI, torch.distributed store, environment).
 *   sosrt_comm_init       every rank, same id: ncclCommInitRank on the handle's device.
 *   sosrt_gather          every rank: counts[world] doubles per rank (ragged shards allowed, counts[r] may be 0),
 *                         d_send = this rank's counts[rank] doubles (device), d_recv on root = the ranks' blocks one
 *                         after the other (device, sum of counts; ignored elsewhere).  Enqueued on the handle's stream
 *                         as ncclSend / ncclRecv pairs in one group -- a gather over the point-to-point links, each
 *                         sender on its own link to the root.
 *   sosrt_comm_destroy    ncclCommDestroy. */
int sosrt_comm_unique_id(void* id_out);
int sosrt_comm_init(sosrt_t* h, int rank, int world, const void* unique_id);
int sosrt_gather(sosrt_t* h, int root, const long long* counts, const double* d_send, double* d_recv);
int sosrt_comm_destroy(sosrt_t* h);

/* ---- helper level (In_limit:70,113) on device, host pointers -------------------------------- */
/* rows [R][N] of downward radiances; returns the idx rewritten values per row: out [R][idx] with
 * out[r][i] = improved_limit_mu_down(rows[r], mu[:N], N, idx, i). */
int sosrt_limit_mu_down(sosrt_t* h, int R, int idx, const double* rows, double* out);
/* out[r] = improved_asymptotic_downward_radiance(J[r][:len[r]], tau[r][:len[r]], tau_t[r], mu[r]);
 * J, tau are [R][stride]. */
int sosrt_asymptotic_down(sosrt_t* h, int R, int stride, const int* len, const double* J, const double* tau,
                          const double* tau_t, const double* mu, double* out);

/* ---- plan introspection (host only, no GPU needed) ------------------------------------------ */
int sosrt_plan_weights(sosrt_t* h, double* w_out /*2N*/);
int sosrt_plan_fold(sosrt_t* h, int which /*0 atm, 1 aer*/, double* W_out /*2N x 2N, W[k][m]*/);
/* a4b table for a given rewritten-angle count idx: s0 (first source lane), ns (sources),
 * C_out [idx][ns] (ns <= 5). */
int sosrt_plan_fix_table(sosrt_t* h, int idx, int* s0, int* ns, double* C_out);
int sosrt_plan_fix_count(double tau_ref, int N);  /* I1_In:124-127 */
/* The kernels an order of the order loop launches, as sosrt_solve_dev decides it (one function of the handle's shape and knobs;
 * needs sosrt_set_grid only): a batch of `batch` columns with up to `zones` zones each (3: the reference's clear / slab / clear)
 * over `surface`, of whose first column group `live` columns are still live, on a device of `cus` CUs (0: the handle's).
 * out[9] = { column groups of the batch, SOSRT_PLAN_GEMM_*, capacity of the live-column contraction (0: dense tiling),
 *            SOSRT_PLAN_TRANSPORT_*, workgroups per column of the chunk-parallel transport, 1 if the general kernel follows as
 *            a repair pass, 1 if this and all later orders go into one order-loop launch, its transport workgroups per
 *            column, its grid }. */
#define SOSRT_PLAN_GEMM_DENSE        0   /* 64-row tiles over the batch's row lists (tiles of converged columns leave at once) */
#define SOSRT_PLAN_GEMM_LIVE64       1   /* tiles laid over the live columns, 64 rows                                           */
#define SOSRT_PLAN_GEMM_LIVE32       2   /* ... 32 rows (at most 200 live columns)                                              */
#define SOSRT_PLAN_GEMM_LIVE32_DEEP  3   /* ... both operands staged two chunks ahead (at most 32 live columns)                */
#define SOSRT_PLAN_GEMM_LIVE16_REGS  4   /* ... 16 rows, a lane's matrix fragments in registers, no barrier in the k-loop (the   */
                                         /*     last few live columns of the symmetric form: a tile's latency is the launch's)    */
#define SOSRT_PLAN_TRANSPORT_GENERAL 0   /* kernels.hip k_transport                                                             */
#define SOSRT_PLAN_TRANSPORT_FAST    1   /* transport_fast.hip (odd N, N > 256)                                                 */
#define SOSRT_PLAN_TRANSPORT_RING    3   /* transport_ring.hip                                                                  */
#define SOSRT_PLAN_TRANSPORT_SCAN    4   /* transport_scan.hip (chunk-parallel)                                                 */
int sosrt_plan_launch(sosrt_t* h, int batch, int live, int surface, int zones, int cus, int* out /*[9]*/);

/* ---- profiling: HIP-event timing of the dominant kernels on the handle's stream ------------- */
#define SOSRT_K_GEMM      0
#define SOSRT_K_TRANSPORT 1
#define SOSRT_K_FIRST     2
#define SOSRT_K_SMALLMU   3
#define SOSRT_K_ORDER_LOOP 4  /* the order-loop launches (several orders of a few columns each) */
#define SOSRT_K_COUNT     5
int sosrt_profile_enable(sosrt_t* h, int on);
int sosrt_profile_reset(sosrt_t* h);
/* total milliseconds and launch count since the last reset (synchronises the stream) */
int sosrt_profile_get(sosrt_t* h, int kernel, double* total_ms, long long* launches, double* work /*flops or bytes*/);

/* ---- diagnostics: device buffer [B][2][8] of clock64() stamps written by the fast transport kernel
 * (start, prologue done, downward done, surface done, upward done, end); NULL switches it off */
int sosrt_debug_stamps(sosrt_t* h, unsigned long long* d_stamps);

/* ---- machine peaks measured on this device (roofline denominators) --------------------------- */
/* which 0: back-to-back v_mfma_f64_16x16x4_f64, TFLOP/s; 1: streaming copy of 1 GiB, GB/s (read+write);
 * 2: v_fma_f64, TFLOP/s. */
int sosrt_microbench(sosrt_t* h, int which, double* result);

#ifdef __cplusplus
}
#endif
#endif /* SOSRT_H */
