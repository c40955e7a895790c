// Latency microbenchmarks for the lone-column analysis: one workgroup, s_memtime stamps.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o gpurun_out/microbench && gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ long long now() { return __builtin_readcyclecounter(); }

// 2. dependent LDS reads
__global__ void k_lds(int n, long long* out, int* sink) {
    __shared__ int s[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (i + 17) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    const long long t0 = now();
    for (int i = 0; i < n; ++i) p = s[p];
    const long long t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (p == -1) *sink = p;
}
// 3. dependent double fma
__global__ void k_fma(int n, double a, double b, long long* out, double* sink) {
    double x = threadIdx.x;
    const long long t0 = now();
#pragma unroll 16
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, a, b);
    const long long t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (x == 12345.0) *sink = x;
}
// 3b. independent double fma (issue rate of one wave)
__global__ void k_fma_ind(int n, double a, double b, long long* out, double* sink) {
    double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    const long long t0 = now();
    for (int i = 0; i < n; i += 8) {
        x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
        x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
    }
    const long long t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    x0 += x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (x0 == 12345.0) *sink = x0;
}
// 4. barrier round trip, nw waves
__global__ void k_barrier(int n, long long* out) {
    const long long t0 = now();
    for (int i = 0; i < n; ++i) asm volatile("s_barrier" ::: "memory");
    const long long t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
// 5. store then wait vmcnt(0): the round trip of a store
__global__ void k_store(double* buf, int n, int stride, long long* out) {
    const long long t0 = now();
    for (int i = 0; i < n; ++i) {
        buf[(size_t)i * stride + threadIdx.x] = i;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
// 6. one load then wait: round trip of an independent load at a fresh address each time (stride in doubles)
__global__ void k_load(const double* buf, int n, int stride, long long* out, double* sink) {
    double acc = 0;
    const long long t0 = now();
    for (int i = 0; i < n; ++i) {
        const double v = __builtin_nontemporal_load(buf + (size_t)i * stride + threadIdx.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += v;
    }
    const long long t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 12345.0) *sink = acc;
}
// 7. the ring skeleton: 2 loader waves stream `rows` rows of 3 arrays (1 KiB per row and array) through LDS with
//    buffer_load..lds, R chunks of TC rows ahead; 2 consumer waves read the chunk from LDS, do `work` dependent fmas per row
//    and store 2 rows of 512 B per wave; one barrier per chunk.  What a lone column costs with no arithmetic at all.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int TC, int NS, int MODE, int WORK, int WIDE>
__global__ __launch_bounds__(256) void k_skel(const double* A, const double* B, const double* C, double* O1, double* O2, int rows,
                                              long long* out) {
    // MODE bits: 1 = no loads issued, 2 = no stores issued, 4 = no LDS reads.  Straight-line loops: every switch is a template
    // parameter (a lone wave pays for every branch).
    extern __shared__ double ring[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int bytes = rows * 2048;
    const size_t col = (size_t)blockIdx.x * rows * 256;
    const __amdgpu_buffer_rsrc_t rA = rsrc(A + col, bytes), rB = rsrc(B + col, bytes), rC = rsrc(C + col, bytes);
    const __amdgpu_buffer_rsrc_t r1 = rsrc(O1 + col, bytes), r2 = rsrc(O2 + col, bytes);
    const int NCH = rows / TC;
    constexpr int R = NS - 1, SLOT = 3 * TC * 128;
    const long long t0 = now();
    if (wid >= 2) {
        const int lid = wid - 2;
        int slot = 0;
        auto issue = [&](int q) {
            double* dst = ring + (size_t)slot * SLOT;
            slot = slot + 1 == NS ? 0 : slot + 1;
            if (MODE & 1) return;
#pragma unroll
            for (int i = 0; i < TC / 2; ++i) {
                const int u = 2 * i + lid;
                const int so = (q * TC + u) * 2048;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(dst + (0 * TC + u) * 128), 16, lane * 16, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(dst + (1 * TC + u) * 128), 16, lane * 16, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rC, (lds_ptr_t)(dst + (2 * TC + u) * 128), 16, lane * 16, so, 0, 0);
            }
        };
        for (int q = 0; q < R && q < NCH; ++q) issue(q);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        constexpr int KEEP = (R - 1) * (3 * TC / 2) > 60 ? 60 : (R - 1) * (3 * TC / 2);
        for (int q = 0; q < NCH; ++q) {
            if (q + R < NCH) {
                issue(q + R);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        asm volatile("s_barrier" ::: "memory");
        double x = 0;
        int slot = 0;
        for (int q = 0; q < NCH; ++q) {
            const double* sp = ring + (size_t)slot * SLOT + tid;
            slot = slot + 1 == NS ? 0 : slot + 1;
            double a[TC], b[TC], c[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                if (MODE & 4) { a[u] = q; b[u] = u; c[u] = tid; }
                else { a[u] = sp[(0 * TC + u) * 128]; b[u] = sp[(1 * TC + u) * 128]; c[u] = sp[(2 * TC + u) * 128]; }
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
#pragma unroll
                for (int w = 0; w < WORK; ++w) x = __builtin_fma(x, b[u], a[u]);
                if (WORK == 0) x += a[u];
                const int so = (q * TC + u) * 2048;
                if (MODE & 2) continue;
                if (WIDE) {
                    if (u & 1) {
                        typedef double d2 __attribute__((ext_vector_type(2)));
                        d2 v1 = {x, x}, v2 = {c[u] + x, c[u - 1] + x};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v1), r1, tid * 16, so - 2048, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v2), r2, tid * 16, so - 2048, 0);
                    }
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, x), r1, tid * 8, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, c[u] + x), r2, tid * 8, so, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (x == 12345.0) out[1] = 1;
    }
    const long long t1 = now();
    if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// 7a. the skeleton with the rows of a chunk split between two sets of consumer waves (set P0 rows 0..TC/2-1, set P1 the
//     rest), two barriers per chunk: in phase X the P0 waves run their part of the dependent chain (state handed over through
//     LDS) and store, the P1 waves read their rows and prepare (WORK independent operations per row); in phase Y the roles
//     swap.  What a column could cost with the recurrence as the only serial part.
template <int TC, int NS, int WORK>
__global__ __launch_bounds__(384) void k_skel_split(const double* A, const double* B, const double* C, double* O1, double* O2, int rows,
                                                    long long* out) {
    extern __shared__ double ring[];
    constexpr int H = TC / 2, SLOT = 3 * TC * 128, R = NS - 2;
    double* s_state = ring + (size_t)NS * SLOT;              // [128] recurrence state handed between the two sets
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int bytes = rows * 2048;
    const size_t col = (size_t)blockIdx.x * rows * 256;
    const __amdgpu_buffer_rsrc_t rA = rsrc(A + col, bytes), rB = rsrc(B + col, bytes), rC = rsrc(C + col, bytes);
    const __amdgpu_buffer_rsrc_t r1 = rsrc(O1 + col, bytes), r2 = rsrc(O2 + col, bytes);
    const int NCH = rows / TC;
    if (tid < 128) s_state[tid] = 0;
    const long long t0 = now();
    if (wid >= 4) {                                          // loaders: one chunk per two barriers
        const int lid = wid - 4;
        int slot = 0;
        auto issue = [&](int q) {
            double* dst = ring + (size_t)slot * SLOT;
            slot = slot + 1 == NS ? 0 : slot + 1;
#pragma unroll
            for (int i = 0; i < TC / 2; ++i) {
                const int u = 2 * i + lid;
                const int so = (q * TC + u) * 2048;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(dst + (0 * TC + u) * 128), 16, lane * 16, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(dst + (1 * TC + u) * 128), 16, lane * 16, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rC, (lds_ptr_t)(dst + (2 * TC + u) * 128), 16, lane * 16, so, 0, 0);
            }
        };
        for (int q = 0; q < R + 1 && q < NCH; ++q) issue(q);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        constexpr int KEEP = R * (3 * TC / 2) > 60 ? 60 : R * (3 * TC / 2);
        for (int q = 0; q < NCH; ++q) {
            if (q + R + 1 < NCH) {
                issue(q + R + 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
    } else {
        const int p = wid >> 1, l = (wid & 1) * 64 + lane;   // row set, direction
        double a[H], b[H], c[H], pre[H];
        auto prepare = [&](int q) {                           // rows p*H .. p*H+H-1 of chunk q
            const double* sp = ring + (size_t)(q % NS) * SLOT + l;
#pragma unroll
            for (int u = 0; u < H; ++u) {
                a[u] = sp[(0 * TC + p * H + u) * 128]; b[u] = sp[(1 * TC + p * H + u) * 128]; c[u] = sp[(2 * TC + p * H + u) * 128];
            }
#pragma unroll
            for (int u = 0; u < H; ++u) {
                double y = a[u];
#pragma unroll
                for (int w = 0; w < WORK; ++w) y = __builtin_fma(y, b[u], a[u]);
                pre[u] = y;
            }
        };
        auto chain = [&](int q) {
            double x = s_state[l];
#pragma unroll
            for (int u = 0; u < H; ++u) {
                x = __builtin_fma(x, b[u], pre[u]);
                const int so = (q * TC + p * H + u) * 2048;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, x), r1, l * 8, so, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, c[u] + x), r2, l * 8, so, 0);
            }
            s_state[l] = x;
        };
        asm volatile("s_barrier" ::: "memory");
        if (p == 0) prepare(0);
        for (int q = 0; q < NCH; ++q) {
            // phase X: P0 chain(q), P1 prepare(q)
            if (p == 0) chain(q); else prepare(q);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // phase Y: P1 chain(q), P0 prepare(q + 1)
            if (p == 1) chain(q); else if (q + 1 < NCH) prepare(q + 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    const long long t1 = now();
    if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// 7b. branch cost: a loop of n iterations with k uniform taken branches each (s_cbranch over one instruction)
template <int K>
__global__ void k_branch(int n, int z, long long* out, int* sink) {
    int acc = 0;
    const long long t0 = now();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            asm volatile("s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 1f\n\tv_add_u32 %0, %0, 1\n1:\n\tv_add_u32 %0, %0, 2" : "+v"(acc) : "s"(z) : "scc");
        }
    }
    const long long t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == -1) *sink = acc;
}
// 7c. issue rates of one wave: independent ds_read_b64, buffer_store_b64 (to one row: no DRAM pressure), v_add_u32
__global__ void k_rates(double* buf, int n, long long* out, double* sink) {
    __shared__ double s[8 * 128];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = i;
    __syncthreads();
    const int tid = threadIdx.x;
    double acc = 0;
    long long t0 = now();
    for (int i = 0; i < n; ++i) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s[u * 128 + tid];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    long long t1 = now();
    if (tid == 0) out[0] = t1 - t0;
    const __amdgpu_buffer_rsrc_t r = rsrc(buf, 1 << 20);
    t0 = now();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, acc), r, tid * 8, u * 2048, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = now();
    if (tid == 0) out[1] = t1 - t0;
    unsigned k = tid;
    t0 = now();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("v_add_u32 %0, %0, 3" : "+v"(k));
    }
    t1 = now();
    if (tid == 0) out[2] = t1 - t0;
    if (acc == 12345.0 || k == 77) *sink = acc;
}

// 8. no LDS, no loader waves: every wave prefetches its own lanes of the next chunk into registers (one chunk ahead)
template <int TC>
__global__ __launch_bounds__(128) void k_regs(const double* A, const double* B, const double* C, double* O1, double* O2, int rows,
                                              long long* out) {
    const int tid = threadIdx.x;
    const int bytes = rows * 2048;
    const size_t col = (size_t)blockIdx.x * rows * 256;
    const __amdgpu_buffer_rsrc_t rA = rsrc(A + col, bytes), rB = rsrc(B + col, bytes), rC = rsrc(C + col, bytes);
    const __amdgpu_buffer_rsrc_t r1 = rsrc(O1 + col, bytes), r2 = rsrc(O2 + col, bytes);
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    const int NCH = rows / TC;
    const long long t0 = now();
    u2 a[TC], b[TC], c[TC], na[TC], nb[TC], nc[TC];
#pragma unroll
    for (int u = 0; u < TC; ++u) {
        a[u] = __builtin_amdgcn_raw_buffer_load_b64(rA, tid * 8, u * 2048, 0);
        b[u] = __builtin_amdgcn_raw_buffer_load_b64(rB, tid * 8, u * 2048, 0);
        c[u] = __builtin_amdgcn_raw_buffer_load_b64(rC, tid * 8, u * 2048, 0);
    }
    double x = 0;
    for (int q = 0; q < NCH; ++q) {
        const int qn = min(q + 1, NCH - 1);
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            na[u] = __builtin_amdgcn_raw_buffer_load_b64(rA, tid * 8, (qn * TC + u) * 2048, 0);
            nb[u] = __builtin_amdgcn_raw_buffer_load_b64(rB, tid * 8, (qn * TC + u) * 2048, 0);
            nc[u] = __builtin_amdgcn_raw_buffer_load_b64(rC, tid * 8, (qn * TC + u) * 2048, 0);
        }
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            x = __builtin_fma(x, __builtin_bit_cast(double, b[u]), __builtin_bit_cast(double, a[u]));
            const int so = (q * TC + u) * 2048;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, x), r1, tid * 8, so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, __builtin_bit_cast(double, c[u]) + x), r2, tid * 8, so, 0);
        }
#pragma unroll
        for (int u = 0; u < TC; ++u) { a[u] = na[u]; b[u] = nb[u]; c[u] = nc[u]; }
    }
    const long long t1 = now();
    if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
// 9. what one CU streams: a workgroup copies `bytes` with 16-byte lanes, `unroll` loads in flight per thread
template <int UN>
__global__ __launch_bounds__(1024) void k_copy(const double* src, double* dst, size_t bytes, long long* out) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const u4* s = reinterpret_cast<const u4*>(src) + (size_t)blockIdx.x * (bytes / 16);
    u4* d = reinterpret_cast<u4*>(dst) + (size_t)blockIdx.x * (bytes / 16);
    const size_t n = bytes / 16;
    const long long t0 = now();
    for (size_t i = threadIdx.x; i < n; i += (size_t)blockDim.x * UN) {
        u4 v[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) v[k] = i + (size_t)k * blockDim.x < n ? __builtin_nontemporal_load(s + i + (size_t)k * blockDim.x) : u4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < UN; ++k) if (i + (size_t)k * blockDim.x < n) d[i + (size_t)k * blockDim.x] = v[k];
    }
    __syncthreads();
    const long long t1 = now();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main() {
    long long* d_out; int* d_sink; double* d_dsink;
    CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_sink, 64)); CK(hipMalloc(&d_dsink, 64));
    auto get = [&]() { long long v; CK(hipMemcpy(&v, d_out, 8, hipMemcpyDeviceToHost)); return (double)v; };
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s, clock %d kHz, wall-clock rate %d kHz\n", pr.name, pr.clockRate, 0);
    // calibrate the cycle counter against hip events
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int n = 1 << 20;
        k_fma<<<1, 64>>>(1 << 16, 1.0000001, 1e-9, d_out, d_dsink);
        CK(hipEventRecord(e0)); k_fma<<<1, 64>>>(n, 1.0000001, 1e-9, d_out, d_dsink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = get();
        printf("dependent v_fma_f64: %.2f counter ticks each; kernel %.3f ms -> counter runs at %.1f MHz; %.2f ns per fma\n", cyc / n, ms,
               cyc / ms / 1e3, ms * 1e6 / n);
        k_fma_ind<<<1, 64>>>(n, 1.0000001, 1e-9, d_out, d_dsink); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_fma_ind<<<1, 64>>>(n, 1.0000001, 1e-9, d_out, d_dsink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("independent v_fma_f64, one wave: %.2f ns each\n", ms * 1e6 / n);
    }
    double tick_ns;
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int n = 1 << 20;
        CK(hipEventRecord(e0)); k_fma<<<1, 64>>>(n, 1.0000001, 1e-9, d_out, d_dsink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        tick_ns = ms * 1e6 / get();
    }
    k_lds<<<1, 64>>>(4096, d_out, d_sink); CK(hipDeviceSynchronize());
    printf("dependent ds_read_b32: %.1f ns\n", get() / 4096 * tick_ns);
    for (int nw : {1, 4, 8, 16}) {
        k_barrier<<<1, 64 * nw>>>(4096, d_out); CK(hipDeviceSynchronize());
        printf("s_barrier, %2d waves: %.1f ns\n", nw, get() / 4096 * tick_ns);
    }
    {
        double* buf; const size_t bytes = (size_t)1 << 30; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
        for (int stride : {64, 256, 8192}) {
            const int n = 2048;
            k_store<<<1, 64>>>(buf, n, stride, d_out); CK(hipDeviceSynchronize());
            const double s = get() / n * tick_ns;
            k_load<<<1, 64>>>(buf + (size_t)n * stride, n, stride, d_out, d_dsink); CK(hipDeviceSynchronize());
            printf("stride %6d B: store + vmcnt(0) %.0f ns, fresh load + vmcnt(0) %.0f ns\n", stride * 8, s, get() / n * tick_ns);
        }
        // the ring skeleton
        const int rows = 400;
        double *A = buf, *B = buf + ((size_t)32 << 20 >> 3), *C = buf + ((size_t)64 << 20 >> 3), *O1 = buf + ((size_t)96 << 20 >> 3),
               *O2 = buf + ((size_t)128 << 20 >> 3);
        B = buf + ((size_t)32 << 20 >> 3);
        auto run_skel = [&](auto kern, size_t shm, int cols, const char* what, int TCv) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            for (int rep = 0; rep < 2; ++rep) { kern<<<cols, 256, shm>>>(A, B, C, O1, O2, rows, d_out); CK(hipDeviceSynchronize()); }
            const double t = get() * tick_ns;
            printf("skeleton %3d col, %-44s %.1f us (%.0f ns/chunk, %.1f ns/row)\n", cols, what, t / 1e3, t / (rows / TCv), t / rows);
        };
#define SK(TC_, NS_, MODE_, WORK_, WIDE_, what) run_skel(k_skel<TC_, NS_, MODE_, WORK_, WIDE_>, (size_t)NS_ * 3 * TC_ * 1024, 1, what, TC_)
        SK(8, 3, 0, 1, 0, "TC 8, 3 slots, 1 fma, 8 B stores: all");
        SK(8, 3, 1, 1, 0, "TC 8, 3 slots, 1 fma, 8 B stores: no loads");
        SK(8, 3, 2, 1, 0, "TC 8, 3 slots, 1 fma: no stores");
        SK(8, 3, 4, 1, 0, "TC 8, 3 slots, 1 fma, 8 B stores: no LDS reads");
        SK(8, 3, 6, 1, 0, "TC 8, 3 slots, 1 fma: no stores, no LDS reads");
        SK(8, 3, 7, 1, 0, "TC 8, 3 slots, 1 fma: barriers only");
        SK(8, 3, 0, 1, 1, "TC 8, 3 slots, 1 fma, 16 B stores: all");
        SK(8, 3, 0, 8, 0, "TC 8, 3 slots, 8 fma, 8 B stores: all");
        SK(8, 3, 0, 32, 0, "TC 8, 3 slots, 32 fma, 8 B stores: all");
        SK(8, 2, 0, 1, 0, "TC 8, 2 slots, 1 fma, 8 B stores: all");
        SK(8, 4, 0, 1, 0, "TC 8, 4 slots, 1 fma, 8 B stores: all");
        SK(16, 3, 0, 1, 0, "TC 16, 3 slots, 1 fma, 8 B stores: all");
        SK(16, 3, 7, 1, 0, "TC 16, 3 slots, 1 fma: barriers only");
        SK(4, 3, 0, 1, 0, "TC 4, 3 slots, 1 fma, 8 B stores: all");
        {
            auto run_split = [&](auto kern, size_t shm, const char* what) {
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                for (int rep = 0; rep < 2; ++rep) { kern<<<1, 384, shm>>>(A, B, C, O1, O2, rows, d_out); CK(hipGetLastError()); CK(hipDeviceSynchronize()); }
                const double t = get() * tick_ns;
                printf("skeleton, rows of a chunk split between two wave sets, %-28s %.1f us (%.0f ns/chunk)\n", what, t / 1e3, t / (rows / 8));
            };
            run_split(k_skel_split<8, 4, 0>, (size_t)4 * 3 * 8 * 1024 + 1024, "0 operations per row:");
            run_split(k_skel_split<8, 4, 3>, (size_t)4 * 3 * 8 * 1024 + 1024, "3 operations per row:");
            run_split(k_skel_split<8, 4, 6>, (size_t)4 * 3 * 8 * 1024 + 1024, "6 operations per row:");
            SK(8, 3, 0, 4, 0, "TC 8, 3 slots, 4 fma, 8 B stores: all");
            SK(8, 3, 0, 7, 0, "TC 8, 3 slots, 7 fma, 8 B stores: all");
        }
        run_skel(k_skel<8, 3, 0, 1, 0>, (size_t)3 * 3 * 8 * 1024, 64, "TC 8, 3 slots, 1 fma, 8 B stores: all", 8);
        run_skel(k_skel<8, 3, 0, 1, 0>, (size_t)3 * 3 * 8 * 1024, 256, "TC 8, 3 slots, 1 fma, 8 B stores: all", 8);
        run_skel(k_skel<8, 3, 0, 1, 0>, (size_t)3 * 3 * 8 * 1024, 512, "TC 8, 3 slots, 1 fma, 8 B stores: all", 8);
        {
            const int n = 4096;
            k_branch<8><<<1, 64>>>(n, 1, d_out, d_sink); CK(hipDeviceSynchronize());
            const double taken = get() / (n * 8.0);
            k_branch<8><<<1, 64>>>(n, 0, d_out, d_sink); CK(hipDeviceSynchronize());
            printf("s_cmp + s_cbranch + v_add: taken %.1f cycles, not taken (+1 v_add) %.1f cycles\n", taken, get() / (n * 8.0));
            k_rates<<<1, 64>>>(O1, n, d_out, d_dsink); CK(hipDeviceSynchronize());
            long long r[3]; CK(hipMemcpy(r, d_out, 24, hipMemcpyDeviceToHost));
            printf("one wave: 8 ds_read_b64 + wait + 8 v_add_f64: %.1f cycles per group; buffer_store_b64 %.1f cycles each; v_add_u32 %.1f cycles each\n",
                   r[0] / (double)n, r[1] / (n * 8.0), r[2] / (n * 8.0));
        }
        for (int cols : {1, 64, 256}) {
            k_regs<8><<<cols, 128>>>(A, B, C, O1, O2, rows, d_out); CK(hipDeviceSynchronize());
            k_regs<8><<<cols, 128>>>(A, B, C, O1, O2, rows, d_out); CK(hipDeviceSynchronize());
            const double t8 = get() * tick_ns;
            k_regs<4><<<cols, 128>>>(A, B, C, O1, O2, rows, d_out); CK(hipDeviceSynchronize());
            k_regs<4><<<cols, 128>>>(A, B, C, O1, O2, rows, d_out); CK(hipDeviceSynchronize());
            printf("register prefetch, %3d col: TC=8 %.1f us (%.1f ns/row)  TC=4 %.1f us\n", cols, t8 / 1e3, t8 / rows, get() * tick_ns / 1e3);
        }
        for (int nt : {256, 512, 1024}) {
            const size_t bytes = (size_t)1 << 20;
            k_copy<4><<<1, nt>>>(A, O1, bytes, d_out); CK(hipDeviceSynchronize());
            k_copy<4><<<1, nt>>>(A, O1, bytes, d_out); CK(hipDeviceSynchronize());
            const double t4 = get() * tick_ns;
            k_copy<8><<<1, nt>>>(A, O1, bytes, d_out); CK(hipDeviceSynchronize());
            k_copy<8><<<1, nt>>>(A, O1, bytes, d_out); CK(hipDeviceSynchronize());
            const double t8 = get() * tick_ns;
            printf("one workgroup of %4d threads copies 1 MiB: %.1f us (%.0f GB/s read + write) with 4 in flight, %.1f us (%.0f GB/s) with 8\n", nt, t4 / 1e3,
                   2.0 * bytes / t4, t8 / 1e3, 2.0 * bytes / t8);
        }
        CK(hipFree(buf));
    }
    return 0;
}
