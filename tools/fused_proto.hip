// fused_proto: a measurement, not a product kernel.
//
// Question: what does it cost, and what does it save, to consume the DOWNWARD half of the source function Jn inside the
// contraction that produces it -- one persistent workgroup per column walking the rows top-down in 16-row blocks, the downward
// recurrence of the block run from the accumulators, only the upward half of Jn parked in memory -- instead of writing all of Jn
// and reading it back in a transport kernel?
//
// The skeleton has the contraction's arithmetic (flip-symmetric form: X = sum_k c (a_k + b_k) S[k][m], Y = sum_k c (a_k - b_k) A[k][m],
// v_mfma_f64_16x16x4_f64, k in steps of 4) and the plain downward recurrence D_t = E_t D_{t-1} + (h/|mu|) (J_{t-1} E_t + J_t) with
// In and the running total I updated in place; it has none of the mu -> 0 treatments, zone boundaries or convergence test (all of
// which only add work to the fused kernel).  Three kernels on the headline shape (512 columns, L = 200, N = 128):
//   MODE 0  contraction only: Jn written whole (what k_jn_gemm does; here one workgroup of 8 waves per column, barrier-free k-loop)
//   MODE 1  fused: upward half of Jn parked, downward half consumed in the kernel (In, I of the downward half written)
//   MODE 3  MODE 1 with the recurrence on two extra waves, one block behind the contraction waves (double-buffered in LDS)
//   MODE 2  the downward sweep alone, reading Jn from memory (the skeleton of what the transport kernel does for that half)
// Compare  t(1), t(3)  with  t(0) + t(2),  and the HBM bytes of the three (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE).
//
//   hipcc -O3 --offload-arch=gfx950 -o fused_proto tools/fused_proto.hip && ./fused_proto [columns]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                                  \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; }      \
    } while (0)

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int L = 200, N = 128, D = 2 * N, BR = 16;           // rows per block
constexpr int RS = D + 2;                                     // LDS row stride of the staged rows of In_1
constexpr int JS = N + 2;                                     // LDS row stride of the block's downward Jn

struct Args {
    const double* In1;      // [B][L][D]
    const double* Ws;       // [N][D]: S | A  (k < N)
    const double* E;        // [8][L][D] attenuation tables (8 optical-depth profiles)
    const double* hd;       // [L] half layer thicknesses
    const double* rmu;      // [N] 1 / |mu|
    double* Jn;             // [B][L][D]
    double* In;             // [B][L][D]
    double* I;              // [B][L][D]
    double coef;
    int skip;               // diagnosis of MODE 3: 1 = no loads of E and I, 2 = no stores of In and I, 4 = no recurrence work at all
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_proto(Args a) {
    extern __shared__ double sm[];
    double* raw = sm;                              // [BR][RS]
    double* jdn = sm + BR * RS;                    // [BR][JS]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const size_t col = (size_t)b * L * D;
    const __amdgpu_buffer_rsrc_t rA = rsrc(a.In1 + col, L * D * 8);
    const double* __restrict__ Et = a.E + (size_t)(b & 7) * L * D;
    const int m0 = 16 * wave;                      // this wave's 16 values of m (X and Y)
    // recurrence state of direction m = tid (threads 0 .. N-1)
    double Dst = 0, Jprev = 0;
    const double rm = tid < N ? a.rmu[tid] : 0.0;

    for (int r0 = 0; r0 < L; r0 += BR) {
        const int nr = min(BR, L - r0);
        f64x4 X = {0, 0, 0, 0}, Y = {0, 0, 0, 0};
        if (MODE != 2) {
            // the block's rows of In_1, whole, by LDS-DMA: 16 rows x 2 pieces of 1 KiB over 8 waves
            for (int q = wave; q < BR * 2; q += 8) {
                const int r = q >> 1, pc = q & 1;
                if (r < nr)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(raw + r * RS + pc * 128), 16, lane * 16, (r0 + r) * D * 8 + pc * 1024, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // barrier-free k-loop: A from LDS, W straight from L2
            const double* __restrict__ Wp = a.Ws + m0 + fr;
            const bool rowok = fr < nr;
#pragma unroll 8
            for (int kk = 0; kk < N; kk += 4) {
                const int k = kk + fk;
                const double av = rowok ? raw[fr * RS + k] : 0.0, bv = rowok ? raw[fr * RS + D - 1 - k] : 0.0;
                const double s = Wp[(size_t)k * D], t = Wp[(size_t)k * D + N];
                X = __builtin_amdgcn_mfma_f64_16x16x4f64(a.coef * (av + bv), s, X, 0, 0, 0);
                Y = __builtin_amdgcn_mfma_f64_16x16x4f64(a.coef * (av - bv), t, Y, 0, 0, 0);
            }
            const int m = m0 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * r + fk;
                if (i < nr) {
                    const double x = X[r], y = Y[r];
                    a.Jn[col + (size_t)(r0 + i) * D + D - 1 - m] = x - y;              // the upward half is parked either way
                    if (MODE == 0) a.Jn[col + (size_t)(r0 + i) * D + m] = x + y;
                    else jdn[i * JS + m] = x + y;
                }
            }
        }
        if (MODE != 0) {
            __syncthreads();
            if (tid < N) {
                // the block's downward rows: loads first, then the serial steps
                double Ev[BR], Iv[BR], Jv[BR];
#pragma unroll
                for (int i = 0; i < BR; ++i) {
                    const int t = min(r0 + i, L - 1);
                    Ev[i] = Et[(size_t)t * D + tid];
                    Iv[i] = a.I[col + (size_t)t * D + tid];
                    Jv[i] = MODE == 2 ? a.Jn[col + (size_t)t * D + tid] : jdn[i * JS + tid];
                }
#pragma unroll
                for (int i = 0; i < BR; ++i) {
                    if (i < nr) {
                        const int t = r0 + i;
                        const double c = a.hd[t] * rm * __builtin_fma(Jprev, Ev[i], Jv[i]);
                        Dst = __builtin_fma(Dst, Ev[i], c);
                        a.In[col + (size_t)t * D + tid] = Dst;
                        a.I[col + (size_t)t * D + tid] = Iv[i] + Dst;
                        Jprev = Jv[i];
                    }
                }
            }
        }
        __syncthreads();                            // the next block reuses raw / jdn
    }
}


// MODE 3: the same fusion with the recurrence on two waves of its own (waves 8 and 9), one block behind the eight contraction waves,
// the block's downward Jn double-buffered in LDS -- the recurrence of block i - 1 runs under the k-loop of block i.
__global__ __launch_bounds__(640) void k_proto_piped(Args a) {
    extern __shared__ double sm[];
    double* raw = sm;                              // [BR][RS]
    double* jdn = sm + BR * RS;                    // [2][BR][JS]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const size_t col = (size_t)b * L * D;
    const __amdgpu_buffer_rsrc_t rA = rsrc(a.In1 + col, L * D * 8);
    const double* __restrict__ Et = a.E + (size_t)(b & 7) * L * D;
    const int m0 = 16 * wave;
    const int rt = tid - 512;                      // direction of a recurrence thread
    double Dst = 0, Jprev = 0;
    const double rm = rt >= 0 ? a.rmu[rt] : 0.0;
    const int nblk = (L + BR - 1) / BR;
    for (int ib = 0; ib <= nblk; ++ib) {
        const int r0 = ib * BR, nr = min(BR, L - r0);
        if (wave < 8) {
            if (ib < nblk) {
                for (int q = wave; q < BR * 2; q += 8) {
                    const int r = q >> 1, pc = q & 1;
                    if (r < nr)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(raw + r * RS + pc * 128), 16, lane * 16, (r0 + r) * D * 8 + pc * 1024, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();                            // raw(ib) staged; jdn[(ib-1)&1] complete (barrier at the end of the last pass)
        if (wave < 8) {
            if (ib < nblk) {
                f64x4 X = {0, 0, 0, 0}, Y = {0, 0, 0, 0};
                const double* __restrict__ Wp = a.Ws + m0 + fr;
                const bool rowok = fr < nr;
#pragma unroll 8
                for (int kk = 0; kk < N; kk += 4) {
                    const int k = kk + fk;
                    const double av = rowok ? raw[fr * RS + k] : 0.0, bv = rowok ? raw[fr * RS + D - 1 - k] : 0.0;
                    const double s = Wp[(size_t)k * D], t = Wp[(size_t)k * D + N];
                    X = __builtin_amdgcn_mfma_f64_16x16x4f64(a.coef * (av + bv), s, X, 0, 0, 0);
                    Y = __builtin_amdgcn_mfma_f64_16x16x4f64(a.coef * (av - bv), t, Y, 0, 0, 0);
                }
                const int m = m0 + fr;
                double* jb = jdn + (ib & 1) * BR * JS;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * r + fk;
                    if (i < nr) {
                        a.Jn[col + (size_t)(r0 + i) * D + D - 1 - m] = X[r] - Y[r];
                        jb[i * JS + m] = X[r] + Y[r];
                    }
                }
            }
        } else if (ib > 0 && !(a.skip & 4)) {
            const int p0 = r0 - BR, pn = min(BR, L - p0);
            const double* jb = jdn + ((ib - 1) & 1) * BR * JS;
            double Ev[BR], Iv[BR];
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int t = min(p0 + i, L - 1);
                Ev[i] = (a.skip & 1) ? 0.9 : Et[(size_t)t * D + rt];
                Iv[i] = (a.skip & 1) ? 0.1 : a.I[col + (size_t)t * D + rt];
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                if (i < pn) {
                    const int t = p0 + i;
                    const double J = jb[i * JS + rt];
                    const double c = a.hd[t] * rm * __builtin_fma(Jprev, Ev[i], J);
                    Dst = __builtin_fma(Dst, Ev[i], c);
                    if (!(a.skip & 2)) {
                        a.In[col + (size_t)t * D + rt] = Dst;
                        a.I[col + (size_t)t * D + rt] = Iv[i] + Dst;
                    }
                    Jprev = J;
                }
            }
        }
        __syncthreads();                            // raw free for the next block; jdn[ib&1] published
    }
}

template <int MODE>
static int run(const char* what, int B, const Args& a, int reps, double flop, double bytes) {
    const size_t shm = (size_t)(BR * RS + (MODE == 3 ? 2 : 1) * BR * JS) * sizeof(double);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto go = [&]() {
        if constexpr (MODE == 3) hipLaunchKernelGGL(k_proto_piped, dim3(B), dim3(640), shm, 0, a);
        else hipLaunchKernelGGL(k_proto<MODE>, dim3(B), dim3(512), shm, 0, a);
    };
    for (int w = 0; w < 3; ++w) go();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) go();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-62s %8.1f us per launch", what, us);
    if (flop > 0) printf("   %5.1f TFLOP/s executed", flop / us * 1e-6);
    if (bytes > 0) printf("   %5.2f TB/s of algorithmic bytes", bytes / us * 1e-6);
    printf("\n");
    return 0;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512;
    const size_t F = (size_t)B * L * D;
    std::vector<double> h(F), w((size_t)N * D), e((size_t)8 * L * D), hd(L), rmu(N);
    srand(1);
    for (auto& x : h) x = rand() / (double)RAND_MAX;
    for (auto& x : w) x = rand() / (double)RAND_MAX / N;
    for (auto& x : e) x = 0.5 + 0.5 * rand() / (double)RAND_MAX;
    for (auto& x : hd) x = 1e-3;
    for (int m = 0; m < N; ++m) rmu[m] = 1.0 / (1.0 - m / (double)N);
    Args a{};
    double *dIn1, *dW, *dE, *dhd, *drmu, *dJn, *dIn, *dI;
    CHK(hipMalloc(&dIn1, F * 8)); CHK(hipMalloc(&dJn, F * 8)); CHK(hipMalloc(&dIn, F * 8)); CHK(hipMalloc(&dI, F * 8));
    CHK(hipMalloc(&dW, w.size() * 8)); CHK(hipMalloc(&dE, e.size() * 8)); CHK(hipMalloc(&dhd, L * 8)); CHK(hipMalloc(&drmu, N * 8));
    CHK(hipMemcpy(dIn1, h.data(), F * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dI, h.data(), F * 8, hipMemcpyHostToDevice));
    CHK(hipMemset(dJn, 0, F * 8)); CHK(hipMemset(dIn, 0, F * 8));
    CHK(hipMemcpy(dW, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dE, e.data(), e.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dhd, hd.data(), L * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(drmu, rmu.data(), N * 8, hipMemcpyHostToDevice));
    a.In1 = dIn1; a.Ws = dW; a.E = dE; a.hd = dhd; a.rmu = drmu; a.Jn = dJn; a.In = dIn; a.I = dI; a.coef = 0.25;
    const size_t shm = (size_t)(BR * RS + BR * JS) * sizeof(double);
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_proto<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_proto<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_proto<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_proto_piped), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(shm + BR * JS * sizeof(double))));
    const double flop = (double)B * L * D * D;                      // L D^2 per column (symmetric form)
    const double LD = (double)B * L * D * 8;
    printf("fused_proto: %d columns, L = %d, N = %d; LDS %zu bytes per workgroup of 8 waves, one workgroup per column\n", B, L, N, shm);
    if (run<0>("MODE 0  contraction alone (Jn written whole)", B, a, 20, flop, 2 * LD)) return 1;
    if (run<1>("MODE 1  fused: contraction + downward sweep, upward Jn parked", B, a, 20, flop, LD + 0.5 * LD + 0.5 * 3 * LD)) return 1;
    if (run<3>("MODE 3  fused, recurrence on two waves of its own, one block behind", B, a, 20, flop, LD + 0.5 * LD + 0.5 * 3 * LD)) return 1;
    if (getenv("PROTO_DIAG")) {
        Args d = a;
        d.skip = 1; if (run<3>("MODE 3 without the loads of E and I", B, d, 20, flop, 0)) return 1;
        d.skip = 2; if (run<3>("MODE 3 without the stores of In and I", B, d, 20, flop, 0)) return 1;
        d.skip = 3; if (run<3>("MODE 3 without either (recurrence arithmetic and LDS only)", B, d, 20, flop, 0)) return 1;
        d.skip = 4; if (run<3>("MODE 3 with idle recurrence waves (barriers only)", B, d, 20, flop, 0)) return 1;
    }
    if (run<2>("MODE 2  downward sweep alone (Jn read back)", B, a, 20, 0, 0.5 * 4 * LD)) return 1;
    printf("algorithmic HBM bytes per launch: MODE 0 %.0f MB (In_1 read, Jn written), MODE 1 %.0f MB (In_1 read, upward Jn written, I read + written and In written\n"
           "for the downward half), MODE 2 %.0f MB (Jn and I read, In and I written, downward half); 0 + 2 together %.0f MB\n",
           2 * LD / 1e6, (LD + 0.5 * LD + 1.5 * LD) / 1e6, 2 * LD / 1e6, 4 * LD / 1e6);
    return 0;
}
