// Device helpers shared by the wave-independent transport kernels (transport_fast.hip,
// transport_ring.hip): cross-lane reads, raw buffer addressing, Python's max() over a row.
#pragma once
#include <hip/hip_runtime.h>

namespace sosrt {
namespace {

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Buffer addressing (raw buffer ops on a per-column descriptor): the row offset lives in an SGPR,
// the lane offset in one VGPR that never changes, so a load or store costs no address arithmetic.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
// the same past the L1 (glc): for data this workgroup stored earlier in the kernel
__device__ __forceinline__ double bload_glc(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 1);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x) {
    u32x2 v;
    v.x = (unsigned)__double2loint(x);
    v.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
// the same with a cache policy (aux bits of the raw buffer intrinsics on gfx940+: 1 = sc0, 16 = sc1; 17 = both: a write-through
// store, performed at the memory the other XCDs' L2s read from, acknowledged (vmcnt) once it is there)
template <int AUX>
__device__ __forceinline__ void bstore_aux(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x) {
    u32x2 v;
    v.x = (unsigned)__double2loint(x);
    v.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, AUX);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// two adjacent doubles per lane (16 bytes): half the vector-memory instructions of the 8-byte forms
__device__ __forceinline__ double2 bload2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}
__device__ __forceinline__ void bstore2(__amdgpu_buffer_rsrc_t r, int voff, int soff, double2 x) {
    u32x4 v;
    v.x = (unsigned)__double2loint(x.x); v.y = (unsigned)__double2hiint(x.x);
    v.z = (unsigned)__double2loint(x.y); v.w = (unsigned)__double2hiint(x.y);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}

// value of lane + 1 (wave_shl:1); lane 63 keeps its own
__device__ __forceinline__ double lane_up1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_fmax_(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The arithmetic of the sweeps, spelled out operation by operation (no contraction left to the compiler): the ring
// kernel and the pipeline kernel must produce the same bits, because which of them transports a column depends on
// how many columns of its batch are still live.
//   source term of a row (spec:336-340, 393-399):  (h / |mu|) (Ja E + Jb),  h = half the layer thickness
__device__ __forceinline__ double rec_src(double h_rmu, double Ja, double E, double Jb) {
#pragma clang fp contract(off)
    return h_rmu * __builtin_fma(Ja, E, Jb);
}
__device__ __forceinline__ double rec_hr(double h, double rmu) {
#pragma clang fp contract(off)
    return h * rmu;
}
//   one step of the recurrence:  S E + c
__device__ __forceinline__ double rec_step(double S, double E, double c) { return __builtin_fma(S, E, c); }
__device__ __forceinline__ double rec_add(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
//   spec:407-409:  (1 - w) r0 + w rk,  w = mu_m / mu_k as mu_m * (1 / mu_k)
__device__ __forceinline__ double blend_weight(double mu_m, double rmu_k) {
#pragma clang fp contract(off)
    return mu_m * rmu_k;
}
__device__ __forceinline__ double blend_val(double w, double r0, double rk) {
#pragma clang fp contract(off)
    const double a = (1 - w) * r0;
    return __builtin_fma(w, rk, a);
}
//   In_limit:113-141 as a linear map: acc + c x
__device__ __forceinline__ double fix_acc(double c, double x, double acc) { return __builtin_fma(c, x, acc); }

// Python's max() over a row (see block_pymax in kernels.hip); first_tid holds element 0.
__device__ double block_pymax_(double x, bool valid, double* s_red, int first_tid) {
    const int tid = threadIdx.x, nw = blockDim.x >> 6;
    double v = wave_fmax_(valid ? x : __builtin_nan(""));
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    if (tid == first_tid) s_red[nw] = x;
    __syncthreads();
    double r = s_red[0];
    for (int i = 1; i < nw; ++i) r = fmax(r, s_red[i]);
    const double first = s_red[nw];
    return (first != first) ? first : r;
}

}  // namespace
}  // namespace sosrt
