// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  For a few mask patterns: 4096 short workgroups on
// the masked stream, every one records (XCC, SE, CU); the host counts the distinct CUs and how they spread over the XCCs.
// hipcc --offload-arch=gfx950 -O2 tools/cu_mask_where.hip -o tools/cu_mask_where && tools/cu_mask_where
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k_where(unsigned* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 15);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 300) __builtin_amdgcn_s_sleep(8);
}
static void run(const char* name, const std::vector<int>& bits, int words) {
    std::vector<uint32_t> m(words, 0);
    for (int b : bits) if (b / 32 < words) m[b / 32] |= 1u << (b % 32);
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, words, m.data());
    if (e != hipSuccess) { printf("%-44s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    const int G = 4096;
    unsigned* d; hipMalloc(&d, G * 4);
    hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(G);
    hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; int per[16] = {0};
    for (unsigned v : h) if (cus.insert(v).second) per[(v >> 16) & 15]++;
    printf("%-44s: %3zu bits set -> %3zu distinct CUs; per XCC:", name, bits.size(), cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
    printf("\n");
    hipFree(d); hipStreamDestroy(st);
}
int main() {
    std::vector<int> all, even, odd, lo, hi, q4, first32, x0;
    for (int i = 0; i < 256; ++i) { all.push_back(i); if (i % 2 == 0) even.push_back(i); else odd.push_back(i); if (i < 128) lo.push_back(i); else hi.push_back(i);
                                    if (i % 4 == 0) q4.push_back(i); if (i < 32) first32.push_back(i); if (i % 8 == 0) x0.push_back(i); }
    run("all 256 bits", all, 8);
    run("even bits", even, 8);
    run("odd bits", odd, 8);
    run("bits 0..127", lo, 8);
    run("bits 128..255", hi, 8);
    run("every 4th bit", q4, 8);
    run("bits 0..31", first32, 8);
    run("every 8th bit", x0, 8);
    run("bits 0..31, one word", first32, 1);
    return 0;
}
