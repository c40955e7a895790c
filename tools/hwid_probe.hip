// Where do the waves of co-resident workgroups sit?  512 workgroups of 4 waves with 77 KB of LDS each (two per CU, like the ring
// transport's dense launch), every wave records HW_ID (wave slot, SIMD, CU, SE, workgroup slot) and XCC_ID; the host prints, per
// workgroup, the SIMD of each of its waves, and which workgroups share a CU.
// hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o tools/hwid_probe && tools/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k_probe(unsigned* out, int spin) {
    extern __shared__ double lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(32);     // stay resident: every workgroup of the launch is placed before any leaves
    if (lds[threadIdx.x] < 0) out[0] = 0;
}
int main() {
    const int G = 512;
    unsigned* d; hipMalloc(&d, G * 4 * 2 * 4);
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
    hipLaunchKernelGGL(k_probe, dim3(G), dim3(256), 77 * 1024, 0, d, 20000);   // 200 us (100 MHz clock)
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 8);
    hipMemcpy(h.data(), d, G * 32, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;      // (xcc, se, sh, cu) -> workgroups
    int rr = 0, same_slot = 0;
    for (int b = 0; b < G; ++b) {
        unsigned s[4], ws[4];
        for (int w = 0; w < 4; ++w) { s[w] = (h[(b * 4 + w) * 2] >> 4) & 3; ws[w] = h[(b * 4 + w) * 2] & 15; }
        const unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
        const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 15);
        cu[key].push_back(b);
        if (s[0] == 0 && s[1] == 1 && s[2] == 2 && s[3] == 3) ++rr;
        if (ws[0] == ws[1] && ws[1] == ws[2] && ws[2] == ws[3]) ++same_slot;
        if (b < 12 || (b >= 256 && b < 264))
            printf("wg %3d: xcc %u se %u cu %2u tg %2u | simd of waves 0-3: %u %u %u %u | wave slots %u %u %u %u\n", b, xcc, (hw >> 13) & 7, (hw >> 8) & 15,
                   (hw >> 16) & 15, s[0], s[1], s[2], s[3], ws[0], ws[1], ws[2], ws[3]);
    }
    printf("%d of %d workgroups have wave k on SIMD k; %d have all four waves in the same wave slot; %zu distinct CUs\n", rr, G, same_slot, cu.size());
    int shown = 0, d256 = 0, pairs = 0;
    for (auto& kv : cu) {
        if (kv.second.size() == 2) { ++pairs; if (abs(kv.second[0] - kv.second[1]) == 256) ++d256; }
        if (shown++ < 6) { printf("CU %05x:", kv.first); for (int b : kv.second) printf(" wg %d (tg %u)", b, (h[b * 8] >> 16) & 15); printf("\n"); }
    }
    printf("%d CUs hold two workgroups; %d of those pairs are 256 apart\n", pairs, d256);
    return 0;
}
