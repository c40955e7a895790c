// k_transport_scan: one order of transport (spec:326-449) with the chunks of a sweep dealt to several waves -- the launchers.
// The kernel's body is transport_scan_body.hpp (shared with the order-loop kernel, order_loop.hip).
#include "transport_scan_body.hpp"

namespace sosrt {

namespace {

template <bool ACC, bool SAVED, bool SPLIT, int NC = 0, bool MZ = false, bool WIDE = false>
__global__ __launch_bounds__(768) void k_transport_scan(TransportArgs a, int fixcap) {
    // SPLIT: ceil(N / 64) workgroups per column (two at N = 128, four at N = 256), part p = blockIdx.x mod that
    const int nparts = SPLIT ? ((NC ? NC : a.g.N) + 63) >> 6 : 1;
    const int part = SPLIT ? (int)(blockIdx.x % nparts) : 0;
    int b = SPLIT ? (int)(blockIdx.x / nparts) : (int)blockIdx.x;
    if (ACC && a.live > 0) {
        b = a.live_list[b];
        if (b < 0) return;                          // fewer live columns than the host's (lagging) count
    } else if (ACC && !a.cv.active[b]) {
        return;
    }
    (void)transport_scan_order<ACC, SAVED, SPLIT, NC, MZ, false, WIDE>(a, fixcap, b, part, ScanFused());
}

// shapes of the split form that take its WIDE instantiation (transport_scan_body.hpp): odd N, N > 256, more than 64 chunks per sweep
inline bool scan_wide(const Grid& g) { return (g.N & 1) || g.N > 256 || (g.L + TC - 1) / TC > 64; }

// the WIDE instantiation of the split form (with or without saved orders; columns of three zones, or of up to a.nzcap: ZoneRows<true>)
void launch_scan_wide(hipStream_t s, dim3 grid, const TransportArgs& a) {
    using C = ScanCfg<true, true>;
    const dim3 block((C::SW + NLOAD) * 64);
    const size_t shm = scan_lds_bytes<true, true>(a.g, a.nzcap);
#define SOSRT_SCAN_LAUNCH_W(SAVED_, MZ_)                                                                       \
    do {                                                                                                       \
        auto kern = k_transport_scan<true, SAVED_, true, 0, MZ_, true>;                                        \
        static PerDeviceOnce big_lds;                                                                                 \
        if (big_lds.first()) {                                                                                        \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)kScanLdsBytes);                                                           \
        }                                                                                                      \
        hipLaunchKernelGGL(kern, grid, block, shm, s, a, scan_fixcap(a.g));                                    \
    } while (0)
    const bool mz = a.nzcap > kRingZones;
    if (a.saved && mz) SOSRT_SCAN_LAUNCH_W(true, true);
    else if (a.saved) SOSRT_SCAN_LAUNCH_W(true, false);
    else if (mz) SOSRT_SCAN_LAUNCH_W(false, true);
    else SOSRT_SCAN_LAUNCH_W(false, false);
#undef SOSRT_SCAN_LAUNCH_W
}

template <bool SPLIT>
void launch_scan_t(hipStream_t s, dim3 grid, const TransportArgs& a) {
    using C = ScanCfg<SPLIT>;
    const int nwc = SPLIT ? 1 : (a.g.N + 63) / 64;
    const dim3 block((nwc * C::SW + NLOAD) * 64);
    const size_t shm = scan_lds_bytes<SPLIT>(a.g, a.nzcap);
#define SOSRT_SCAN_LAUNCH_Z(ACC_, SAVED_, NC_, MZ_)                                                            \
    do {                                                                                                       \
        auto kern = k_transport_scan<ACC_, SAVED_, SPLIT, NC_, MZ_>;                                           \
        static PerDeviceOnce big_lds;                                                                                 \
        if (big_lds.first()) {                                                                                        \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)kScanLdsBytes);                                                           \
        }                                                                                                      \
        hipLaunchKernelGGL(kern, grid, block, shm, s, a, scan_fixcap(a.g));                                    \
    } while (0)
#define SOSRT_SCAN_LAUNCH_N(ACC_, SAVED_, NC_) SOSRT_SCAN_LAUNCH_Z(ACC_, SAVED_, NC_, false)
    // (a batch with a column of more than three zones: the instantiation that tests every boundary of the zone table, ZoneRows<true>)
#define SOSRT_SCAN_LAUNCH(ACC_, SAVED_)                                                                        \
    do {                                                                                                       \
        if (a.nzcap > kRingZones) SOSRT_SCAN_LAUNCH_Z(ACC_, SAVED_, 0, true);                                  \
        else SOSRT_SCAN_LAUNCH_Z(ACC_, SAVED_, 0, false);                                                      \
    } while (0)
    static const bool kFixN = !(getenv("SOSRT_SCAN_NC") && atoi(getenv("SOSRT_SCAN_NC")) == 0);      // (A/B: 0 = the generic instantiation)
    if (a.accumulate) {
        if (a.saved) SOSRT_SCAN_LAUNCH(true, true);
        else if (SPLIT && kFixN && a.nzcap <= kRingZones && a.g.N == 128) SOSRT_SCAN_LAUNCH_N(true, false, SPLIT ? 128 : 0);
        else if (SPLIT && kFixN && a.nzcap <= kRingZones && a.g.N == 256) SOSRT_SCAN_LAUNCH_N(true, false, SPLIT ? 256 : 0);
        else SOSRT_SCAN_LAUNCH(true, false);
    } else {
        SOSRT_SCAN_LAUNCH(false, false);
    }
#undef SOSRT_SCAN_LAUNCH
#undef SOSRT_SCAN_LAUNCH_N
#undef SOSRT_SCAN_LAUNCH_Z
}

}  // namespace

// Half rows of one 1-KiB piece (N <= 128, N even: 16-byte lanes), at most 64 chunks per sweep (the mask of the chunks
// with a zone boundary), stages and tables within the LDS of a CU.
bool transport_scan_ok(const Grid& g) {
    if (g.N % 2 || g.N < 4 || g.N > 128) return false;
    if ((g.L + TC - 1) / TC > 64) return false;
    return scan_lds_bytes<false>(g) <= kScanLdsBytes;
}
// ceil(N / 64) workgroups per column: at least two lane groups to deal, at most kScanDirs directions per hemisphere (the exchange
// rows), at most kScanChunks chunks per sweep.  N in (128, 256] has this form only; odd N, N > 256 and L > 512 its WIDE
// instantiation only (the reference's shipped N = 501, L = 800: eight workgroups per column).
bool transport_scan_split_ok(const Grid& g) {
    if (g.N <= 64 || g.N > kScanDirs) return false;
    if (scan_wide(g)) return (g.L + TC - 1) / TC <= kScanChunks && scan_lds_bytes<true, true>(g) <= kScanLdsBytes;
    return scan_lds_bytes<true>(g) <= kScanLdsBytes;
}
int transport_scan_parts(const Grid& g) { return (g.N + 63) / 64; }
// whether the per-zone tables of a batch whose columns have up to nzcap zones still fit beside the stages
bool transport_scan_fits(const Grid& g, int nzcap, bool split) {
    if (split && scan_wide(g)) return scan_lds_bytes<true, true>(g, nzcap) <= kScanLdsBytes;
    return (split ? scan_lds_bytes<true>(g, nzcap) : scan_lds_bytes<false>(g, nzcap)) <= kScanLdsBytes;
}
size_t transport_scan_scratch_doubles() { return kScanScratch; }

// a.scan_split: two workgroups per column (the grid is then twice the columns; specular surface or none; a.scan_scratch /
// a.scan_sync: kScanScratch doubles and two zeroed ints per column of the batch)
void launch_transport_scan(hipStream_t s, dim3 grid, const TransportArgs& a) {
    if (a.scan_split && a.accumulate && scan_wide(a.g)) launch_scan_wide(s, dim3(transport_scan_parts(a.g) * grid.x), a);
    else if (a.scan_split && a.accumulate) launch_scan_t<true>(s, dim3(transport_scan_parts(a.g) * grid.x), a);
    else launch_scan_t<false>(s, grid, a);
}

}  // namespace sosrt
