// msdr_chain_block.hip -- the block-cadence kernels (one AUDIO_BLOCK per call, Minimal-SDR.ino:518-530) and their launchers.
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "msdr_chain_mfb.hiph"
#include "msdr_chain_q15mb.hiph"
#include "msdr_block.h"

namespace msdr {

hipError_t launch_chain_mfb(hipStream_t stream, int stages, bool am, unsigned grid, unsigned block, size_t lds, const ChainParams &p_in)
{
    ChainParams p = p_in;
#ifdef MSDR_MB_STAMPS
    static unsigned long long *stamp_buf = nullptr;
    const size_t stamp_n = (size_t)grid * (block / 64) * 8;
    if (!stamp_buf && hipMalloc(&stamp_buf, (1u << 20) * 8 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(stamp_buf, 0, stamp_n * 8, stream);
    p.dbg_buf = stamp_buf;
#endif
#define MSDR_MFB(SS, AMF) hipLaunchKernelGGL((chain_mfb_kernel<SS, AMF>), dim3(grid), dim3(block), lds, stream, p)
    switch (stages) {
    case 0: if (am) MSDR_MFB(0, true); else MSDR_MFB(0, false); break;
    case 1: if (am) MSDR_MFB(1, true); else MSDR_MFB(1, false); break;
    case 2: if (am) MSDR_MFB(2, true); else MSDR_MFB(2, false); break;
    default: return hipErrorInvalidValue;
    }
#undef MSDR_MFB
#ifdef MSDR_MB_STAMPS
    if (getenv("MSDR_STAMP_PRINT")) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(stamp_n);
        (void)hipMemcpy(h.data(), stamp_buf, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[8] = {0}; size_t cnt = 0;
        for (size_t u = 0; u < h.size() / 8; u++) { if (!h[u * 8 + 1]) continue; cnt++; for (int k = 0; k < 8; k++) sum[k] += (double)h[u * 8 + k]; }
        if (cnt) fprintf(stderr, "mfb stamps (cycles per wave, %zu waves, tpw %d): issue %.0f | barrier %.0f | land %.0f | burst %.0f | cascade %.0f | park+state %.0f | stores %.0f\n",
                         cnt, p.nseg, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt);
    }
#endif
    return hipGetLastError();
}

hipError_t launch_chain_q15mb(hipStream_t stream, int flavour, bool nodes, unsigned grid, unsigned block, size_t lds, const ChainParams &p)
{
    if (nodes) {
        switch (flavour) {
        case 0: hipLaunchKernelGGL((chain_q15mb_kernel<0, true>), dim3(grid), dim3(block), lds, stream, p); break;
        case 1: hipLaunchKernelGGL((chain_q15mb_kernel<1, true>), dim3(grid), dim3(block), lds, stream, p); break;
        case 2: hipLaunchKernelGGL((chain_q15mb_kernel<2, true>), dim3(grid), dim3(block), lds, stream, p); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (flavour) {
    case 0: hipLaunchKernelGGL((chain_q15mb_kernel<0>), dim3(grid), dim3(block), lds, stream, p); break;
    case 1: hipLaunchKernelGGL((chain_q15mb_kernel<1>), dim3(grid), dim3(block), lds, stream, p); break;
    case 2: hipLaunchKernelGGL((chain_q15mb_kernel<2>), dim3(grid), dim3(block), lds, stream, p); break;
    case 3: hipLaunchKernelGGL((chain_q15mb_kernel<3>), dim3(grid), dim3(block), lds, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace msdr
