// msdr_chain_block.hip -- the block-cadence kernels (one AUDIO_BLOCK per call, Minimal-SDR.ino:518-530) and their launchers.
#include <type_traits>
#include "msdr_chain_mfb.hiph"
#include "msdr_block.h"

namespace msdr {

hipError_t launch_chain_mfb(hipStream_t stream, int stages, bool am, unsigned grid, unsigned block, size_t lds, const ChainParams &p)
{
#define MSDR_MFB(SS, AMF) hipLaunchKernelGGL((chain_mfb_kernel<SS, AMF>), dim3(grid), dim3(block), lds, stream, p)
    switch (stages) {
    case 0: if (am) MSDR_MFB(0, true); else MSDR_MFB(0, false); break;
    case 1: if (am) MSDR_MFB(1, true); else MSDR_MFB(1, false); break;
    case 2: if (am) MSDR_MFB(2, true); else MSDR_MFB(2, false); break;
    default: return hipErrorInvalidValue;
    }
#undef MSDR_MFB
    return hipGetLastError();
}

}  // namespace msdr
