// msdr_chain_stream.hip -- the long-call chain kernels and their launchers (their own translation unit: most of the library's device code).
#include <type_traits>
#include "msdr_kernels.hiph"
#include "msdr_chain_fold.hiph"
#include "msdr_chain_mfma.hiph"
#include "msdr_chain_mfw.hiph"
#include "msdr_chain_q15mf.hiph"
#include "msdr_fir_f32mf.hiph"
#include "msdr_fir_f32tr.hiph"
#include "msdr_chain_amtr.hiph"
#include "msdr_block.h"

namespace msdr {

hipError_t launch_chain_mfw(hipStream_t stream, int stages, bool am, bool fold, bool fr, unsigned grid, unsigned block, size_t lds, const ChainParams &q)
{
    // SSB-table units and envelope-table units are separate launches of separate kernels (register allocation per flavour); the cascade as
    // matrix products (fold) exists for one and two sections
#define MSDR_MFW_LAUNCH(SS, AMF, FO) do { if (fr) hipLaunchKernelGGL((chain_mfw_kernel<SS, AMF, FO, true>), dim3(grid), dim3(block), lds, stream, q); \
                                          else hipLaunchKernelGGL((chain_mfw_kernel<SS, AMF, FO, false>), dim3(grid), dim3(block), lds, stream, q); } while (0)
#define MSDR_MFW_PLAIN(SS) do { if (!am) MSDR_MFW_LAUNCH(SS, false, false); else MSDR_MFW_LAUNCH(SS, true, false); } while (0)
#define MSDR_MFW_FOLDS(SS) do { if (!fold) MSDR_MFW_PLAIN(SS); else if (!am) MSDR_MFW_LAUNCH(SS, false, true); else MSDR_MFW_LAUNCH(SS, true, true); } while (0)
    switch (stages) {
    case 0: MSDR_MFW_PLAIN(0); break;
    case 1: MSDR_MFW_FOLDS(1); break;
    case 2: MSDR_MFW_FOLDS(2); break;
    case 3: MSDR_MFW_PLAIN(3); break;
    default: MSDR_MFW_PLAIN(4); break;
    }
#undef MSDR_MFW_FOLDS
#undef MSDR_MFW_PLAIN
#undef MSDR_MFW_LAUNCH
    return hipGetLastError();
}

hipError_t launch_chain_amtr(hipStream_t stream, int ns, int stages, unsigned grid, unsigned block, size_t lds, const ChainParams &a)
{
#define MSDR_AT_LAUNCH(NS_, SS_) hipLaunchKernelGGL((chain_amtr_kernel<NS_, SS_>), dim3(grid), dim3(block), lds, stream, a)
#define MSDR_AT_STAGES(NS_) switch (stages) { case 0: MSDR_AT_LAUNCH(NS_, 0); break; case 1: MSDR_AT_LAUNCH(NS_, 1); break; case 2: MSDR_AT_LAUNCH(NS_, 2); break; \
                                               case 3: MSDR_AT_LAUNCH(NS_, 3); break; default: MSDR_AT_LAUNCH(NS_, 4); break; }
    switch (ns) {
    case 2: MSDR_AT_STAGES(2) break;
    case 3: MSDR_AT_STAGES(3) break;
    case 4: MSDR_AT_STAGES(4) break;
    default: MSDR_AT_STAGES(5) break;
    }
#undef MSDR_AT_STAGES
#undef MSDR_AT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_chain_fold(hipStream_t stream, int period, unsigned grid, size_t lds, const ChainParams &p)
{
    if (period == 4) hipLaunchKernelGGL((chain_fold_kernel<4>), dim3(grid), dim3(kThreads), lds, stream, p);
    else if (period == 2) hipLaunchKernelGGL((chain_fold_kernel<2>), dim3(grid), dim3(kThreads), lds, stream, p);
    else hipLaunchKernelGGL((chain_fold_kernel<1>), dim3(grid), dim3(kThreads), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_chain_generic(hipStream_t stream, bool q15, unsigned grid, size_t lds, const ChainParams &p)
{
    if (q15) hipLaunchKernelGGL((chain_kernel<ArithQ15>), dim3(grid), dim3(kThreads), lds, stream, p);
    else hipLaunchKernelGGL((chain_kernel<ArithF32>), dim3(grid), dim3(kThreads), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_chain_q15mf(hipStream_t stream, int flavour, bool fr, unsigned grid, unsigned block, size_t lds, const ChainParams &q)
{
#define MSDR_QM_LAUNCH(FL) do { if (fr) hipLaunchKernelGGL((chain_q15mf_kernel<FL, true>), dim3(grid), dim3(block), lds, stream, q); \
                                else hipLaunchKernelGGL((chain_q15mf_kernel<FL, false>), dim3(grid), dim3(block), lds, stream, q); } while (0)
    switch (flavour) {
    case 0: MSDR_QM_LAUNCH(0); break;
    case 1: MSDR_QM_LAUNCH(1); break;
    case 2: MSDR_QM_LAUNCH(2); break;
    default: hipLaunchKernelGGL((chain_q15mf_kernel<3, false>), dim3(grid), dim3(block), lds, stream, q); break;      // the arm_fir_fast_q15 stage: no mixer, no full-rate layout
    }
#undef MSDR_QM_LAUNCH
    return hipGetLastError();
}

}  // namespace msdr
