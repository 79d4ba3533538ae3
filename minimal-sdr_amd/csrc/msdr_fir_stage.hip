// msdr_fir_stage.hip -- the arm_fir_f32 stage's matrix-core kernels and their launchers (their own translation unit).
#include <type_traits>
#include "msdr_fir_f32mf.hiph"
#include "msdr_fir_f32tr.hiph"
#include "msdr_fir_f32tq.hiph"
#include "msdr_block.h"

namespace msdr {

hipError_t launch_fir_f32tq(hipStream_t stream, int ns, bool skip1, unsigned grid, size_t lds, const TqParams &q)
{
#define MSDR_TQ_LAUNCH(NS_) case NS_: \
        if (skip1) hipLaunchKernelGGL((fir_f32tq_kernel<NS_, true>), dim3(grid), dim3(256), lds, stream, q); \
        else hipLaunchKernelGGL((fir_f32tq_kernel<NS_, false>), dim3(grid), dim3(256), lds, stream, q); \
        break;
    switch (ns) {
        MSDR_TQ_LAUNCH(2) MSDR_TQ_LAUNCH(3) MSDR_TQ_LAUNCH(4) MSDR_TQ_LAUNCH(5) MSDR_TQ_LAUNCH(6)
        MSDR_TQ_LAUNCH(7) MSDR_TQ_LAUNCH(8) MSDR_TQ_LAUNCH(9) MSDR_TQ_LAUNCH(10)
        default: return hipErrorInvalidValue;
    }
#undef MSDR_TQ_LAUNCH
    return hipGetLastError();
}

hipError_t launch_fir_f32mf(hipStream_t stream, unsigned grid, unsigned block, size_t lds, const float *x, float *y, const float *hist, const char *tab,
                            long long n, int channels, int nseg, long long seg_len, int hist_len, int halo, int nsteps, int nw)
{
    hipLaunchKernelGGL(fir_f32mf_kernel, dim3(grid), dim3(block), lds, stream, x, y, hist, tab, n, channels, nseg, seg_len, hist_len, halo, nsteps, nw);
    return hipGetLastError();
}

}  // namespace msdr
