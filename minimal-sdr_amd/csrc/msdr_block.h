// msdr_block.h -- launchers of the kernels that live in translation units of their own, so that a change to a kernel does not rebuild the
// C-ABI layer (msdr_api.hip: host logic and the small stage kernels) and vice versa:
//   msdr_chain_block.hip   the block-cadence kernels (round 5)
//   msdr_chain_stream.hip  the long-call chain kernels: chain_mfw_kernel (all flavours), chain_amtr_kernel, chain_fold_kernel, chain_kernel<Arith>, chain_q15mf_kernel
//   msdr_fir_stage.hip     the arm_fir_f32 stage: fir_f32tq_kernel, fir_f32mf_kernel
// Host-side geometry helpers (LDS sizes, table formats) live with the kernels' headers.  Every launcher returns the HIP error of its launch.
#pragma once
#include <hip/hip_runtime.h>
#include "msdr_shared.h"

namespace msdr {
// chain_mfb_kernel<S, AM> (msdr_chain_mfb.hiph): stages = 0, 1, 2; am = the workgroups' tables are envelope tables.  Returns the HIP error of the launch.
hipError_t launch_chain_mfb(hipStream_t stream, int stages, bool am, unsigned grid, unsigned block, size_t lds_bytes, const ChainParams &p);
// chain_q15mb_kernel<FLAVOUR> (msdr_chain_q15mb.hiph): 0 = LSB / USB channels, 1 = envelope (sqrtf), 2 = envelope (arm_sqrt_q31), 3 = the arm_fir_fast_q15 stage alone
// nodes: the two AudioFilterBiquad nodes as the kernel's second phase (p.bq_state / p.bq_state_out = their records; one tile per wave, >= 3 waves, n = 128)
hipError_t launch_chain_q15mb(hipStream_t stream, int flavour, bool nodes, unsigned grid, unsigned block, size_t lds_bytes, const ChainParams &p);
// ---- msdr_chain_stream.hip ----
hipError_t launch_chain_mfw(hipStream_t stream, int stages, bool am, bool fold, bool full_rate, unsigned grid, unsigned block, size_t lds_bytes, const ChainParams &p);
hipError_t launch_chain_amtr(hipStream_t stream, int ns, int stages, unsigned grid, unsigned block, size_t lds_bytes, const ChainParams &p);
hipError_t launch_chain_fold(hipStream_t stream, int period, unsigned grid, size_t lds_bytes, const ChainParams &p);           // period 1, 2, 4
hipError_t launch_chain_generic(hipStream_t stream, bool q15, unsigned grid, size_t lds_bytes, const ChainParams &p);          // chain_kernel<ArithF32 / ArithQ15>
hipError_t launch_chain_q15mf(hipStream_t stream, int flavour, bool full_rate, unsigned grid, unsigned block, size_t lds_bytes, const ChainParams &p);   // flavour 0..3 (3: arm_fir_fast_q15 stage)
// ---- msdr_fir_stage.hip ----
struct TqParams;
hipError_t launch_fir_f32tq(hipStream_t stream, int ns, bool skip1, unsigned grid, size_t lds_bytes, const TqParams &q);
hipError_t launch_fir_f32mf(hipStream_t stream, unsigned grid, unsigned block, size_t lds_bytes, const float *x, float *y, const float *hist, const char *tab,
                            long long n, int channels, int nseg, long long seg_len, int hist_len, int halo, int nsteps, int nw);
}  // namespace msdr
