// msdr_block.h -- launchers of the block-cadence kernels (msdr_chain_block.hip: its own translation unit, so that a change to these
// kernels does not rebuild msdr_api.hip and vice versa).  Host-side geometry helpers live with the kernels' headers.
#pragma once
#include <hip/hip_runtime.h>
#include "msdr_shared.h"

namespace msdr {
// chain_mfb_kernel<S, AM> (msdr_chain_mfb.hiph): stages = 0, 1, 2; am = the workgroups' tables are envelope tables.  Returns the HIP error of the launch.
hipError_t launch_chain_mfb(hipStream_t stream, int stages, bool am, unsigned grid, unsigned block, size_t lds_bytes, const ChainParams &p);
// chain_q15mb_kernel<FLAVOUR> (msdr_chain_q15mb.hiph): 0 = LSB / USB channels, 1 = envelope (sqrtf), 2 = envelope (arm_sqrt_q31)
hipError_t launch_chain_q15mb(hipStream_t stream, int flavour, unsigned grid, unsigned block, size_t lds_bytes, const ChainParams &p);
}  // namespace msdr
