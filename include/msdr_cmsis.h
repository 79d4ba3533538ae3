/* msdr_cmsis.h -- the reference's CMSIS-DSP entry points on this path with their OWN argument lists, so that a sketch's calls are
 * relinked instead of rewritten (kernel-function face of the boundary):
 *
 *   arm_fir_init_q15 / arm_fir_fast_q15                 src/CMSIS_5/arm_math.h:1106-1128, sources arm_fir_init_q15.c:78-138,
 *                                                       arm_fir_fast_q15.c:60-329, called at Minimal-SDR.ino:574-575, 906-927
 *   arm_fir_init_f32 / arm_fir_f32                      arm_math.h:1182-1202 (prototypes only in the reference)
 *   arm_biquad_cascade_df1_init_f32 / ..._df1_f32       arm_math.h:1333-1351 (prototypes only)
 *
 * What changes for the caller, and nothing else:
 *   * msdr_cmsis_bind(ctx, channels) once: the CMSIS signatures carry neither a device nor a batch width.  Every call then works
 *     on a BLOCK BATCH -- pSrc / pDst are DEVICE pointers to [channels][blockSize] samples (msdr_malloc), one filter state per
 *     channel kept by the library in HBM.  channels = 1 is the reference's shape.
 *   * the instance struct is the reference's (same fields).  The caller-owned pCoeffs stay the caller's, as in CMSIS (the instance
 *     holds the pointer, arm_fir_init_q15.c:100-109): every process call compares the array with the bytes its device tables were
 *     built from and, if the caller has rewritten it in place -- the bandwidth menu does, with no init_FIR(), UI.cpp:337-345 +
 *     Minimal-SDR.ino:221-223 -- rebuilds the tables and carries on with the filter state kept (numTaps host compares per call; the
 *     biquad cascade keeps arm_biquad_cascade_df1_f32's pState semantics across the change, msdr_biquad_df1_f32_set_coeffs).  A
 *     re-run of the init zeroes the state, as init_FIR() does on every retune (Minimal-SDR.ino:901-930).  The caller-owned pState
 *     is cleared as the reference's init does and is otherwise unused (the state lives on the device).
 *   * a rebuild that fails (a cascade whose state cannot be carried to the new coefficients, an allocation) leaves the filter on its old
 *     tables for this call -- pDst is still written -- and is tried again on the next call; msdr_last_error() has the text.
 *   * errors: the init keeps arm_fir_init_q15's contract (odd numTaps -> ARM_MATH_ARGUMENT_ERROR, instance left untouched,
 *     arm_fir_init_q15.c:93-96); the void process functions cannot report anything -- msdr_last_error() has the text.
 * Define MSDR_CMSIS_NAMES before including this header to get the arm_* names themselves as macros. */
#ifndef MSDR_CMSIS_H
#define MSDR_CMSIS_H

#include "msdr.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {                                   /* arm_math.h:404-413 */
    MSDR_ARM_MATH_SUCCESS = 0, MSDR_ARM_MATH_ARGUMENT_ERROR = -1, MSDR_ARM_MATH_LENGTH_ERROR = -2, MSDR_ARM_MATH_SIZE_MISMATCH = -3
} msdr_arm_status;

typedef struct { uint16_t numTaps; q15_t *pState; q15_t *pCoeffs; } msdr_arm_fir_instance_q15;               /* arm_math.h:1027-1032 */
typedef struct { uint16_t numTaps; float32_t *pState; float32_t *pCoeffs; } msdr_arm_fir_instance_f32;       /* arm_math.h:1047-1052 */
typedef struct { uint32_t numStages; float32_t *pState; float32_t *pCoeffs; } msdr_arm_biquad_casd_df1_inst_f32;   /* arm_math.h:1230-1235 */

/* the context and batch width the shims below work with; NULL unbinds and frees every instance created through them */
int msdr_cmsis_bind(msdr_ctx *ctx, uint32_t channels);
/* The same with pSrc / pDst as HOST arrays of [channels][blockSize] samples -- what the sketch passes: `arm_fir_fast_q15(&FIR_I, I_buffer,
 * I_FIR_out, AUDIO_BLOCK_SAMPLES)` on stack arrays (Minimal-SDR.ino:525-526, 574-575); channels = 1 is exactly that call.  Every process
 * call stages the block through two device buffers on the context's stream and returns when pDst holds the result (a PCIe round trip per
 * call: the drop-in for a sketch that keeps its buffers where they are; a caller that cares for throughput keeps them on the device and
 * binds with msdr_cmsis_bind). */
int msdr_cmsis_bind_host(msdr_ctx *ctx, uint32_t channels);

msdr_arm_status msdr_arm_fir_init_q15(msdr_arm_fir_instance_q15 *S, uint16_t numTaps, q15_t *pCoeffs, q15_t *pState, uint32_t blockSize);
void msdr_arm_fir_fast_q15(const msdr_arm_fir_instance_q15 *S, q15_t *pSrc, q15_t *pDst, uint32_t blockSize);
void msdr_arm_fir_init_f32(msdr_arm_fir_instance_f32 *S, uint16_t numTaps, float32_t *pCoeffs, float32_t *pState, uint32_t blockSize);
void msdr_arm_fir_f32(const msdr_arm_fir_instance_f32 *S, float32_t *pSrc, float32_t *pDst, uint32_t blockSize);
void msdr_arm_biquad_cascade_df1_init_f32(msdr_arm_biquad_casd_df1_inst_f32 *S, uint8_t numStages, float32_t *pCoeffs, float32_t *pState);
void msdr_arm_biquad_cascade_df1_f32(const msdr_arm_biquad_casd_df1_inst_f32 *S, float32_t *pSrc, float32_t *pDst, uint32_t blockSize);

#ifdef MSDR_CMSIS_NAMES
#define arm_status msdr_arm_status
#define ARM_MATH_SUCCESS MSDR_ARM_MATH_SUCCESS
#define ARM_MATH_ARGUMENT_ERROR MSDR_ARM_MATH_ARGUMENT_ERROR
#define arm_fir_instance_q15 msdr_arm_fir_instance_q15
#define arm_fir_instance_f32 msdr_arm_fir_instance_f32
#define arm_biquad_casd_df1_inst_f32 msdr_arm_biquad_casd_df1_inst_f32
#define arm_fir_init_q15 msdr_arm_fir_init_q15
#define arm_fir_fast_q15 msdr_arm_fir_fast_q15
#define arm_fir_init_f32 msdr_arm_fir_init_f32
#define arm_fir_f32 msdr_arm_fir_f32
#define arm_biquad_cascade_df1_init_f32 msdr_arm_biquad_cascade_df1_init_f32
#define arm_biquad_cascade_df1_f32 msdr_arm_biquad_cascade_df1_f32
#endif

#ifdef __cplusplus
}
#endif
#endif
