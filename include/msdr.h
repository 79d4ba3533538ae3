/*
 * include/msdr.h -- C ABI of the MI355X-native Minimal-SDR demodulation chain.
 *
 * Drop-in boundary for the hot path of FrankBoesing/Minimal-SDR (citations relative to the
 * reference tree): IF samples -> I/Q mix -> FIR pair -> AM/SSB demod -> cascaded IIR biquad.
 * Plain C: pointers and sizes only; no C++/torch types.  Every function returns an msdr_status
 * (0 = OK; negative values mirror CMSIS `arm_status`, src/CMSIS_5/arm_math.h:404-413) and never
 * throws.  The library never takes ownership of caller buffers.
 *
 * Memory convention: pointers named `d_*` are DEVICE pointers (hipMalloc / msdr_malloc /
 * torch tensor data_ptr) valid on the context's GPU; all other pointers are HOST pointers that
 * are read during the call and not retained (the reference's CMSIS functions retain the caller's
 * coefficient/state pointers, arm_fir_init_q15.c:100-109 -- here the library copies coefficients
 * to the device and owns the state, which lives in HBM between calls).
 *
 * Batch layout in HBM: a "block batch" is [channels][block_len] row-major, one contiguous time
 * series per receiver channel (the reference's unit is one audio_block_t = int16[128] of ONE
 * channel, freq_conv.cpp:70, filter_biquad.cpp:42).  Channels are independent.
 *
 * All work is enqueued on the context's HIP stream and is asynchronous with respect to the
 * host; msdr_ctx_synchronize() (or synchronising the stream you supplied) waits for it.
 * There is NO CPU fallback: if no HIP device is usable every entry point fails with
 * MSDR_STATUS_NO_DEVICE.
 */
#ifndef MSDR_H
#define MSDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDR_VERSION_MAJOR 0
#define MSDR_VERSION_MINOR 1

typedef int16_t q15_t;     /* arm_math.h:423 */
typedef float float32_t;   /* arm_math.h:438 */

/* ---- status: arm_status values (arm_math.h:404-413) + device errors ------------------- */
typedef enum {
    MSDR_STATUS_SUCCESS        = 0,   /* ARM_MATH_SUCCESS */
    MSDR_STATUS_ARGUMENT_ERROR = -1,  /* ARM_MATH_ARGUMENT_ERROR (e.g. odd numTaps, arm_fir_init_q15.c:93-96) */
    MSDR_STATUS_LENGTH_ERROR   = -2,  /* ARM_MATH_LENGTH_ERROR */
    MSDR_STATUS_SIZE_MISMATCH  = -3,  /* ARM_MATH_SIZE_MISMATCH */
    MSDR_STATUS_NO_DEVICE      = -100,/* no usable HIP device / wrong architecture */
    MSDR_STATUS_HIP_ERROR      = -101,/* a HIP runtime call failed: see msdr_last_error() */
    MSDR_STATUS_OUT_OF_MEMORY  = -102
} msdr_status;

/* ---- demodulator modes: stations.h:4  enum { SYNCAM, AM, LSB, USB, CW } ---------------- */
enum { MSDR_MODE_SYNCAM = 0, MSDR_MODE_AM = 1, MSDR_MODE_LSB = 2, MSDR_MODE_USB = 3, MSDR_MODE_CW = 4 };
/* AM/CW magnitude: Minimal-SDR.ino:606-616 (Teensy 3.6: sqrtf) or :618-627 (Teensy 3.2: arm_sqrt_q31>>16) */
enum { MSDR_SQRT_F32 = 0, MSDR_SQRT_Q31 = 1 };
/* mixer in front of the FIR pair */
enum {
    MSDR_MIXER_FS4 = 0,   /* multiplication-free Fs/4 mixer, Minimal-SDR.ino:546-558 */
    MSDR_MIXER_NCO = 1    /* AudioEffectFreqConv, freq_conv.cpp:30-116: IF on port 0, zeros on port 1, dir = 1 */
};
/* arithmetic flavour of a chain */
enum {
    MSDR_ARITH_Q15 = 0,   /* the reference as written: arm_fir_fast_q15 + integer demod + Teensy biquad; bit-exact */
    MSDR_ARITH_F32 = 1    /* arm_fir_f32 / arm_biquad_cascade_df1_f32 semantics, int16 in, fp32 out */
};
#define MSDR_AUDIO_BLOCK_SAMPLES 128           /* Teensy core default (Minimal-SDR.ino:525) */
#define MSDR_AUDIO_SAMPLE_RATE_EXACT 44117.64706 /* Teensy core constant used by filter_biquad.h:58 */
#define MSDR_MAX_BIQUAD_STAGES 4               /* filter_biquad.cpp:86 */
#define MSDR_MAX_TAPSETS 8

/* ======================================================================================
 * Context: one GPU + one HIP stream.
 * ====================================================================================== */
typedef struct msdr_ctx msdr_ctx;

/* device: HIP device ordinal.  hip_stream: a hipStream_t to enqueue on (e.g. torch's current
 * stream), or NULL to let the context create and own one. */
int  msdr_ctx_create(int device, void *hip_stream, msdr_ctx **out);
int  msdr_ctx_destroy(msdr_ctx *ctx);
int  msdr_ctx_synchronize(msdr_ctx *ctx);
void *msdr_ctx_stream(msdr_ctx *ctx);                       /* the hipStream_t in use */
const char *msdr_last_error(void);                          /* thread-local text of the last failure */
const char *msdr_version(void);
const char *msdr_build_rev(void);                           /* source revision the library was built from ("<git hash>[+dirty]"), for measurement records */
int  msdr_device_count(void);                               /* usable gfx950 devices; never initialises one */

/* device-memory helpers (callers may equally pass pointers from hipMalloc or a torch tensor) */
int msdr_malloc(msdr_ctx *ctx, size_t bytes, void **d_ptr);
int msdr_free(msdr_ctx *ctx, void *d_ptr);
int msdr_memcpy_h2d(msdr_ctx *ctx, void *d_dst, const void *src, size_t bytes);   /* stream-ordered, returns after the copy */
/* Optional: HIP events on the context's stream around the MAIN kernel of each msdr_fir_q15_process / msdr_fir_f32_process call
 * (not its history kernel), for roofline arithmetic; msdr_ctx_get_kernel_time synchronises the stream and returns the sum of
 * the durations and the number of launches since the last reset.  (The fused chain has its own pair: msdr_chain_enable_timing.) */
int msdr_ctx_enable_kernel_timing(msdr_ctx *ctx, int on);
int msdr_ctx_get_kernel_time(msdr_ctx *ctx, double *total_ms, uint64_t *launches, int reset);
int msdr_memcpy_d2h(msdr_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int msdr_memcpy_d2d(msdr_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);     /* stream-ordered, asynchronous */
int msdr_memset(msdr_ctx *ctx, void *d_dst, int value, size_t bytes);

/* ======================================================================================
 * The path's one exchange step: gathering demodulated audio across the GPUs of a node (one
 * process per GPU; channels are sharded).  RCCL over xGMI, driven from C; RCCL is
 * loaded on first use (dlopen of librccl.so.1), so a single-GPU application does not need it.
 *   rank 0: msdr_comm_get_unique_id(id) -> hand the 128 bytes to the other ranks by any host channel
 *   all   : msdr_comm_create(ctx, id, rank, world, &comm)
 *   per block k (two or more audio buffers, slot = k % MSDR_GATHER_SLOTS):
 *           msdr_chain_process(..., audio[k % 2], ...);                    compute on the context's stream
 *           msdr_gather_audio_begin(comm, slot, audio[k % 2], bytes, d_all, root);   asynchronous, on the communicator's own
 *                                                                          stream, ordered after the compute queued so far
 *           ... block k + 1 is demodulated meanwhile ...
 *           msdr_gather_audio_wait(comm, slot, 0) before anything reuses audio[k % 2] / reads d_all (device-side wait on the
 *                                                 context's stream; host_wait = 1 blocks the calling thread instead)
 * root >= 0: only that rank receives, d_recv = [world][local_bytes] in rank order (peers send over one link each);
 * root < 0 : every rank receives (all-gather).  Every rank passes the same local_bytes (pad the last shard).
 * ====================================================================================== */
#define MSDR_GATHER_SLOTS 4
typedef struct msdr_comm msdr_comm;
int msdr_comm_get_unique_id(void *id128);
int msdr_comm_create(msdr_ctx *ctx, const void *id128, int rank, int world, msdr_comm **out);
int msdr_gather_audio_begin(msdr_comm *comm, int slot, const void *d_local, size_t local_bytes, void *d_recv, int root);
int msdr_gather_audio_wait(msdr_comm *comm, int slot, int host_wait);
int msdr_comm_destroy(msdr_comm *comm);

/* ======================================================================================
 * Host-side designers (setup path; pure CPU arithmetic, no device needed).
 * ====================================================================================== */
/* calc_FIR_coeffs, Minimal-SDR.ino:782-872 (with m_sinc :874-881, Izero :883-899).  Same
 * argument list and same overrun behaviour as the reference: type 0 writes numCoeffs entries,
 * types 2/3 write numCoeffs+1, type 4 (Hilbert) writes up to 2*numCoeffs+2. */
void msdr_calc_FIR_coeffs(int16_t *coeffs, int numCoeffs, float32_t fc, float32_t Astop, int type,
                          float dfc, float Fsamprate);
/* The sketch takes `PI` from outside (PIH = PI / 2, .ino:779).  msdr_calc_FIR_coeffs evaluates it as the vendored CMSIS
 * header's float fallback (src/CMSIS_5/arm_math.h:365-367); on a Teensy, Arduino.h's double literal is defined first and
 * `m * PIH` etc. are then evaluated in double, which moves single taps by 1 LSB after the truncation to int16.
 * msdr_calc_FIR_coeffs_pid is that variant.  Both are bit-exact against the compiled reference built with the matching PI. */
void msdr_calc_FIR_coeffs_pid(int16_t *coeffs, int numCoeffs, float32_t fc, float32_t Astop, int type,
                              float dfc, float Fsamprate);
/* AudioFilterBiquad::setLowpass/.../setHighShelf, src/Audio/filter_biquad.h:56-149.
 * coef[5] = {b0,b1,b2,a1,a2} scaled by 2^30 in textbook sign, exactly what the reference hands
 * to setCoefficients(stage, const int*).  sample_rate: the reference hard-codes
 * AUDIO_SAMPLE_RATE_EXACT (callers multiply cut-offs by CORR_FACT, Minimal-SDR.ino:86,391). */
enum { MSDR_BQ_LOWPASS = 0, MSDR_BQ_HIGHPASS, MSDR_BQ_BANDPASS, MSDR_BQ_NOTCH, MSDR_BQ_LOWSHELF, MSDR_BQ_HIGHSHELF };
int msdr_biquad_design(int kind, float frequency, float q_or_gain, float slope, double sample_rate, int32_t coef[5]);

/* ======================================================================================
 * Kernel-function API: CMSIS-DSP / Teensy-Audio mirrors, batched over channels.
 * Instances are opaque handles (state lives in HBM); `blockSize` may differ between calls.
 * d_src/d_dst: [channels][blockSize].  In-place (d_dst == d_src) is allowed where the
 * reference allows it.
 * ====================================================================================== */

/* arm_fir_init_q15 / arm_fir_fast_q15 (arm_math.h:1106-1128; arm_fir_init_q15.c:78-138,
 * arm_fir_fast_q15.c:60-329).  pCoeffs: host, numTaps entries in CMSIS (time-reversed) order,
 * shared by all channels.  Odd numTaps -> MSDR_STATUS_ARGUMENT_ERROR like the reference.
 * Creation zeroes the state (arm_fir_init_q15.c:106). */
typedef struct msdr_fir_q15 msdr_fir_q15;
int msdr_fir_q15_create(msdr_ctx *ctx, uint16_t numTaps, const q15_t *pCoeffs, uint32_t channels, msdr_fir_q15 **out);
int msdr_fir_q15_process(msdr_fir_q15 *S, const q15_t *d_src, q15_t *d_dst, uint32_t blockSize);
int msdr_fir_q15_reset(msdr_fir_q15 *S);
/* New coefficients for a running filter, state kept: what rewriting the caller-owned array behind S->pCoeffs does in the reference
 * (the instance only holds the pointer, arm_fir_init_q15.c:100-109; the bandwidth menu rewrites FIR_AM_coeffs in place with no
 * init_FIR(), UI.cpp:337-345, Minimal-SDR.ino:221-223).  Same numTaps as at creation.  Stream-ordered behind the calls queued so far. */
int msdr_fir_q15_set_coeffs(msdr_fir_q15 *S, const q15_t *pCoeffs);
int msdr_fir_q15_destroy(msdr_fir_q15 *S);

/* arm_fir_init_f32 / arm_fir_f32 (prototypes arm_math.h:1182-1202; CMSIS-DSP V1.5.x). Any numTaps >= 1. */
typedef struct msdr_fir_f32 msdr_fir_f32;
int msdr_fir_f32_create(msdr_ctx *ctx, uint16_t numTaps, const float32_t *pCoeffs, uint32_t channels, msdr_fir_f32 **out);
/* (The 16..~290-tap kernel deals its tiles from a queue that the launch itself leaves as it found it -- its last wave out re-zeroes the
 * counters --, so a launch can be replayed; what alternates from call to call is the pair of history buffers, as in the chain: a HIP graph
 * a caller records over this stage should hold an even number of consecutive calls.  msdr_fir_f32_kernel_name reports the kernel for
 * blocks below 2^31 tiles x channels; beyond that the one-stream-per-wave kernel runs.) */
int msdr_fir_f32_process(msdr_fir_f32 *S, const float32_t *d_src, float32_t *d_dst, uint32_t blockSize);
int msdr_fir_f32_reset(msdr_fir_f32 *S);
int msdr_fir_f32_set_coeffs(msdr_fir_f32 *S, const float32_t *pCoeffs);   /* as msdr_fir_q15_set_coeffs: state kept, same numTaps */
const char *msdr_fir_f32_kernel_name(msdr_fir_f32 *S);      /* the kernel msdr_fir_f32_process launches for this instance */
/* Filters of 16..513 taps run on the matrix cores with the samples as two fp16 pieces (22 bits) after a power-of-two scale.  By
 * default the scale is chosen per 1024-output tile from the data (block floating point): nothing to declare.  max_abs > 0 pins
 * one scale for samples below that magnitude (saves the per-tile maximum; samples above it would overflow); 0 = automatic again.
 * No counterpart in CMSIS (arm_fir_f32 is plain fp32).
 * Guaranteed precision (arm_fir_f32 has none that depends on the input; this does, and here is the bound).  Let Mw be the largest
 * magnitude among the samples a tile's outputs can meet (its 1024 samples and the numTaps - 1 before them).  Every sample x of that
 * window enters the products with an error of at most max(2^-21 |x|, 2^-39 Mw), every tap h with at most max(2^-21 |h|, 2^-38 hmax),
 * the sums are fp32.  So a sample keeps its full 22 bits down to 2^-18 of Mw; below that it keeps 22 - (R - 18) bits at 2^-R of Mw
 * (19 bits one part in a million below the tile's peak).  An isolated spike therefore costs the quiet outputs of ITS OWN tile window a
 * relative 2^-19 ... 2^-15 for spikes 10^6 ... 10^7 times the quiet level, never the stretches before or after that window
 * (tests/test_gpu_stages.py: test_fir_f32_isolated_spike, test_fir_f32_matrix_core_input_ranges). */
int msdr_fir_f32_set_input_range(msdr_fir_f32 *S, float max_abs);
int msdr_fir_f32_destroy(msdr_fir_f32 *S);

/* arm_biquad_cascade_df1_init_f32 / arm_biquad_cascade_df1_f32 (prototypes arm_math.h:1333-1351).
 * pCoeffs: host, 5*numStages {b0,b1,b2,a1,a2}, feedback terms ADDED (CMSIS convention). */
typedef struct msdr_biquad_df1_f32 msdr_biquad_df1_f32;
int msdr_biquad_df1_f32_create(msdr_ctx *ctx, uint8_t numStages, const float32_t *pCoeffs, uint32_t channels, msdr_biquad_df1_f32 **out);
int msdr_biquad_df1_f32_process(msdr_biquad_df1_f32 *S, const float32_t *d_src, float32_t *d_dst, uint32_t blockSize);
int msdr_biquad_df1_f32_reset(msdr_biquad_df1_f32 *S);
/* New coefficients (all 5 * numStages of them) for a running cascade with CMSIS semantics: the filter carries on from the state
 * arm_biquad_cascade_df1_f32 would hold in pState -- x[n-1], x[n-2], y[n-1], y[n-2] of every stage (arm_math.h:1233) -- as it does
 * when a caller rewrites pCoeffs between two calls.  The block-parallel kernels keep their state in another basis (numerator history
 * + all-pole section states), which depends on the coefficients: the library converts old basis -> CMSIS state -> new basis on the
 * host (csrc/msdr_cascade_state.h), per channel, and switches between the block-parallel and the CMSIS-order kernel if the new
 * cascade's conditioning asks for it (msdr_biquad_df1_f32_cascade_info).  Synchronises the stream.  ARGUMENT_ERROR if the old
 * cascade's state cannot be expressed as a CMSIS state (a later stage's numerator shares a root with an earlier stage's denominator). */
int msdr_biquad_df1_f32_set_coeffs(msdr_biquad_df1_f32 *S, const float32_t *pCoeffs);
/* The state as CMSIS keeps it: pState[4 * numStages] of one channel (synchronises the stream). */
int msdr_biquad_df1_f32_get_cmsis_state(msdr_biquad_df1_f32 *S, uint32_t channel, float32_t *pState);
int msdr_biquad_df1_f32_destroy(msdr_biquad_df1_f32 *S);
/* Host only (no device needed): how the library will evaluate this cascade.  *kappa = ||c||_1 ||g||_1 / ||h||_1 (conditioning of
 * the parallel "numerators first" form), *fp32_noise = distance of the sequential fp32 evaluation (arm_biquad_cascade_df1_f32 as
 * written) from a double evaluation on a fixed test signal, *cmsis_order = 1 if instances created with these coefficients run the
 * cascade section by section in CMSIS order instead of the block-parallel solver (the study: profiles/r02/DESIGN_r02_log.md 4.4c).  Any output pointer may be NULL. */
int msdr_biquad_df1_f32_cascade_info(uint8_t numStages, const float32_t *pCoeffs, double *kappa, double *fp32_noise, int *cmsis_order);
/* Host only: the two state conventions of one cascade (see msdr_biquad_df1_f32_set_coeffs).  lib_state[16]: [0..7] = the cascade's
 * last inputs d[n-1-k] (2 * numStages used), [8 + 2 s], [9 + 2 s] = w[n-1], w[n-2] of all-pole section s.  pState: CMSIS, 4 per stage.
 * from_cmsis keeps lib_state[0..7] as given on entry (the input history; entries 0, 1 must equal pState[0], pState[1]) and fills
 * [8..15] so that the block-parallel form continues exactly as arm_biquad_cascade_df1_f32 would from pState. */
int msdr_biquad_df1_f32_state_to_cmsis(uint8_t numStages, const float32_t *pCoeffs, const float32_t lib_state[16], float32_t *pState);
int msdr_biquad_df1_f32_state_from_cmsis(uint8_t numStages, const float32_t *pCoeffs, const float32_t *pState, float32_t lib_state[16]);

/* AudioFilterBiquad (src/Audio/filter_biquad.cpp:33-100, filter_biquad.h:33-155): up to 4 stages,
 * Q2.30 coefficients, int16 data, 14-bit error feedback.  A new node passes nothing (all-zero
 * definition, filter_biquad.h:36-39).  set_coefficients keeps the filter history like the
 * reference (filter_biquad.cpp:95-97) and silently ignores stage >= 4 (:86).
 * update(): in place on d_data [channels][blockSize]; blockSize must be even (the reference
 * processes sample pairs, :54-74). */
typedef struct msdr_biquad_q15 msdr_biquad_q15;
int msdr_biquad_q15_create(msdr_ctx *ctx, uint32_t channels, msdr_biquad_q15 **out);
int msdr_biquad_q15_set_coefficients(msdr_biquad_q15 *S, uint32_t stage, const int32_t coef[5]);
int msdr_biquad_q15_update(msdr_biquad_q15 *S, q15_t *d_data, uint32_t blockSize);
int msdr_biquad_q15_get_definition(msdr_biquad_q15 *S, uint32_t channel, int32_t definition[32]); /* filter_biquad.h:152 */
int msdr_biquad_q15_destroy(msdr_biquad_q15 *S);

/* ======================================================================================
 * the front end in front of queue_adc (Minimal-SDR.ino:66-69, :76):
 *   adc1 (DC-block high-pass, src/Audio/input_adc.cpp:198-212) -> amp_adc (AudioAmplifier,
 *   src/Audio/mixer.cpp:34-47, :134-159) and AGC() (Minimal-SDR.ino:446-515), which
 *   demodulation() runs on every block it dequeues (:534) and which re-tunes amp_adc.
 * One instance keeps, per channel, what the reference keeps in statics: hpf_x1/hpf_y1
 * (input_adc.h:44-45), amp_adc's multiplier, AGC_val / AGC_on (.ino:100-104) and AGC()'s
 * 25-entry peak buffer + index.  Integer-exact (bit-for-bit with the oracle).
 * ====================================================================================== */
typedef struct msdr_frontend msdr_frontend;
#define MSDR_FE_DCBLOCK 1u           /* input is raw unsigned 16-bit conversions; run the DC-block filter */
#define MSDR_FE_AMP 2u               /* apply amp_adc's gain */
#define MSDR_FE_AGC 4u               /* run AGC() on every 128-sample block (needs MSDR_FE_AMP to have an effect) */
#define MSDR_FE_ALL 7u
#define MSDR_FE_STATE_WORDS 32u      /* [0] hpf_y1 [1] hpf_x1 [2] multiplier [3] agc_idx [4] AGC_val (float bits) [5] AGC_on [6..18] agc_buffer */
/* AudioInputAnalog() + AudioAmplifier() + the sketch's globals: hpf 0/0, AGC_val = AGC_start = 0.25 (.ino:94,:104,:385), AGC_on = 1 */
int msdr_frontend_create(msdr_ctx *ctx, uint32_t channels, msdr_frontend **out);
/* AudioInputAnalog::init, input_adc.cpp:60-63: hpf_x1 = first conversion << 14, hpf_y1 = 0.
 * first_conversion: host array of `count` values, count = 1 (all channels alike) or = channels. */
int msdr_frontend_prime(msdr_frontend *fe, const uint16_t *first_conversion, uint32_t count);
int msdr_frontend_set_agc(msdr_frontend *fe, int on);                     /* AGC_on, .ino:100 */
int msdr_frontend_gain(msdr_frontend *fe, float n);                       /* AGC_val = n; amp_adc.gain(n), mixer.h:75-79 */
/* d_adc: [channels][blockSize] uint16 (MSDR_FE_DCBLOCK) or int16; d_out: [channels][blockSize] int16 (may alias d_adc).
 * blockSize must be a multiple of 128 (AUDIO_BLOCK_SAMPLES: the AGC's update cadence), else MSDR_STATUS_LENGTH_ERROR.
 * A zero multiplier makes AudioAmplifier transmit nothing (mixer.cpp:139-142): such blocks come out as zeros here
 * and do not reach AGC(). */
int msdr_frontend_update(msdr_frontend *fe, const void *d_adc, q15_t *d_out, uint32_t blockSize, uint32_t stages);
int msdr_frontend_get_state(msdr_frontend *fe, uint32_t channel, int32_t state[MSDR_FE_STATE_WORDS]);
int msdr_frontend_destroy(msdr_frontend *fe);
/* AudioAmplifier as a stateless stage: multiplier as gain() computes it (mixer.h:75-79); in place.
 * *transmitted (may be NULL) = 0 when the node would transmit nothing (multiplier 0), else 1. */
int32_t msdr_amp_multiplier(float n);
int msdr_amp_q15(msdr_ctx *ctx, int32_t multiplier, q15_t *d_data, uint32_t channels, uint32_t blockSize, int *transmitted);

/* ======================================================================================
 * synchronous AM -- the PLL branch of the demod switch on Teensy 3.5/3.6
 * (Minimal-SDR.ino:631-688).  State per channel = the statics fil_out, omega2, phzerror
 * (:643-645).  d_mode: device int32 [channels] or NULL; with a mode array only SYNCAM
 * channels run the PLL, the others pass d_I through.  d_out may alias d_I.
 * sinf/cosf/atan2f are evaluated correctly rounded (see oracle/msdr_oracle.h, row f2).
 * In the fused Q15 chain, MSDR_CHAIN_SYNCAM_PLL selects this branch for SYNCAM channels
 * (default: SYNCAM demodulates like AM, the Teensy 3.2 build, .ino:618-627).
 * ====================================================================================== */
typedef struct msdr_syncam msdr_syncam;
int msdr_syncam_create(msdr_ctx *ctx, uint32_t channels, msdr_syncam **out);
int msdr_syncam_q15(msdr_syncam *S, const int32_t *d_mode, const q15_t *d_I, const q15_t *d_Q, q15_t *d_out, uint32_t blockSize);
int msdr_syncam_reset(msdr_syncam *S);
int msdr_syncam_get_state(msdr_syncam *S, uint32_t channel, float state[3]);    /* fil_out, omega2, phzerror */
void msdr_syncam_constants(float c[4]);                                         /* omega_min, omega_max, g1, g2 (:639-642) */
int msdr_syncam_destroy(msdr_syncam *S);

/* ======================================================================================
 * LMS automatic notch / noise reduction (Minimal-SDR.ino:702-770), the step
 * between the demod switch and queue_dac.playBuffer().  ANR_on: 0 = off, 1 = notch filter
 * (output = error), 2 = noise reduction (output = y).  State per channel = the statics
 * ANR_lidx, ANR_ngamma, ANR_in_idx, ANR_w[64] and the part of ANR_d[512] that is ever read.
 * d_anr_on: device int32 [channels] or NULL (then anr_on_all).  In place.
 * In the fused Q15 chain: msdr_chain_set_anr().
 * ====================================================================================== */
typedef struct msdr_anr msdr_anr;
#define MSDR_ANR_STATE_FLOATS 196u   /* lidx, ngamma, in_idx (int bits), pad | w[64] | d[128] (slot = ANR_in_idx & 127) */
int msdr_anr_create(msdr_ctx *ctx, uint32_t channels, msdr_anr **out);
int msdr_anr_q15(msdr_anr *A, const int32_t *d_anr_on, int32_t anr_on_all, q15_t *d_data, uint32_t blockSize);
int msdr_anr_reset(msdr_anr *A);
int msdr_anr_get_state(msdr_anr *A, uint32_t channel, float state[MSDR_ANR_STATE_FLOATS]);
int msdr_anr_destroy(msdr_anr *A);

/* ======================================================================================
 * SURVEY.md 8(f4), second half: the spectrum display's FFT (UI.cpp:520-592).
 *   msdr_rfft128_q15       <-> arm_rfft_q15(&FFT, data, FFT_out) with FFT = arm_rfft_init_q15(&FFT, 128, 0, 1) (UI.cpp:523,
 *                              :551; arm_rfft_q15.c:74-112, Cortex-M4 branches), batched: transform f reads 128 int16 at
 *                              d_src + f * src_stride (src_stride a multiple of 8 samples, d_src 16-byte aligned) and writes
 *                              d_fft_out[f][256] (interleaved re, im of 128 bins, the upper half the conjugate mirror) and/or
 *                              d_columns[f][128] = the 127 column heights showSpectrum draws,
 *                              y_new[x] = min(abs(FFT_out[127 - x]) / 200, 16) (UI.cpp:557-572), entry 127 = 0.
 *                              Either output may be NULL.  Unlike the reference, d_src is NOT transformed in place.
 *   msdr_rfft_q15_init_check   the argument check of arm_rfft_init_q15 (arm_rfft_init_q15.c:2154-2225): SUCCESS for
 *                              fftLenReal in {32 .. 8192} powers of two, else ARGUMENT_ERROR; this library runs 128 forward
 *                              with bit reversal only (what initSpectrum asks for) and reports other valid sizes as
 *                              MSDR_STATUS_LENGTH_ERROR.
 *   msdr_spectrum_*        <-> initSpectrum() / showSpectrum(): Spectrum_on, and spectrumCounter's "every 25th call"
 *                              cadence (UI.cpp:122, :534-536; the first call after create draws).  *drawn = 1 when this
 *                              call computed a spectrum (outputs written), 0 when it returned early like the reference.
 *   msdr_rfft128_tables    host side, no device: the constant tables the transform uses, regenerated from their documented
 *                              formulas -- twiddleCoef_64_q15[96] | realCoefAQ15 pairs at stride 64 [128] | realCoefBQ15 ditto [128]
 *                              (arm_common_tables.c:12914, arm_rfft_init_q15.c:44-56, :1086-1098).
 */
void msdr_rfft128_tables(int16_t tables[352]);
int msdr_rfft_q15_init_check(uint32_t fftLenReal, uint32_t ifftFlagR, uint32_t bitReverseFlag);
int msdr_rfft128_q15(msdr_ctx *ctx, const q15_t *d_src, uint64_t src_stride, q15_t *d_fft_out, uint8_t *d_columns, uint32_t nfft);
typedef struct msdr_spectrum msdr_spectrum;
int msdr_spectrum_create(msdr_ctx *ctx, uint32_t channels, msdr_spectrum **out);
int msdr_spectrum_set_on(msdr_spectrum *S, int spectrum_on);
int msdr_spectrum_show(msdr_spectrum *S, const q15_t *d_data, uint64_t channel_stride, q15_t *d_fft_out, uint8_t *d_columns, int *drawn);
int msdr_spectrum_destroy(msdr_spectrum *S);

/* Stateless per-block stages. */
/* what AudioOutputAnalog::isr writes to the 12-bit DAC, output_dac.cpp:139-151: (sample + 32768) >> 4;
 * d_src == NULL = no block arrived: 2048 (mid-scale).  d_dest may alias d_src. */
int msdr_dac_format_q15(msdr_ctx *ctx, const q15_t *d_src, q15_t *d_dest, uint32_t channels, uint32_t blockSize);
/* Minimal-SDR.ino:546-558; block must start at a sample index = 0 (mod 4), as every 128-block does */
int msdr_mix_fs4_q15(msdr_ctx *ctx, const q15_t *d_x, q15_t *d_i, q15_t *d_q, uint32_t channels, uint32_t blockSize);
/* AudioEffectFreqConv::update, freq_conv.cpp:30-116, in place on d_i/d_q.  osc_i/osc_q: host tables of
 * osc_len entries (Osc_I_buffer_i / Osc_Q_buffer_i, freq_conv.h:33-34), applied at index (n mod osc_len).
 * `pass` keeps the reference's inverted meaning: pass == 0 forwards untouched (:49-56). */
int msdr_freqconv_q15(msdr_ctx *ctx, q15_t *d_i, q15_t *d_q, const q15_t *osc_i, const q15_t *osc_q, uint32_t osc_len,
                      int dir, int pass, uint32_t channels, uint32_t blockSize);
int msdr_freqconv_f32(msdr_ctx *ctx, float32_t *d_i, float32_t *d_q, const float32_t *osc_i, const float32_t *osc_q,
                      uint32_t osc_len, int dir, int pass, uint32_t channels, uint32_t blockSize);
/* demod switch, Minimal-SDR.ino:589-627.  d_mode: device int32 [channels] or NULL (then `mode` for all). */
int msdr_demod_q15(msdr_ctx *ctx, int mode, const int32_t *d_mode, int sqrt_kind, const q15_t *d_i, const q15_t *d_q,
                   q15_t *d_out, uint32_t channels, uint32_t blockSize);
int msdr_demod_f32(msdr_ctx *ctx, int mode, const int32_t *d_mode, const float32_t *d_i, const float32_t *d_q,
                   float32_t *d_out, uint32_t channels, uint32_t blockSize);

/* ======================================================================================
 * Fused chain = demodulation() (Minimal-SDR.ino:518-775: mix :546-558, FIR pair :574-575,
 * demod :589-691) + the biquad nodes wired behind queue_dac (.ino:77-81), ONE kernel pass:
 * one HBM read of the IF block batch, one HBM write of the audio block batch.
 * ====================================================================================== */
typedef struct {
    uint32_t struct_size;            /* = sizeof(msdr_chain_config) */
    int32_t  arith;                  /* MSDR_ARITH_Q15 | MSDR_ARITH_F32 */
    uint32_t channels;
    int32_t  mixer;                  /* MSDR_MIXER_FS4 | MSDR_MIXER_NCO */
    /* tap sets: the reference binds one coefficient pair per mode (init_FIR, .ino:901-930) */
    uint32_t num_taps;               /* taps per filter; q15 needs it even */
    uint32_t num_tapsets;            /* 1..MSDR_MAX_TAPSETS */
    const void *coeffs_i[MSDR_MAX_TAPSETS];   /* host; q15_t[num_taps] or float32_t[num_taps] by arith */
    const void *coeffs_q[MSDR_MAX_TAPSETS];
    /* per-channel selection (host arrays of `channels` entries, or NULL for the defaults) */
    int32_t  default_mode;           /* MSDR_MODE_AM / LSB / USB / CW */
    const int32_t *mode;             /* demod mode per channel */
    const int32_t *tapset;           /* tap-set index per channel (default 0) */
    int32_t  sqrt_kind;              /* MSDR_SQRT_F32 | MSDR_SQRT_Q31 (q15 arithmetic only) */
    /* NCO tables for MSDR_MIXER_NCO: q15_t or float32_t by arith; index (n mod osc_len), n counted from reset */
    uint32_t osc_len;
    const void *osc_i;               /* "sin", Osc_I_buffer_i */
    const void *osc_q;               /* "cos", Osc_Q_buffer_i */
    /* IIR stage */
    float    in_scale;               /* F32: int16 -> float scale; 0 means 1/32768 (arm_q15_to_float) */
    uint32_t num_biquad_stages;      /* F32: 0..4 stages of arm_biquad_cascade_df1_f32 */
    const float32_t *biquad_coeffs;  /* F32: host, 5*num_biquad_stages.  ACCURACY CONTRACT of the fp32 chain (tests/test_gpu_f32_contract.py):
                                      *   every output row is within 1e-5 relative RMS of arm_fir_f32 + arm_biquad_cascade_df1_f32 evaluated in CMSIS order in
                                      *   fp32 -- or, where that sequential fp32 evaluation is itself further than that from the exact (float64) result, within
                                      *   TWICE its distance from the exact result (+ fp32_noise + 1e-6, fp32_noise as below).  The second clause matters for cascades whose sections ring
                                      *   (narrow notches, resonant low / high-passes, three or more sections): their fp32 rounding noise, in any order of
                                      *   evaluation, is msdr_biquad_df1_f32_cascade_info()'s *fp32_noise (4e-7 for the reference's LP + notch; 1e-5 and more for
                                      *   random Q 8 sections below 1 kHz), and the block-parallel solver adds up to *kappa times the per-sample rounding; the
                                      *   library switches to the CMSIS order itself (one lane per channel behind the main kernel, *cmsis_order = 1) when
                                      *   kappa > 30 (20 from three sections on), kappa x fp32_noise > 2e-5 or fp32_noise > 2e-6, when a host-side emulation of
                                      *   the block-parallel evaluation is more than 1.5 x fp32_noise + 2e-7 from float64 on the host's test signal, and for three or four sections behind the
                                      *   general kernel.  Where the cascade removes most of its input (stacked high-passes), 1e-5 and 1e-6 are referred to the
                                      *   level of the cascade's input.  With the reference's filters and all BASELINE configurations: 5-6e-7.
                                      *   Round 5, frozen: the second clause is held on EVERY output row, excused or not -- the library is never further from the
                                      *   exact result than twice the CMSIS order's own distance + fp32_noise + 1e-6 of the cascade's input level (measured: at most
                                      *   0.47 of that bound).  That is what gives the gate teeth below 1e-5: builds with the taps cut to 16 bits (4e-6), the lo
                                      *   tap pieces dropped, or time segments without warm-up fail it (tests/test_gpu_f32_teeth.py).  The error model behind each
                                      *   term is DESIGN.md 5; a further term needs an argument from the reference's arithmetic, not a fuzz finding. */
    uint32_t num_biquad_nodes;       /* Q15: 0..2 AudioFilterBiquad nodes in series (biquad1_dac, biquad2_dac) */
    uint32_t node_stages[2];         /* Q15: stages used in each node (1..4) */
    const int32_t *node_coefs[2];    /* Q15: host, 5*node_stages[k] ints as given to setCoefficients */
    /* time segmentation of one call (F32 only; see DESIGN.md "IIR along time"):
     * 0 = automatic; 1 = never split a channel's block in time (IIR state carried exactly);
     * k > 1 = split every channel's block into k segments, each re-converging the IIR over
     * `biquad_warmup` samples (0 = derive from the pole radii for < 1e-9 residual). */
    uint32_t time_segments;
    uint32_t biquad_warmup;
    uint32_t flags;                  /* MSDR_CHAIN_* */
} msdr_chain_config;
#define MSDR_CHAIN_NO_TAP_FOLDING 1u /* F32: keep mixer and FIR pair as separate arithmetic steps (as written) */
#define MSDR_CHAIN_NO_FFT 4u         /* deprecated, accepted and ignored (round 2's FFT kernel is gone) */
#define MSDR_CHAIN_MFMA_WG 16u       /* deprecated, accepted and ignored (round 2's workgroup-tile matrix-core kernel is gone) */
#define MSDR_CHAIN_OUT_I16 128u      /* F32: d_audio is int16 [channels][n_samples] -- the play queue's sample type (src/Audio/play_queue.h:41) -- written as
                                        arm_float_to_q15 converts (prototype arm_math.h:6592; CMSIS-DSP 1.5.x: (q15_t) __SSAT((q31_t)(x * 32768.0f), 16)): 4 B
                                        per sample through HBM instead of 6, and half the bytes in the audio gather.  The matrix-core kernels convert in
                                        their store phase; chains that run a pass behind the main kernel (CMSIS-order cascade, PLL / LMS channels) or
                                        one of the vector-ALU kernels go through an fp32 scratch block batch owned by the chain: channels x n_samples x 4 bytes,
                                        allocated at the first such call and grown (with a stream synchronisation) when a longer call comes, plus one
                                        conversion pass over it -- 10 B per sample through HBM, not 4; call such chains in blocks, not in 2^18-sample
                                        calls (4 GB on c3's shape).  Ignored by Q15 chains. */
/* any other bit in msdr_chain_config.flags is refused with MSDR_STATUS_ARGUMENT_ERROR */
#define MSDR_CHAIN_NO_MFMA 8u        /* never run the FIR on the matrix cores (F32: split-fp16 MFMA kernel; Q15: byte-split i8 MFMA kernel) */
#define MSDR_CHAIN_FOLD_ANY_PERIOD 64u /* F32, NCO: fold the mixer into the taps (matrix-core kernel) also when the oscillator table's only period is its
                                         own length (8, 16 or 32 samples); a table that repeats within its length with period 1 .. 32 is folded by default */
#define MSDR_CHAIN_SYNCAM_PLL 32u    /* SYNCAM channels run the PLL demodulator (.ino:631-688) instead of the AM branch.  Q15: as the reference, on the int16
                                       FIR outputs.  F32 (an extension): the same loop on the fp32 FIR outputs, audio = corr[0] untruncated */

typedef struct msdr_chain msdr_chain;
int msdr_chain_create(msdr_ctx *ctx, const msdr_chain_config *cfg, msdr_chain **out);
/* d_if: int16 [channels][n_samples]; d_audio: int16 (Q15, or F32 with MSDR_CHAIN_OUT_I16) or float (F32) [channels][n_samples].
 * State (FIR history, IIR state, NCO phase) is carried from call to call, so calling with
 * n_samples = 128 reproduces the reference's block cadence and one call with a long block is the
 * same stream.  Q15 arithmetic needs n_samples even when biquad nodes are present. */
int msdr_chain_process(msdr_chain *chain, const int16_t *d_if, void *d_audio, uint64_t n_samples);
/* init_FIR() (Minimal-SDR.ino:901-930: memset + arm_fir_init_q15 of both instances): zeroes the FIR state and nothing else --
 * biquad / PLL / LMS state and the mixer's table position persist across a retune, as in the reference. */
int msdr_chain_init_fir(msdr_chain *chain);
/* stream restart (no counterpart in the reference): FIR state, fp32 cascade state, PLL and LMS state cleared, mixer phase 0.
 * The Teensy biquad NODES of a Q15 chain keep their history (the reference never clears it, filter_biquad.cpp:95-97). */
int msdr_chain_reset(msdr_chain *chain);
int msdr_chain_set_mode(msdr_chain *chain, uint32_t channel, int32_t mode, int32_t tapset);
/* ---- live updates: what the reference changes while the stream runs, every filter state kept ----------------------------------
 * All four synchronise the context's stream, rebuild the device tables of the chain on the host and take effect with the next
 * msdr_chain_process(); FIR history, cascade / node state, PLL and LMS state and the mixer's table position are untouched.
 *
 * msdr_chain_set_taps: new coefficients for one tap set (q15_t[num_taps] or float32_t[num_taps] by arith, CMSIS order) -- the
 *   bandwidth menu rewriting FIR_AM_coeffs in place under the running arm_fir_fast_q15 (UI.cpp:337-345, Minimal-SDR.ino:221-223:
 *   no init_FIR(), the instance holds a pointer, arm_fir_init_q15.c:100-109).  Every channel on that tap set hears the new filter
 *   from its next sample on, over the old filter's history.  F32 chains whose SSB tables carry the cascade's numerator hand the
 *   true numerator history over as first-sample corrections, as msdr_chain_set_mode does.
 * msdr_chain_set_node_coefficients (Q15): AudioFilterBiquad::setCoefficients(stage, coef) on biquad node `node` (0 = biquad1_dac,
 *   1 = biquad2_dac) of a running chain -- tune() re-programming the notch on every retune, Minimal-SDR.ino:356 ->
 *   filter_biquad.cpp:84-100; history kept (:95-97), stage >= 4 silently ignored (:86).
 * msdr_chain_set_biquad_coeffs (F32): all 5 * num_biquad_stages coefficients of the arm_biquad_cascade_df1_f32 stage, CMSIS
 *   semantics (the filter carries on from the pState arm_biquad_cascade_df1_f32 would hold; see msdr_biquad_df1_f32_set_coeffs).
 *   The number of stages is fixed at creation, as numStages is in CMSIS.
 * msdr_chain_set_osc: new contents for the NCO tables (same osc_len) -- AudioEffectFreqConv reads the global Osc_I_buffer_i /
 *   Osc_Q_buffer_i on every update() (freq_conv.h:33-34, freq_conv.cpp:70-103), so a sketch that rewrites them retunes the mixer
 *   without touching anything else; the position in the table carries on.  The samples already in the FIR history were mixed with the
 *   table of their own time, as in the reference (the mixer runs in front of the FIR's state buffer): the library keeps up to 16 earlier
 *   tables for as long as the history holds samples of theirs (a 17th change inside ONE history length drops the oldest). */
int msdr_chain_set_taps(msdr_chain *chain, uint32_t tapset, const void *coeffs_i, const void *coeffs_q);
int msdr_chain_set_node_coefficients(msdr_chain *chain, uint32_t node, uint32_t stage, const int32_t coef[5]);
int msdr_chain_set_biquad_coeffs(msdr_chain *chain, const float32_t *biquad_coeffs);
int msdr_chain_set_osc(msdr_chain *chain, const void *osc_i, const void *osc_q);
/* ANR_on per channel (host array of `channels` values, or NULL: anr_on_all for every channel); the LMS filter then runs between
 * the demodulator and the biquad nodes / cascade (Minimal-SDR.ino:702-770).  Its state is created on first use and cleared by
 * msdr_chain_reset() (not by msdr_chain_init_fir()).  Q15 chains: as the reference, on the int16 audio.  F32 chains (an
 * extension): the same statements on the fp32 audio, in the reference's sample units (x 32768 in, / 32768 out), nothing truncated;
 * such channels (and SYNCAM channels under MSDR_CHAIN_SYNCAM_PLL) are demodulated by an auxiliary pass behind the main kernel --
 * one lane per channel, serial in time like the reference -- and their cascade restarts when a retune changes the set. */
int msdr_chain_set_anr(msdr_chain *chain, const int32_t *anr_on, int32_t anr_on_all);
int msdr_chain_destroy(msdr_chain *chain);
/* introspection for benchmarks/tests: name of the main kernel variant and launch geometry of the last call */
typedef struct {
    char     kernel[64];
    uint32_t grid, block, lds_bytes;
    uint32_t time_segments, warmup, tile;
    uint32_t taps_padded;
    uint32_t mfma_ksteps;            /* matrix-core kernel: 16-sample k-steps (3 MFMAs each) per 1024-output wave tile, else 0 */
} msdr_chain_info;
int msdr_chain_get_info(msdr_chain *chain, msdr_chain_info *info);
/* Measurement aid (bench.py): when enabled every msdr_chain_process() brackets its MAIN kernel with
 * HIP events on the context's stream; get_kernel_time synchronises and returns the accumulated
 * device time in ms and the number of launches timed (reset != 0 clears the accumulators). */
int msdr_chain_enable_timing(msdr_chain *chain, int on);
int msdr_chain_get_kernel_time(msdr_chain *chain, double *total_ms, uint64_t *launches, int reset);


/* ---- block cadence: `ticks` consecutive calls as ONE HIP graph ------------------------------------------------------------------
 * The reference runs one AUDIO_BLOCK_SAMPLES = 128 block per call (Minimal-SDR.ino:518-530, FIR calls :574-575); at that size a call
 * is a few microseconds of GPU work and the host's launch cost is of the same order.  msdr_chain_graph_create records the launches
 * of `ticks` consecutive msdr_chain_process(chain, d_if[k], d_audio[k], n_samples) calls, k = 0 .. ticks - 1, into a HIP graph on the
 * context's stream; msdr_chain_graph_launch enqueues all of them at once (the caller refills d_if[] and collects d_audio[] between
 * launches, ordered on the same stream).  The chain's state advances exactly as under the direct calls: direct calls, replays and
 * live updates may follow one another -- a replay is REFUSED (MSDR_STATUS_ARGUMENT_ERROR, nothing enqueued) once a live update, a reset
 * or an odd number of direct calls has moved what the recorded launches point at; make the graph again then.
 *   ticks: even, 2 .. 1024 (history and cascade state alternate between two buffers from call to call).
 *   n_samples: a block-cadence length (32, 64, 128, 256 or 512; 16-byte aligned buffers) on a chain whose call is ONE fixed set of
 *   launches: matrix-core tables in use, no PLL / LMS channels behind the kernel, no cascade in CMSIS order behind it, no pending
 *   oscillator change, oscillator period dividing n_samples; anything else returns MSDR_STATUS_ARGUMENT_ERROR with the reason in
 *   msdr_last_error() and leaves the chain untouched. */
typedef struct msdr_chain_graph msdr_chain_graph;
int msdr_chain_graph_create(msdr_chain *chain, uint32_t ticks, const int16_t *const *d_if, void *const *d_audio, uint64_t n_samples, msdr_chain_graph **out);
int msdr_chain_graph_launch(msdr_chain_graph *graph);
int msdr_chain_graph_destroy(msdr_chain_graph *graph);

#ifdef __cplusplus
}
#endif
#endif /* MSDR_H */
