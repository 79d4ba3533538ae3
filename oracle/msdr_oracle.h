/*
 * oracle/msdr_oracle.h -- CPU restatement of Minimal-SDR's per-block demodulation chain.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library; the product path
 * (minimal-sdr_amd/, include/msdr.h) never links, loads or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Pinning status (see DESIGN.md "Oracle"):
 *   PINNED   against oracle/_ref/libmsdr_ref.so (the reference's own sources compiled here by
 *            oracle/build_ref.sh): orc_fir_init_q15, orc_fir_fast_q15, orc_copy_q15,
 *            orc_sqrt_q31, orc_calc_fir_coeffs, orc_izero, orc_m_sinc; the q15 FFT stages (row f4); arm_sqrt_f32 as the
 *            AM branch calls it (a static inline of arm_math.h, orc_prim_sqrt_f32);
 *            the arithmetic PRIMITIVES of orc_biquad_teensy_*, orc_amp_* and the DC blocker -- mulw16 on either
 *            half-word, ssat16(v >> s), the history-word packing -- against the plain-C bodies dspinst.h itself
 *            carries for the Cortex-M0+ (src/Audio/utility/dspinst.h:34-51, :70-92, :151-184; orc_prim_* below).
 *   UNPINNED ("parity unpinned": the reference code cannot be built here without stand-ins for
 *            the un-vendored Teensyduino core / ARM inline asm, or has no source at all):
 *            orc_mix_fs4_q15, orc_freqconv_q15 (+ orc_mult/add/sub_q15), orc_demod_q15,
 *            the loop of orc_biquad_teensy_* (its multiply-accumulate wrappers smlawb / smlawt are assembly only,
 *            dspinst.h:235-249, and filter_biquad.cpp needs the Teensy core), and every *_f32 function (arm_fir_f32 and
 *            arm_biquad_cascade_df1_f32 are prototypes only in the reference:
 *            src/CMSIS_5/arm_math.h:1182-1202, :1333-1351; module CMSIS-DSP V1.5.x).
 */
#ifndef MSDR_ORACLE_H
#define MSDR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int16_t q15_t;
typedef int32_t q31_t;

/* stations.h:4  enum { SYNCAM, AM, LSB, USB, CW } */
enum { ORC_SYNCAM = 0, ORC_AM = 1, ORC_LSB = 2, ORC_USB = 3, ORC_CW = 4 };
/* which square root the AM/CW branch uses: Minimal-SDR.ino:606-616 (T3.6, sqrtf) or :618-627 (T3.2, arm_sqrt_q31) */
enum { ORC_SQRT_F32 = 0, ORC_SQRT_Q31 = 1 };
/* arm_math.h:404-413 */
enum { ORC_SUCCESS = 0, ORC_ARGUMENT_ERROR = -1 };

#define ORC_BLOCK 128 /* AUDIO_BLOCK_SAMPLES, Teensy core default; used e.g. Minimal-SDR.ino:525 */

/* ---- A1: Fs/4 I/Q mixer, Minimal-SDR.ino:546-558 ------------------------------------- */
void orc_mix_fs4_q15(const int16_t *x, int16_t *I, int16_t *Q, uint32_t n);

/* ---- A2: freq_conv.cpp:30-116 (arithmetic = CMSIS-DSP 1.5.x arm_mult/add/sub_q15) ------ */
void orc_mult_q15(const q15_t *a, const q15_t *b, q15_t *dst, uint32_t n);
void orc_add_q15(const q15_t *a, const q15_t *b, q15_t *dst, uint32_t n);
void orc_sub_q15(const q15_t *a, const q15_t *b, q15_t *dst, uint32_t n);
/* in place on I,Q like the node; osc_i = Osc_I_buffer_i ("sin"), osc_q = Osc_Q_buffer_i ("cos") */
void orc_freqconv_q15(q15_t *I, q15_t *Q, const q15_t *osc_i, const q15_t *osc_q,
                      int dir, int pass, uint32_t n);
void orc_freqconv_f32(float *I, float *Q, const float *osc_i, const float *osc_q,
                      int dir, int pass, uint32_t n);

/* ---- A3/A4: arm_fir_init_q15.c:78-138, arm_fir_fast_q15.c:60-329 ----------------------- */
typedef struct {
    uint16_t numTaps;
    q15_t *pState;        /* numTaps + blockSize elements (arm_fir_init_q15.c:106) */
    const q15_t *pCoeffs; /* numTaps, "time reversed" storage (arm_fir_init_q15.c:51-54) */
} orc_fir_instance_q15;
int orc_fir_init_q15(orc_fir_instance_q15 *S, uint16_t numTaps, const q15_t *pCoeffs,
                     q15_t *pState, uint32_t blockSize);
void orc_fir_fast_q15(const orc_fir_instance_q15 *S, const q15_t *pSrc, q15_t *pDst,
                      uint32_t blockSize);
void orc_copy_q15(const q15_t *src, q15_t *dst, uint32_t n); /* arm_copy_q15.c:48-98 */

/* ---- A5: demod switch, Minimal-SDR.ino:589-691 ------------------------------------------ */
int orc_sqrt_q31(q31_t in, q31_t *out);                       /* arm_sqrt_q31.c:50-138 */
void orc_demod_q15(int mode, int sqrt_kind, const int16_t *I, const int16_t *Q,
                   int16_t *out, uint32_t n);
void orc_demod_f32(int mode, const float *I, const float *Q, float *out, uint32_t n);

/* ---- A6: arm_fir_f32 (CMSIS-DSP 1.5.x documented semantics; prototype arm_math.h:1182) -- */
typedef struct {
    uint16_t numTaps;
    float *pState;        /* numTaps + blockSize - 1 (arm_math.h:1050) */
    const float *pCoeffs;
} orc_fir_instance_f32;
void orc_fir_init_f32(orc_fir_instance_f32 *S, uint16_t numTaps, const float *pCoeffs,
                      float *pState, uint32_t blockSize);
void orc_fir_f32(const orc_fir_instance_f32 *S, const float *pSrc, float *pDst,
                 uint32_t blockSize);

/* ---- A7: Teensy AudioFilterBiquad, src/Audio/filter_biquad.cpp:33-100, filter_biquad.h --- */
typedef struct orc_biquad_teensy_s { int32_t definition[32]; } orc_biquad_teensy; /* filter_biquad.h:152 */
void orc_biquad_teensy_init(orc_biquad_teensy *b);                       /* h:36-39 */
void orc_biquad_teensy_set_coefficients(orc_biquad_teensy *b, uint32_t stage,
                                        const int32_t coef[5]);          /* cpp:84-100 */
void orc_biquad_teensy_update(orc_biquad_teensy *b, int16_t *data, uint32_t n); /* cpp:33-82 */
/* cookbook designers, filter_biquad.h:56-149; sample_rate = AUDIO_SAMPLE_RATE_EXACT there */
enum { ORC_BQ_LOWPASS = 0, ORC_BQ_HIGHPASS, ORC_BQ_BANDPASS, ORC_BQ_NOTCH,
       ORC_BQ_LOWSHELF, ORC_BQ_HIGHSHELF };
void orc_biquad_design(int kind, float frequency, float q_or_gain, float slope,
                       double sample_rate, int32_t coef[5]);
#define ORC_AUDIO_SAMPLE_RATE_EXACT 44117.64706 /* Teensy core constant (not in the repo) */

/* ---- A8: arm_biquad_cascade_df1_f32 (CMSIS-DSP 1.5.x semantics; prototype arm_math.h:1333) */
typedef struct {
    uint32_t numStages;
    float *pState;        /* 4*numStages: x[n-1],x[n-2],y[n-1],y[n-2] (arm_math.h:1233) */
    const float *pCoeffs; /* 5*numStages: b0,b1,b2,a1,a2 (feedback terms ADDED) */
} orc_biquad_df1_f32;
void orc_biquad_df1_init_f32(orc_biquad_df1_f32 *S, uint8_t numStages, const float *pCoeffs,
                             float *pState);
void orc_biquad_df1_f32_run(const orc_biquad_df1_f32 *S, const float *pSrc, float *pDst,
                            uint32_t blockSize);

/* ---- A9: FIR designer, Minimal-SDR.ino:782-899 ------------------------------------------ */
float orc_izero(float x);            /* :883-899 */
float orc_m_sinc(int m, float fc);   /* :874-881 */
void orc_set_pi_double(int on);      /* which `PI` the sketch saw: see msdr_oracle.c A9 */
void orc_calc_fir_coeffs(int16_t *coeffs, int numCoeffs, float fc, float Astop, int type,
                         float dfc, float Fsamprate);

/* ---- whole-chain drivers (compositions of the above, as demodulation() composes them) ---- */
typedef struct {
    int mode;                 /* ORC_AM / ORC_LSB / ORC_USB / ORC_CW */
    int sqrt_kind;            /* ORC_SQRT_F32 | ORC_SQRT_Q31 */
    int mixer;                /* 0 = Fs/4 inline mixer (A1); 1 = freq_conv node (A2), dir=1, Q input = 0 */
    uint32_t num_taps;
    const q15_t *coeffs_i;    /* num_taps */
    const q15_t *coeffs_q;
    const q15_t *osc_i;       /* ORC_BLOCK entries, used when mixer==1 */
    const q15_t *osc_q;
    uint32_t n_biquad_nodes;  /* 0..2 AudioFilterBiquad nodes after the demodulator (.ino:77-81) */
    const struct orc_biquad_teensy_s *bq_init; /* batch drivers: initial node records (coefficients) */
} orc_chain_q15_cfg;
typedef struct {
    q15_t *state_i;           /* num_taps + ORC_BLOCK */
    q15_t *state_q;
    orc_biquad_teensy bq[2];
} orc_chain_q15_state;
/* Processes n_blocks consecutive blocks of ORC_BLOCK samples of ONE channel.
 * i_out/q_out (may be NULL) receive the post-FIR I/Q intermediates. */
void orc_chain_q15(const orc_chain_q15_cfg *cfg, orc_chain_q15_state *st, const int16_t *x,
                   int16_t *audio, int16_t *i_out, int16_t *q_out, uint32_t n_blocks);

typedef struct {
    int mode;                 /* ORC_AM / ORC_LSB / ORC_USB / ORC_CW */
    float in_scale;           /* int16 -> float scale (1/32768 = arm_q15_to_float convention) */
    uint32_t num_taps;
    const float *coeffs_i;
    const float *coeffs_q;
    uint32_t osc_len;         /* NCO table period (freq_conv: one block) */
    const float *osc_i;       /* "sin" table */
    const float *osc_q;       /* "cos" table */
    uint32_t num_stages;      /* 0..4 */
    const float *bq_coeffs;   /* 5*num_stages */
} orc_chain_f32_cfg;
typedef struct {
    float *hist_i;            /* num_taps-1 history */
    float *hist_q;
    float bq_state[16];
    uint64_t n0;              /* absolute sample index of the next input sample (NCO phase) */
} orc_chain_f32_state;
/* One channel, n samples, sample-sequential, sequential tap order, no FMA contraction. */
void orc_chain_f32(const orc_chain_f32_cfg *cfg, orc_chain_f32_state *st, const int16_t *x,
                   float *audio, uint64_t n);

/* batched CPU-baseline drivers: channel c uses x + c*n, audio + c*n; fresh zero state per call.
 * Channels are spread over `threads` OpenMP threads (schedule(static)). Returns threads used. */
int orc_chain_f32_batch(const orc_chain_f32_cfg *cfg, const int32_t *mode_per_channel,
                        const int16_t *x, float *audio, uint32_t channels, uint64_t n,
                        int threads);
/* FIR stage over a block batch (fresh zero state per row, `block` samples per call); return the OpenMP team size used */
int orc_fir_f32_batch(const float *coeffs, uint16_t num_taps, const float *x, float *y, uint32_t channels,
                      uint64_t n, uint32_t block, int threads);
int orc_fir_q15_batch(const q15_t *coeffs, uint16_t num_taps, const q15_t *x, q15_t *y, uint32_t channels,
                      uint64_t n, uint32_t block, int threads);
int orc_chain_q15_batch(const orc_chain_q15_cfg *cfg, const int32_t *mode_per_channel,
                        const int16_t *x, int16_t *audio, uint32_t channels,
                        uint32_t n_blocks, int threads);

/* ======================================================================================
 * Row f1: the front end in front of queue_adc (Minimal-SDR.ino:66-69, 76):
 *   adc1 (DC-block high-pass, src/Audio/input_adc.cpp:198-212) -> amp_adc (AudioAmplifier,
 *   src/Audio/mixer.cpp:34-47, :134-159, gain() mixer.h:75-79) -> queue_adc, and AGC()
 *   (Minimal-SDR.ino:446-515), which demodulation() runs on every block it dequeues (:534).
 * UNPINNED: input_adc.cpp / mixer.cpp need the un-vendored Teensyduino core and ARM inline asm
 * (dspinst.h smulwb/ssat/smull), AGC() needs __SSUB16/__SEL (ARM-only in cmsis_gcc.h:1684,:2022).
 * ====================================================================================== */
typedef struct { int32_t hpf_y1, hpf_x1; } orc_dcblock;                 /* input_adc.h:44-45 */
void orc_dcblock_init(orc_dcblock *s, uint16_t first_conversion);          /* input_adc.cpp:60-63 */
/* adc: raw unsigned 16-bit conversions; out: the block AudioInputAnalog transmits */
void orc_dcblock_update(orc_dcblock *s, const uint16_t *adc, int16_t *out, uint32_t n);
/* test hooks: the oracle's smulwb / smulwt / ssat-asr primitives (pinned against dspinst.h's own C bodies) */
int32_t orc_prim_mulw16b(int32_t a, uint32_t b);
int32_t orc_prim_mulw16t(int32_t a, uint32_t b);
int32_t orc_prim_ssat16_rshift(int32_t v, int rshift);
uint32_t orc_prim_pack_hist(int32_t newer, int32_t older);
float orc_prim_sqrt_f32(float in);                                         /* the AM branch's arm_sqrt_f32 (arm_math.h:5733-5758) */
int32_t orc_amp_multiplier(float gain);                                    /* AudioAmplifier::gain, mixer.h:75-79 */
/* AudioAmplifier::update, mixer.cpp:134-159: returns 0 when nothing is transmitted (multiplier 0), else 1; in place */
int orc_amp_update(int32_t multiplier, int16_t *data, uint32_t n);
#define ORC_AGCBUF_SIZE 25                                                 /* Minimal-SDR.ino:445 */
typedef struct {
    int16_t agc_buffer[ORC_AGCBUF_SIZE];   /* static in AGC(), zero initialised */
    int32_t agc_idx;                       /* static, starts at AGCBUF_SIZE */
    float AGC_val;                         /* Minimal-SDR.ino:104, AGC_start = 0.25f (:94) */
    int32_t AGC_on;                        /* :100 */
    int32_t multiplier;                    /* amp_adc's, set by amp_adc.gain(AGC_val) (:385, :492 ...) */
} orc_agc;
void orc_agc_init(orc_agc *a);
void orc_agc_block(orc_agc *a, const int16_t *block);   /* AGC(p_adc) on one 128-sample block */
typedef struct { orc_dcblock dc; orc_agc agc; } orc_frontend;
void orc_frontend_init(orc_frontend *f, uint16_t first_conversion);
/* n_blocks blocks of ORC_BLOCK raw conversions -> the blocks demodulation() reads.  The gain AGC() sets while looking at
 * block k applies from block k+1 on (the reference has the record queue in between; one block is the shortest delay). */
void orc_frontend_run(orc_frontend *f, const uint16_t *adc, int16_t *out, uint32_t n_blocks);

/* ======================================================================================
 * Row f2: synchronous AM, the PLL branch of the demod switch on Teensy 3.5/3.6
 * (Minimal-SDR.ino:631-688; "code adapted from the wdsp library").  Sample-recursive.
 * UNPINNED (demodulation() cannot be built here).  Reading of the library calls: sinf / cosf /
 * atan2f are taken as CORRECTLY ROUNDED float functions (computed in double, rounded once);
 * newlib's on the Teensy are within 1 ulp of that.  `p_dac[i] = corr[0]` converts like the
 * Cortex-M4 does (vcvt to int32, store halfword): truncate toward zero, keep the low 16 bits.
 * ====================================================================================== */
typedef struct { float fil_out, omega2, phzerror; } orc_syncam;             /* the statics at :643-645 */
void orc_syncam_init(orc_syncam *s);
void orc_syncam_constants(float c[4]);      /* omega_min, omega_max, g1, g2 as the initialisers at :639-642 evaluate */
void orc_syncam_q15(orc_syncam *s, const int16_t *I, const int16_t *Q, int16_t *out, uint32_t n);

/* ======================================================================================
 * Row f3: LMS automatic notch / noise reduction between the demod switch and
 * queue_dac.playBuffer() (Minimal-SDR.ino:702-770, "variable leak LMS ... Warren Pratt's wdsp").
 * Sample-recursive, 64 taps, delay 16, 512-entry delay line.  UNPINNED (inside demodulation()).
 * Arithmetic as the source states it under C's conversions: float variables, double where an
 * unsuffixed literal (1.0, 1e-10, 0.0) enters the expression; the two 64-term sums run j = 0..63
 * in order; `p_dac[i] = error` converts like `p_dac[i] = corr[0]` in row f2.
 * ====================================================================================== */
#define ORC_ANR_DLINE 512
#define ORC_ANR_TAPS 64
#define ORC_ANR_DELAY 16
typedef struct {
    float lidx, ngamma;            /* ANR_lidx = 120, ANR_ngamma = 0.001 (:715, :718) */
    int32_t in_idx;                /* ANR_in_idx (:723) */
    float d[ORC_ANR_DLINE];        /* ANR_d (:724) */
    float w[ORC_ANR_DLINE];        /* ANR_w (:725; only the first 64 are used) */
} orc_anr;
void orc_anr_init(orc_anr *a);
/* anr_on: 1 = notch filter (output = error), 2 = noise reduction (output = y); 0 = off (data untouched, state untouched) */
void orc_anr_q15(orc_anr *a, int anr_on, int16_t *data, uint32_t n);

/* ======================================================================================
 * Rows f2 / f3 inside the FP32 chain (an extension: the reference runs both on int16 samples
 * only).  The same statements as orc_syncam_q15 / orc_anr_q15 on float samples, nothing
 * truncated.  The PLL is scale-free; the LMS filter has absolute constants (1e-10 under the
 * normalisation), so it works in the reference's int16 units: sample x 32768 in, / 32768 out
 * (both exact).  UNPINNED by construction (no reference counterpart); the GPU flavour is checked
 * against this restatement and both against the q15 functions on integer-valued input.
 * ====================================================================================== */
float orc_syncam_step_f32(orc_syncam *s, float I, float Q);                    /* one sample: corr[0] */
void orc_syncam_f32(orc_syncam *s, const float *I, const float *Q, float *out, uint32_t n);
float orc_anr_step_f32(orc_anr *a, int anr_on, float sample);                  /* one sample, full scale = 1.0 */
void orc_anr_f32(orc_anr *a, int anr_on, float *data, uint32_t n);
typedef struct {
    int32_t pll;                   /* != 0: ORC_SYNCAM channels demodulate through the PLL (else like AM) */
    int32_t anr_on;                /* 0 off, 1 notch, 2 noise reduction: between demodulator and biquad cascade (:702-770) */
    orc_syncam pll_state;
    orc_anr anr_state;
} orc_chain_f32_post;
void orc_chain_f32_post_init(orc_chain_f32_post *p, int pll, int anr_on);
/* orc_chain_f32 with the PLL branch and the LMS filter in their places; post == NULL: orc_chain_f32 itself */
void orc_chain_f32_post_run(const orc_chain_f32_cfg *cfg, orc_chain_f32_state *st, orc_chain_f32_post *post, const int16_t *x,
                            float *audio, uint64_t n);

/* ---- Row f4 (first half): what AudioOutputAnalog::isr hands the 12-bit DAC, src/Audio/output_dac.cpp:139-151:
 * ((sample) + 32768) >> 4 per sample; 2048 (mid-scale) when no block arrived.  UNPINNED (needs the Teensyduino core). */
void orc_dac_format(const int16_t *src /* NULL = no block */, int16_t *dest, uint32_t n);

/* ---- Row f4 (second half): the spectrum display's 128-point q15 real FFT, UI.cpp:520-592 -> arm_rfft_q15.c,
 * arm_cfft_q15.c, arm_cfft_radix4_q15.c (ARM_MATH_DSP branches), arm_bitreversal2.S.  PINNED stage by stage against the
 * compiled reference (oracle/_ref): butterfly + bit reversal vs arm_cfft_radix4_q15, split vs arm_split_rfft_q15, tables
 * vs arm_common_tables.c / arm_rfft_init_q15.c; the column heights (UI.cpp:557-572) are three lines, unpinned. */
void orc_fft_tables(int16_t twiddle64[96], int16_t coefA[128], int16_t coefB[128]);
uint32_t orc_bitrev_table64(uint16_t table[56]);
void orc_bitreversal_16(int16_t *buf, uint32_t bitRevLen, const uint16_t *table);
void orc_radix4_butterfly64_q15(int16_t *p, const int16_t *tw);
void orc_split_rfft64_q15(const int16_t *X, const int16_t *A, const int16_t *B, int16_t *dst);
void orc_rfft128_q15(const int16_t *data, int16_t *fft_out /* 256 */, int16_t *work /* 128 or NULL */);
void orc_spectrum_columns(const int16_t *fft_out, uint8_t *y_new /* 127 */);
int orc_spectrum_tick(int spectrum_on, int *counter);

#ifdef __cplusplus
}
#endif
#endif
