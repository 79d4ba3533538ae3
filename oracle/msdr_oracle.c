/*
 * oracle/msdr_oracle.c -- CPU restatement of Minimal-SDR's per-block demodulation chain.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see msdr_oracle.h for who may load it and for the
 * pinned / "parity unpinned" status of every function).  Written from the reference's
 * arithmetic contract: most functions are the per-sample formula the cited lines
 * implement, in this file's own words.  Two parts necessarily follow the sketch statement by statement,
 * because their results depend on its order of float operations and must match it bit for bit -- the tap
 * designer (orc_calc_fir_coeffs / orc_m_sinc / orc_izero after Minimal-SDR.ino:782-899) and the LMS filter's
 * update (after Minimal-SDR.ino:702-770): about twenty lines there are close to the sketch's own.
 * Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off -fwrapv -fno-strict-aliasing [-fopenmp]).
 */
#include "msdr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
/* The batch drivers double as the timed CPU baseline: the two fp32 FIR loops are compiled once per vector ISA and picked at load
 * time (the library is built in one container and run on another host, so -march=native is out). */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define ORC_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define ORC_CLONES
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int32_t ssat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
static inline int32_t wrap_add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

/* ======================================================================================
 * A1  Minimal-SDR.ino:546-558  multiplication-free Fs/4 mixer.
 * I = x*{1,0,-1,0}, Q = x*{0,1,0,-1}; the negate is formed in int and narrowed to int16,
 * so -(-32768) stays -32768.  A block always starts at n = 0 (mod 4) (B = 128).
 * ====================================================================================== */
void orc_mix_fs4_q15(const int16_t *x, int16_t *I, int16_t *Q, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        int16_t neg = (int16_t)(uint16_t)(-(int32_t)x[i]);
        switch (i & 3u) {
        case 0: I[i] = x[i]; Q[i] = 0;    break;
        case 1: I[i] = 0;    Q[i] = x[i]; break;
        case 2: I[i] = neg;  Q[i] = 0;    break;
        default: I[i] = 0;   Q[i] = neg;  break;
        }
    }
}

/* ======================================================================================
 * A2  freq_conv.cpp:30-116.  The three vector ops are CMSIS-DSP V1.5.x functions whose
 * sources are NOT in the reference (un-vendored; Teensyduino ships them prebuilt).  Their
 * published definitions: mult = sat16((a*b)>>15), add/sub = sat16(a +/- b).
 * ====================================================================================== */
void orc_mult_q15(const q15_t *a, const q15_t *b, q15_t *dst, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) dst[i] = (q15_t)ssat16(((int32_t)a[i] * b[i]) >> 15);
}
void orc_add_q15(const q15_t *a, const q15_t *b, q15_t *dst, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) dst[i] = (q15_t)ssat16((int32_t)a[i] + b[i]);
}
void orc_sub_q15(const q15_t *a, const q15_t *b, q15_t *dst, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) dst[i] = (q15_t)ssat16((int32_t)a[i] - b[i]);
}

/* `pass` is inverted in the reference: pass == 0 forwards the inputs untouched
 * (freq_conv.cpp:49-56), the default pass(1) processes (freq_conv.h:39).
 * dir == 0 (:67-84):  I' = I*oq + Q*oi,  Q' = Q*oq - I*oi
 * dir == 1 (:86-103): Q' = Q*oq + I*oi,  I' = I*oq - Q*oi
 * All four products are formed from the ORIGINAL I,Q before either is overwritten. */
void orc_freqconv_q15(q15_t *I, q15_t *Q, const q15_t *osc_i, const q15_t *osc_q,
                      int dir, int pass, uint32_t n)
{
    if (!pass) return;
    for (uint32_t i = 0; i < n; i++) {
        int32_t Ioq = ssat16(((int32_t)I[i] * osc_q[i]) >> 15);
        int32_t Ioi = ssat16(((int32_t)I[i] * osc_i[i]) >> 15);
        int32_t Qoq = ssat16(((int32_t)Q[i] * osc_q[i]) >> 15);
        int32_t Qoi = ssat16(((int32_t)Q[i] * osc_i[i]) >> 15);
        if (!dir) { I[i] = (q15_t)ssat16(Ioq + Qoi); Q[i] = (q15_t)ssat16(Qoq - Ioi); }
        else      { Q[i] = (q15_t)ssat16(Qoq + Ioi); I[i] = (q15_t)ssat16(Ioq - Qoi); }
    }
}
void orc_freqconv_f32(float *I, float *Q, const float *osc_i, const float *osc_q,
                      int dir, int pass, uint32_t n)
{
    if (!pass) return;
    for (uint32_t i = 0; i < n; i++) {
        float Ioq = I[i] * osc_q[i], Ioi = I[i] * osc_i[i];
        float Qoq = Q[i] * osc_q[i], Qoi = Q[i] * osc_i[i];
        if (!dir) { I[i] = Ioq + Qoi; Q[i] = Qoq - Ioi; }
        else      { Q[i] = Qoq + Ioi; I[i] = Ioq - Qoi; }
    }
}

/* ======================================================================================
 * A3  arm_fir_init_q15.c:78-138 (Cortex-M4 branch): odd numTaps is an argument error
 * (:93-96), state of numTaps+blockSize elements is cleared (:106).
 * ====================================================================================== */
int orc_fir_init_q15(orc_fir_instance_q15 *S, uint16_t numTaps, const q15_t *pCoeffs,
                     q15_t *pState, uint32_t blockSize)
{
    if (numTaps & 1u) return ORC_ARGUMENT_ERROR;
    S->numTaps = numTaps;
    S->pCoeffs = pCoeffs;
    memset(pState, 0, ((size_t)numTaps + blockSize) * sizeof(q15_t));
    S->pState = pState;
    return ORC_SUCCESS;
}

/* ======================================================================================
 * A4  arm_fir_fast_q15.c:60-329.
 *   y[n] = sat16( wrap32( sum_{k=0}^{N-1} pCoeffs[k] * w[n+k] ) >> 15 ),
 * w = [N-1 history samples | the block].  32-bit WRAPPING accumulator (doc :49-53),
 * arithmetic shift, saturate on store (:234-238, :287); tail of the window is carried to
 * the front of pState for the next call (:296-327).
 * ====================================================================================== */
void orc_fir_fast_q15(const orc_fir_instance_q15 *S, const q15_t *pSrc, q15_t *pDst,
                      uint32_t blockSize)
{
    const uint32_t N = S->numTaps;
    q15_t *w = S->pState;
    memcpy(w + (N - 1), pSrc, blockSize * sizeof(q15_t));
    for (uint32_t n = 0; n < blockSize; n++) {
        uint32_t acc = 0;
        for (uint32_t k = 0; k < N; k++)
            acc += (uint32_t)((int32_t)S->pCoeffs[k] * (int32_t)w[n + k]);
        pDst[n] = (q15_t)ssat16(((int32_t)acc) >> 15);
    }
    memmove(w, w + blockSize, (N - 1) * sizeof(q15_t));
}

void orc_copy_q15(const q15_t *src, q15_t *dst, uint32_t n) { memmove(dst, src, n * sizeof(q15_t)); }

/* ======================================================================================
 * arm_sqrt_q31.c:50-138  Newton inverse-sqrt in Q31 (Teensy 3.2 AM branch).
 * ====================================================================================== */
int orc_sqrt_q31(q31_t in, q31_t *out)
{
    if (in <= 0) { *out = 0; return ORC_ARGUMENT_ERROR; }
    int32_t number = in;
    int32_t sign_bits = (int32_t)(uint8_t)__builtin_clz((uint32_t)number) - 1;
    int32_t sh = (sign_bits % 2 == 0) ? sign_bits : sign_bits - 1;
    number = (int32_t)((uint32_t)number << sh);
    int32_t half = number >> 1, temp1 = number;
    union { int32_t i; float f; } cv;
    cv.f = (float)number * 4.6566128731e-010f;
    cv.i = 0x5f3759df - (cv.i >> 1);
    int32_t var1 = (int32_t)(cv.f * 1073741824);
    for (int it = 0; it < 3; it++) {
        int32_t sq = (int32_t)(((int64_t)var1 * var1) >> 31);
        int32_t t  = (int32_t)(((int64_t)sq * (int64_t)half) >> 31);
        var1 = (int32_t)((uint32_t)((int32_t)(((int64_t)var1 * (0x30000000 - t)) >> 31)) << 2);
    }
    var1 = (int32_t)((uint32_t)((int32_t)(((int64_t)temp1 * var1) >> 31)) << 1);
    *out = var1 >> (sh / 2);
    return ORC_SUCCESS;
}

/* ======================================================================================
 * A5  Minimal-SDR.ino:589-691.
 *  LSB :591-596  (int16)(I - Q)      USB :598-604  (int16)(I + Q)   -- truncating store
 *  AM/CW T3.6 :607-616  (int16)trunc(sqrtf((float)(I*I + Q*Q))), sum formed in int32
 *             (arm_sqrt_f32 returns 0 for negative input, arm_math.h:5733-5758)
 *  AM/CW T3.2 :618-627  arm_sqrt_q31(I*I+Q*Q) >> 16
 * Out-of-range float->int16 (only reachable with |I|,|Q| > 23170 together) is taken as
 * convert-to-int32-then-truncate, which is what both ARM (vcvt + strh) and x86 do.
 * ====================================================================================== */
/* arm_sqrt_f32, arm_math.h:5733-5758: sqrtf for in >= 0, else 0 (and ARM_MATH_ARGUMENT_ERROR, which the sketch ignores, .ino:612).
 * Pinned against the header's own inline (oracle/build_ref.sh step 7) through the hook below. */
static inline float am_sqrt_f32(float f) { return (f >= 0.0f) ? sqrtf(f) : 0.0f; }
float orc_prim_sqrt_f32(float in) { return am_sqrt_f32(in); }
void orc_demod_q15(int mode, int sqrt_kind, const int16_t *I, const int16_t *Q,
                   int16_t *out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        int32_t a = I[i], b = Q[i];
        switch (mode) {
        case ORC_LSB: out[i] = (int16_t)(uint16_t)(uint32_t)(a - b); break;
        case ORC_USB: out[i] = (int16_t)(uint16_t)(uint32_t)(a + b); break;
        default: {
            int32_t s = wrap_add32(a * a, b * b);
            if (sqrt_kind == ORC_SQRT_Q31) {
                q31_t r; orc_sqrt_q31(s, &r);
                out[i] = (int16_t)(r >> 16);
            } else {
                float f = (float)s;
                float r = am_sqrt_f32(f);
                out[i] = (int16_t)(uint16_t)(uint32_t)(int32_t)r;
            }
        } break;
        }
    }
}
void orc_demod_f32(int mode, const float *I, const float *Q, float *out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        switch (mode) {
        case ORC_LSB: out[i] = I[i] - Q[i]; break;
        case ORC_USB: out[i] = I[i] + Q[i]; break;
        default:      out[i] = sqrtf(I[i] * I[i] + Q[i] * Q[i]); break;
        }
    }
}

/* ======================================================================================
 * A6  arm_fir_f32 -- CMSIS-DSP V1.5.x, source NOT in the reference (prototype
 * arm_math.h:1182-1186, init :1197-1202, instance :1047-1052).  Published algorithm:
 * y[n] = sum_k pCoeffs[k]*pState[n+k], pState = [N-1 history | block], each output
 * accumulated from 0 in ascending k, separate multiply and add.   PARITY UNPINNED.
 * ====================================================================================== */
void orc_fir_init_f32(orc_fir_instance_f32 *S, uint16_t numTaps, const float *pCoeffs,
                      float *pState, uint32_t blockSize)
{
    S->numTaps = numTaps; S->pCoeffs = pCoeffs; S->pState = pState;
    memset(pState, 0, ((size_t)numTaps + blockSize - 1u) * sizeof(float));
}
#define ORC_FB 16u
ORC_CLONES void orc_fir_f32(const orc_fir_instance_f32 *S, const float *pSrc, float *pDst,
                 uint32_t blockSize)
{
    const uint32_t N = S->numTaps;
    float *w = S->pState;
    memcpy(w + (N - 1), pSrc, blockSize * sizeof(float));
    uint32_t n = 0;
    /* ORC_FB outputs side by side: each output's sum is still built from 0 in ascending k, one separately rounded multiply and
     * one add per tap (the compiler may put the ORC_FB independent chains into vector lanes; it may not reorder any of them) */
    for (; n + ORC_FB <= blockSize; n += ORC_FB) {
        float acc[ORC_FB];
        for (uint32_t j = 0; j < ORC_FB; j++) acc[j] = 0.0f;
        for (uint32_t k = 0; k < N; k++) {
            const float c = S->pCoeffs[k];
            for (uint32_t j = 0; j < ORC_FB; j++) acc[j] += w[n + j + k] * c;
        }
        for (uint32_t j = 0; j < ORC_FB; j++) pDst[n + j] = acc[j];
    }
    for (; n < blockSize; n++) {
        float acc = 0.0f;
        for (uint32_t k = 0; k < N; k++) acc += w[n + k] * S->pCoeffs[k];
        pDst[n] = acc;
    }
    memmove(w, w + blockSize, (N - 1) * sizeof(float));
}

/* ======================================================================================
 * A7  src/Audio/filter_biquad.cpp:33-100 with src/Audio/utility/dspinst.h:34-51,235-249.
 * Stage record = 8 words {b0,b1,b2,-a1,-a2, x[n-1]:x[n-2], y[n-1]:y[n-2], residue|flag}.
 * Per sample:  acc = residue + sum of five (coef * s16) >> 16   (top 32 of the 48-bit
 * product, i.e. smlawb/smlawt; 32-bit wrapping adds), y = ssat16(acc >> 14),
 * residue = acc & 0x3FFF (first-order error feedback).  Stages run one after another over
 * the whole block; bit 31 of word 7 says another stage follows.   PARITY UNPINNED.
 * ====================================================================================== */
static inline int32_t mulw16(int32_t coef, int16_t s) { return (int32_t)(((int64_t)coef * s) >> 16); }
/* Test hooks: the two primitives above exactly as the biquad, the amplifier and the DC blocker below use them, so that
 * tests/test_oracle_pinned.py can hold them against the reference's own plain-C bodies (dspinst.h, KINETISL branch, compiled by
 * oracle/build_ref.sh): smulwb / smulwt = mulw16 on the bottom / top half-word, ssat #16 with asr = ssat16(v >> rshift). */
int32_t orc_prim_mulw16b(int32_t a, uint32_t b) { return mulw16(a, (int16_t)(b & 0xFFFFu)); }
int32_t orc_prim_mulw16t(int32_t a, uint32_t b) { return mulw16(a, (int16_t)(b >> 16)); }
int32_t orc_prim_ssat16_rshift(int32_t v, int rshift) { return ssat16(v >> rshift); }
/* history words 5 / 6 of a stage record: the newer sample in the top half (filter_biquad.cpp:72-73: pack_16b_16b(newer, older)) */
static inline int32_t pack_hist(int16_t newer, int16_t older) { return (int32_t)(((uint32_t)(uint16_t)newer << 16) | (uint16_t)older); }
uint32_t orc_prim_pack_hist(int32_t newer, int32_t older) { return (uint32_t)pack_hist((int16_t)newer, (int16_t)older); }

void orc_biquad_teensy_init(orc_biquad_teensy *b) { memset(b->definition, 0, sizeof b->definition); }

void orc_biquad_teensy_set_coefficients(orc_biquad_teensy *b, uint32_t stage, const int32_t coef[5])
{
    if (stage >= 4) return;                                   /* cpp:86 */
    int32_t *d = b->definition + (stage << 3);
    if (stage > 0) d[-1] = (int32_t)((uint32_t)d[-1] | 0x80000000u);  /* cpp:89 */
    d[0] = coef[0]; d[1] = coef[1]; d[2] = coef[2];
    d[3] = (int32_t)(0u - (uint32_t)coef[3]);                 /* cpp:93-94: stored negated */
    d[4] = (int32_t)(0u - (uint32_t)coef[4]);
    /* words 5,6 (history) are deliberately kept (cpp:95-97) */
    d[7] = (int32_t)((uint32_t)d[7] & 0x80000000u);           /* cpp:98 */
}

void orc_biquad_teensy_update(orc_biquad_teensy *b, int16_t *data, uint32_t n)
{
    int32_t *st = b->definition;
    uint32_t flag;
    do {
        const int32_t b0 = st[0], b1 = st[1], b2 = st[2], a1 = st[3], a2 = st[4];
        int16_t x1 = (int16_t)((uint32_t)st[5] >> 16), x2 = (int16_t)(st[5] & 0xFFFF);
        int16_t y1 = (int16_t)((uint32_t)st[6] >> 16), y2 = (int16_t)(st[6] & 0xFFFF);
        int32_t res = st[7] & 0x3FFF;
        for (uint32_t i = 0; i < n; i++) {
            int16_t x0 = data[i];
            int32_t acc = res;
            acc = wrap_add32(acc, mulw16(b0, x0));
            acc = wrap_add32(acc, mulw16(b1, x1));
            acc = wrap_add32(acc, mulw16(b2, x2));
            acc = wrap_add32(acc, mulw16(a1, y1));
            acc = wrap_add32(acc, mulw16(a2, y2));
            int16_t y0 = (int16_t)ssat16(acc >> 14);
            res = acc & 0x3FFF;
            x2 = x1; x1 = x0; y2 = y1; y1 = y0;
            data[i] = y0;
        }
        flag = (uint32_t)st[7] & 0x80000000u;
        st[5] = pack_hist(x1, x2);
        st[6] = pack_hist(y1, y2);
        st[7] = (int32_t)((uint32_t)res | flag);
        st += 8;
    } while (flag);
}

/* filter_biquad.h:56-149 -- Audio-EQ-cookbook designers; coefficients scaled by 2^30 and
 * truncated to int; a1,a2 in textbook sign (negated later by setCoefficients). */
void orc_biquad_design(int kind, float frequency, float q_or_gain, float slope,
                       double sample_rate, int32_t coef[5])
{
    double w0 = frequency * (2 * 3.141592654 / sample_rate);
    double sinW0 = sin(w0), cosW0 = cos(w0);
    if (kind <= ORC_BQ_NOTCH) {
        double alpha = sinW0 / ((double)q_or_gain * 2.0);
        double scale = 1073741824.0 / (1.0 + alpha);
        switch (kind) {
        case ORC_BQ_LOWPASS:
            coef[0] = (int32_t)(((1.0 - cosW0) / 2.0) * scale);
            coef[1] = (int32_t)((1.0 - cosW0) * scale);
            coef[2] = coef[0];
            break;
        case ORC_BQ_HIGHPASS:
            coef[0] = (int32_t)(((1.0 + cosW0) / 2.0) * scale);
            coef[1] = (int32_t)(-(1.0 + cosW0) * scale);
            coef[2] = coef[0];
            break;
        case ORC_BQ_BANDPASS:
            coef[0] = (int32_t)(alpha * scale);
            coef[1] = 0;
            coef[2] = (int32_t)((-alpha) * scale);
            break;
        default: /* notch */
            coef[0] = (int32_t)scale;
            coef[1] = (int32_t)((-2.0 * cosW0) * scale);
            coef[2] = coef[0];
            break;
        }
        coef[3] = (int32_t)((-2.0 * cosW0) * scale);
        coef[4] = (int32_t)((1.0 - alpha) * scale);
        return;
    }
    double a = pow(10.0, q_or_gain / 40.0);
    double sinsq = sinW0 * sqrt((pow(a, 2.0) + 1.0) * (1.0 / slope - 1.0) + 2.0 * a);
    double aMinus = (a - 1.0) * cosW0, aPlus = (a + 1.0) * cosW0;
    if (kind == ORC_BQ_LOWSHELF) {
        double scale = 1073741824.0 / ((a + 1.0) + aMinus + sinsq);
        coef[0] = (int32_t)(a * ((a + 1.0) - aMinus + sinsq) * scale);
        coef[1] = (int32_t)(2.0 * a * ((a - 1.0) - aPlus) * scale);
        coef[2] = (int32_t)(a * ((a + 1.0) - aMinus - sinsq) * scale);
        coef[3] = (int32_t)(-2.0 * ((a - 1.0) + aPlus) * scale);
        coef[4] = (int32_t)(((a + 1.0) + aMinus - sinsq) * scale);
    } else {
        double scale = 1073741824.0 / ((a + 1.0) - aMinus + sinsq);
        coef[0] = (int32_t)(a * ((a + 1.0) + aMinus + sinsq) * scale);
        coef[1] = (int32_t)(-2.0 * a * ((a - 1.0) + aPlus) * scale);
        coef[2] = (int32_t)(a * ((a + 1.0) + aMinus - sinsq) * scale);
        coef[3] = (int32_t)(2.0 * ((a - 1.0) - aPlus) * scale);
        coef[4] = (int32_t)(((a + 1.0) - aMinus - sinsq) * scale);
    }
}

/* ======================================================================================
 * A8  arm_biquad_cascade_df1_f32 -- CMSIS-DSP V1.5.x, source NOT in the reference
 * (prototype arm_math.h:1333-1337, init :1347-1351, instance :1230-1235).  Published
 * algorithm: y = b0*x + b1*x1 + b2*x2 + a1*y1 + a2*y2 evaluated left to right, feedback
 * terms ADDED (caller passes -a1,-a2 of the textbook form); stage s feeds stage s+1;
 * state per stage {x1,x2,y1,y2}.   PARITY UNPINNED.
 * ====================================================================================== */
void orc_biquad_df1_init_f32(orc_biquad_df1_f32 *S, uint8_t numStages, const float *pCoeffs,
                             float *pState)
{
    S->numStages = numStages; S->pCoeffs = pCoeffs; S->pState = pState;
    memset(pState, 0, 4u * numStages * sizeof(float));
}
void orc_biquad_df1_f32_run(const orc_biquad_df1_f32 *S, const float *pSrc, float *pDst,
                            uint32_t blockSize)
{
    const float *in = pSrc;
    for (uint32_t s = 0; s < S->numStages; s++) {
        const float *c = S->pCoeffs + 5 * s;
        float *st = S->pState + 4 * s;
        float x1 = st[0], x2 = st[1], y1 = st[2], y2 = st[3];
        for (uint32_t i = 0; i < blockSize; i++) {
            float x0 = in[i];
            float y0 = (c[0] * x0) + (c[1] * x1) + (c[2] * x2) + (c[3] * y1) + (c[4] * y2);
            x2 = x1; x1 = x0; y2 = y1; y1 = y0;
            pDst[i] = y0;
        }
        st[0] = x1; st[1] = x2; st[2] = y1; st[3] = y2;
        in = pDst;
    }
    if (S->numStages == 0 && pDst != pSrc) memcpy(pDst, pSrc, blockSize * sizeof(float));
}

/* ======================================================================================
 * A9  Minimal-SDR.ino:782-899  Kaiser-window FIR designer.  Operand types are kept exactly
 * (float vs double per C promotion rules in the sketch) because the result is truncated to
 * int16.  `PI`: arm_math.h:365-367 gives the float fallback 3.14159265358979f; on a Teensy
 * Arduino.h (un-vendored) supplies a double literal first.  orc_set_pi_double(1) selects the
 * latter; default is the float fallback (what the vendored sources alone give; both agree
 * on every tap set in tests/golden).
 * ====================================================================================== */
static int g_pi_double = 0;
void orc_set_pi_double(int on) { g_pi_double = on; }
#define PI_F 3.14159265358979f
#define PI_D 3.1415926535897932384626433832795

float orc_izero(float x)
{
    static const float errorlimit = 1e-9;
    float x2 = x / 2.0;
    float summe = 1.0, ds = 1.0, di = 1.0, tmp;
    do {
        tmp = x2 / di;
        tmp *= tmp;
        ds *= tmp;
        summe += ds;
        di += 1.0;
    } while (ds >= errorlimit * summe);
    return summe;
}

float orc_m_sinc(int m, float fc)
{
    if (m == 0) return 1.0f;
    float x = g_pi_double ? (float)(m * (PI_D / 2)) : (m * (PI_F / 2));
    return sinf(x * fc) / (fc * x);
}

static float pih_times(int k, float fc) /* PIH * k * fc, :862,:866 */
{
    return g_pi_double ? (float)((PI_D / 2) * k * fc) : ((PI_F / 2) * k * fc);
}

void orc_calc_fir_coeffs(int16_t *coeffs, int numCoeffs, float fc, float Astop, int type,
                         float dfc, float Fsamprate)
{
    int ii, jj;
    float Beta, izb, fcf = fc;
    int nc = numCoeffs;
    fc = fc / Fsamprate;
    dfc = dfc / Fsamprate;
    if (Astop < 20.96) Beta = 0.0;                                          /* :801-806 */
    else if (Astop >= 50.0) Beta = 0.1102 * (Astop - 8.71);
    else Beta = 0.5842 * powf((float)(Astop - 20.96), (float)0.4) + 0.07886 * (Astop - 20.96);
    izb = orc_izero(Beta);
    switch (type) {
    case 0: fcf = fc * 2.0; nc = numCoeffs; break;                          /* :812-815 */
    case 1: fcf = -fc; nc = 2 * (numCoeffs / 2); break;
    case 2:
    case 3: fcf = dfc; nc = 2 * (numCoeffs / 2); break;
    case 4: {                                                               /* :828-845 */
        nc = 2 * (numCoeffs / 2);
        for (ii = 0; ii < 2 * (nc - 1); ii++) coeffs[ii] = 0;
        coeffs[nc] = 1;
        for (ii = 1; ii < (nc + 1); ii += 2) {
            if (2 * ii == nc) continue;
            float x = (float)(2 * ii - nc) / (float)nc;
            float w = orc_izero(Beta * sqrtf(1.0f - x * x)) / izb;
            if (g_pi_double)
                coeffs[2 * ii + 1] = (int16_t)(int32_t)(32767 * (1.0f / ((PI_D / 2) * (float)(ii - nc / 2)) * w));
            else
                coeffs[2 * ii + 1] = (int16_t)(int32_t)(32767 * (1.0f / ((PI_F / 2) * (float)(ii - nc / 2)) * w));
        }
        return;
    }
    default: return;
    }
    for (ii = -nc, jj = 0; ii < nc; ii += 2, jj++) {                         /* :850-855 */
        float x = (float)ii / (float)nc;
        float w = orc_izero(Beta * sqrtf(1.0f - x * x)) / izb;
        coeffs[jj] = (int16_t)(int32_t)(fcf * orc_m_sinc(ii, fcf) * w * 32767);
    }
    switch (type) {                                                          /* :857-871 */
    case 1: coeffs[nc / 2] += 1; break;
    case 2:
        for (jj = 0; jj < nc + 1; jj++)
            coeffs[jj] = (int16_t)(int32_t)(coeffs[jj] * (2.0f * cosf(pih_times(2 * jj - nc, fc))));
        break;
    case 3:
        for (jj = 0; jj < nc + 1; jj++)
            coeffs[jj] = (int16_t)(int32_t)(coeffs[jj] * (-2.0f * cosf(pih_times(2 * jj - nc, fc))));
        coeffs[nc / 2] += 1;
        break;
    }
}

/* ======================================================================================
 * Whole chain, q15: demodulation() Minimal-SDR.ino:518-775 (mix :546-558, FIR pair
 * :574-575, copy :577-578, demod :589-691) followed by the AudioFilterBiquad nodes wired
 * after queue_dac (.ino:77-81), one update() per block each.
 * mixer == 1 substitutes the AudioEffectFreqConv node (A2) fed with the real IF on port 0,
 * zeros on port 1, dir = 1  =>  I = x*osc_q, Q = x*osc_i.
 * ====================================================================================== */
void orc_chain_q15(const orc_chain_q15_cfg *cfg, orc_chain_q15_state *st, const int16_t *x,
                   int16_t *audio, int16_t *i_out, int16_t *q_out, uint32_t n_blocks)
{
    orc_fir_instance_q15 FI = { (uint16_t)cfg->num_taps, st->state_i, cfg->coeffs_i };
    orc_fir_instance_q15 FQ = { (uint16_t)cfg->num_taps, st->state_q, cfg->coeffs_q };
    int16_t I[ORC_BLOCK], Q[ORC_BLOCK], If[ORC_BLOCK], Qf[ORC_BLOCK];
    for (uint32_t b = 0; b < n_blocks; b++) {
        const int16_t *xb = x + (size_t)b * ORC_BLOCK;
        int16_t *ab = audio + (size_t)b * ORC_BLOCK;
        if (cfg->mixer == 0) {
            orc_mix_fs4_q15(xb, I, Q, ORC_BLOCK);
        } else {
            memcpy(I, xb, sizeof I);
            memset(Q, 0, sizeof Q);
            orc_freqconv_q15(I, Q, cfg->osc_i, cfg->osc_q, 1, 1, ORC_BLOCK);
        }
        orc_fir_fast_q15(&FI, I, If, ORC_BLOCK);
        orc_fir_fast_q15(&FQ, Q, Qf, ORC_BLOCK);
        if (i_out) memcpy(i_out + (size_t)b * ORC_BLOCK, If, sizeof If);
        if (q_out) memcpy(q_out + (size_t)b * ORC_BLOCK, Qf, sizeof Qf);
        orc_demod_q15(cfg->mode, cfg->sqrt_kind, If, Qf, ab, ORC_BLOCK);
        for (uint32_t k = 0; k < cfg->n_biquad_nodes; k++)
            orc_biquad_teensy_update(&st->bq[k], ab, ORC_BLOCK);
    }
}

/* ======================================================================================
 * Whole chain, fp32 (the north-star flavour): int16 IF -> float (x * in_scale) ->
 * NCO mix I = x*osc_q[n % P], Q = x*osc_i[n % P] (freq_conv dir=1, Q-in = 0, in fp32) ->
 * arm_fir_f32 pair (A6) -> demod (A5 in fp32) -> arm_biquad_cascade_df1_f32 (A8).
 * Sample-sequential; every product and sum is a separately rounded fp32 operation.
 * ====================================================================================== */
static void orc_chain_f32_core(const orc_chain_f32_cfg *cfg, orc_chain_f32_state *st, orc_chain_f32_post *post, const int16_t *x,
                               float *audio, uint64_t n);
void orc_chain_f32(const orc_chain_f32_cfg *cfg, orc_chain_f32_state *st, const int16_t *x, float *audio, uint64_t n)
{
    orc_chain_f32_core(cfg, st, NULL, x, audio, n);
}
void orc_chain_f32_post_run(const orc_chain_f32_cfg *cfg, orc_chain_f32_state *st, orc_chain_f32_post *post, const int16_t *x,
                            float *audio, uint64_t n)
{
    orc_chain_f32_core(cfg, st, post, x, audio, n);
}
void orc_chain_f32_post_init(orc_chain_f32_post *p, int pll, int anr_on)
{
    p->pll = pll; p->anr_on = anr_on;
    orc_syncam_init(&p->pll_state);
    orc_anr_init(&p->anr_state);
}
ORC_CLONES static void orc_chain_f32_core(const orc_chain_f32_cfg *cfg, orc_chain_f32_state *st, orc_chain_f32_post *post, const int16_t *x,
                               float *audio, uint64_t n)
{
    const uint32_t N = cfg->num_taps, H = N - 1;
    /* circular-free sliding window: keep 2 copies so a window is always contiguous */
    const uint32_t CH = 4096;
    float *wi = (float *)malloc((size_t)(H + CH) * sizeof(float));
    float *wq = (float *)malloc((size_t)(H + CH) * sizeof(float));
    float *fi = (float *)malloc((size_t)CH * sizeof(float));
    float *fq = (float *)malloc((size_t)CH * sizeof(float));
    memcpy(wi, st->hist_i, H * sizeof(float));
    memcpy(wq, st->hist_q, H * sizeof(float));
    float *bs = st->bq_state;
    for (uint64_t base = 0; base < n; base += CH) {
        uint32_t len = (uint32_t)((n - base < CH) ? (n - base) : CH);
        for (uint32_t i = 0; i < len; i++) {
            float xf = (float)x[base + i] * cfg->in_scale;
            uint32_t ph = (uint32_t)((st->n0 + base + i) % cfg->osc_len);
            wi[H + i] = xf * cfg->osc_q[ph];
            wq[H + i] = xf * cfg->osc_i[ph];
        }
        /* the FIR pair, ORC_FB outputs side by side (see orc_fir_f32: every output's own sum keeps its order and roundings) */
        uint32_t i0 = 0;
        for (; i0 + ORC_FB <= len; i0 += ORC_FB) {
            float ai[ORC_FB], aq[ORC_FB];
            for (uint32_t j = 0; j < ORC_FB; j++) { ai[j] = 0.0f; aq[j] = 0.0f; }
            for (uint32_t k = 0; k < N; k++) {
                const float ci = cfg->coeffs_i[k], cq = cfg->coeffs_q[k];
                for (uint32_t j = 0; j < ORC_FB; j++) { ai[j] += wi[i0 + j + k] * ci; aq[j] += wq[i0 + j + k] * cq; }
            }
            for (uint32_t j = 0; j < ORC_FB; j++) { fi[i0 + j] = ai[j]; fq[i0 + j] = aq[j]; }
        }
        for (; i0 < len; i0++) {
            float ai = 0.0f, aq = 0.0f;
            for (uint32_t k = 0; k < N; k++) {
                ai += wi[i0 + k] * cfg->coeffs_i[k];
                aq += wq[i0 + k] * cfg->coeffs_q[k];
            }
            fi[i0] = ai; fq[i0] = aq;
        }
        for (uint32_t i = 0; i < len; i++) {
            const float ai = fi[i], aq = fq[i];
            float d;
            switch (cfg->mode) {
            case ORC_LSB: d = ai - aq; break;
            case ORC_USB: d = ai + aq; break;
            default:
                if (post && post->pll && cfg->mode == ORC_SYNCAM) d = orc_syncam_step_f32(&post->pll_state, ai, aq);   /* .ino:631-688 */
                else d = sqrtf(ai * ai + aq * aq);
                break;
            }
            if (post && post->anr_on > 0) d = orc_anr_step_f32(&post->anr_state, post->anr_on, d);                   /* .ino:702-770 */
            for (uint32_t s = 0; s < cfg->num_stages; s++) {
                const float *c = cfg->bq_coeffs + 5 * s;
                float *q = bs + 4 * s;
                float y = (c[0] * d) + (c[1] * q[0]) + (c[2] * q[1]) + (c[3] * q[2]) + (c[4] * q[3]);
                q[1] = q[0]; q[0] = d; q[3] = q[2]; q[2] = y;
                d = y;
            }
            audio[base + i] = d;
        }
        memmove(wi, wi + len, H * sizeof(float));
        memmove(wq, wq + len, H * sizeof(float));
    }
    memcpy(st->hist_i, wi, H * sizeof(float));
    memcpy(st->hist_q, wq, H * sizeof(float));
    st->n0 += n;
    free(wi); free(wq); free(fi); free(fq);
}

int orc_chain_f32_batch(const orc_chain_f32_cfg *cfg, const int32_t *mode_per_channel,
                        const int16_t *x, float *audio, uint32_t channels, uint64_t n,
                        int threads)
{
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = threads > 0 ? threads : omp_get_max_threads();
#endif
    const uint32_t H = cfg->num_taps - 1;
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < (int64_t)channels; c++) {
        orc_chain_f32_cfg lc = *cfg;
        if (mode_per_channel) lc.mode = mode_per_channel[c];
        orc_chain_f32_state st;
        st.hist_i = (float *)calloc(H ? H : 1, sizeof(float));
        st.hist_q = (float *)calloc(H ? H : 1, sizeof(float));
        memset(st.bq_state, 0, sizeof st.bq_state);
        st.n0 = 0;
        orc_chain_f32(&lc, &st, x + (size_t)c * n, audio + (size_t)c * n, n);
        free(st.hist_i); free(st.hist_q);
    }
    return used;
}

/* The FIR stage alone over a block batch, fresh zero state per row, `block` samples per call (CMSIS cadence): the timed CPU
 * baseline of bench.py's FIR record, on the same OpenMP team logic as the chain drivers. */
int orc_fir_f32_batch(const float *coeffs, uint16_t num_taps, const float *x, float *y, uint32_t channels,
                      uint64_t n, uint32_t block, int threads)
{
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = threads > 0 ? threads : omp_get_max_threads();
#endif
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < (int64_t)channels; c++) {
        float *st = (float *)calloc((size_t)num_taps + block, sizeof(float));
        orc_fir_instance_f32 S;
        orc_fir_init_f32(&S, num_taps, coeffs, st, block);
        for (uint64_t o = 0; o < n; o += block) {
            const uint32_t m = (uint32_t)((n - o < block) ? (n - o) : block);
            orc_fir_f32(&S, x + (size_t)c * n + o, y + (size_t)c * n + o, m);
        }
        free(st);
    }
    return used;
}
int orc_fir_q15_batch(const q15_t *coeffs, uint16_t num_taps, const q15_t *x, q15_t *y, uint32_t channels,
                      uint64_t n, uint32_t block, int threads)
{
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = threads > 0 ? threads : omp_get_max_threads();
#endif
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < (int64_t)channels; c++) {
        q15_t *st = (q15_t *)calloc((size_t)num_taps + block, sizeof(q15_t));
        orc_fir_instance_q15 S;
        if (orc_fir_init_q15(&S, num_taps, coeffs, st, block) == 0)
            for (uint64_t o = 0; o + block <= n; o += block)
                orc_fir_fast_q15(&S, x + (size_t)c * n + o, y + (size_t)c * n + o, block);
        free(st);
    }
    return used;
}

int orc_chain_q15_batch(const orc_chain_q15_cfg *cfg, const int32_t *mode_per_channel,
                        const int16_t *x, int16_t *audio, uint32_t channels,
                        uint32_t n_blocks, int threads)
{
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = threads > 0 ? threads : omp_get_max_threads();
#endif
    const size_t n = (size_t)n_blocks * ORC_BLOCK;
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < (int64_t)channels; c++) {
        orc_chain_q15_cfg lc = *cfg;
        if (mode_per_channel) lc.mode = mode_per_channel[c];
        orc_chain_q15_state st;
        st.state_i = (q15_t *)calloc(cfg->num_taps + ORC_BLOCK, sizeof(q15_t));
        st.state_q = (q15_t *)calloc(cfg->num_taps + ORC_BLOCK, sizeof(q15_t));
        for (uint32_t k = 0; k < 2; k++) {
            if (k < cfg->n_biquad_nodes && cfg->bq_init) st.bq[k] = cfg->bq_init[k];
            else orc_biquad_teensy_init(&st.bq[k]);
        }
        orc_chain_q15(&lc, &st, x + c * n, audio + c * n, NULL, NULL, n_blocks);
        free(st.state_i); free(st.state_q);
    }
    return used;
}

/* ======================================================================================
 * Row f1: front end.  See msdr_oracle.h.
 * ====================================================================================== */
/* input_adc.cpp:32 */
#define ORC_COEF_HPF_DCBLOCK (1048300 << 10)

/* dspinst.h:358-368 FRACMUL_SHL(x, y, z): smull t:t2 = x*y; result = (t2 << (z+1)) | (t >>(logical) (31-z)),
 * i.e. bits [62-z : 31-z] of the 64-bit product */
static int32_t fracmul_shl(int32_t x, int32_t y, int z)
{
    int64_t p = (int64_t)x * (int64_t)y;
    uint32_t lo = (uint32_t)p, hi = (uint32_t)((uint64_t)p >> 32);
    return (int32_t)((hi << (z + 1)) | (lo >> (31 - z)));
}

void orc_dcblock_init(orc_dcblock *s, uint16_t first_conversion)
{
    s->hpf_x1 = (int32_t)first_conversion << 14;     /* input_adc.cpp:60-62 */
    s->hpf_y1 = 0;                                   /* :63 */
}

/* input_adc.cpp:198-212 */
void orc_dcblock_update(orc_dcblock *s, const uint16_t *adc, int16_t *out, uint32_t n)
{
    int32_t y1 = s->hpf_y1, x1 = s->hpf_x1;
    for (uint32_t i = 0; i < n; i++) {
        int32_t tmp = (int32_t)adc[i] << 14;                          /* :201-202 */
        int32_t acc = wrap_add32(wrap_add32(y1, (int32_t)(0u - (uint32_t)x1)), tmp);   /* :203-205 */
        y1 = fracmul_shl(acc, ORC_COEF_HPF_DCBLOCK, 1);               /* :206 */
        x1 = tmp;                                                     /* :207 */
        out[i] = (int16_t)ssat16(y1 >> 14);                           /* :208 signed_saturate_rshift(hpf_y1, 16, 14) */
    }
    s->hpf_y1 = y1; s->hpf_x1 = x1;
}

/* mixer.h:75-79 */
int32_t orc_amp_multiplier(float n)
{
    if (n > 32767.0f) n = 32767.0f;
    else if (n < -32767.0f) n = -32767.0f;
    return (int32_t)(n * 65536.0f);
}

/* mixer.cpp:134-159 with applyGain :34-47 (smulwb/smulwt = (mult * s16) >> 16, then ssat 16) */
int orc_amp_update(int32_t mult, int16_t *data, uint32_t n)
{
    if (mult == 0) return 0;                       /* :139-142 nothing transmitted */
    if (mult == 65536) return 1;                   /* :143-149 passed on unchanged */
    for (uint32_t i = 0; i < n; i++) data[i] = (int16_t)ssat16(mulw16(mult, data[i]));
    return 1;
}

void orc_agc_init(orc_agc *a)
{
    memset(a->agc_buffer, 0, sizeof a->agc_buffer);
    a->agc_idx = ORC_AGCBUF_SIZE;                  /* Minimal-SDR.ino:451 */
    a->AGC_val = 0.25f;                            /* :94, :104 */
    a->AGC_on = 1;                                 /* :100 */
    a->multiplier = orc_amp_multiplier(a->AGC_val);/* :385 */
}

/* __SSUB16 (cmsis_gcc.h:1684): per halfword a - b, GE[1:0]/GE[3:2] set when the (exact) signed result >= 0;
 * __SEL (cmsis_gcc.h:2022): per byte, a where GE is set, else b.  Returns the selected word. */
static uint32_t ssub16_sel(uint32_t cmp_a, uint32_t cmp_b, uint32_t sel_a, uint32_t sel_b)
{
    const int ge_lo = (int32_t)(int16_t)(cmp_a & 0xFFFF) - (int32_t)(int16_t)(cmp_b & 0xFFFF) >= 0;
    const int ge_hi = (int32_t)(int16_t)(cmp_a >> 16) - (int32_t)(int16_t)(cmp_b >> 16) >= 0;
    return ((ge_lo ? sel_a : sel_b) & 0x0000FFFFu) | ((ge_hi ? sel_a : sel_b) & 0xFFFF0000u);
}

/* Minimal-SDR.ino:446-515.  All arithmetic as the source states it under standard C++: `int` words for the packed
 * min/max (so the odd-sample halves start at -1 / 0, :457-458), abs() on the whole word (:475-476), float for
 * f / fagc / AGC_val, double for the comparisons against 1.3, 0.1, 0.6 ... (unsuffixed literals).  The store to
 * agc_buffer[-1] on every 26th call (:480-481 decrement, then test) is outside the array: it is dropped here. */
void orc_agc_block(orc_agc *a, const int16_t *block)
{
    if (!a->AGC_on) return;
    int32_t minv = 32767, maxv = -32767;
    for (int i = 0; i < ORC_BLOCK / 2; i++) {
        uint32_t data = (uint32_t)(uint16_t)block[2 * i] | ((uint32_t)(uint16_t)block[2 * i + 1] << 16);
        maxv = (int32_t)ssub16_sel((uint32_t)maxv, data, (uint32_t)maxv, data);
        minv = (int32_t)ssub16_sel(data, (uint32_t)minv, (uint32_t)minv, data);
    }
    maxv = (int32_t)ssub16_sel((uint32_t)maxv, (uint32_t)(maxv >> 16), (uint32_t)maxv, (uint32_t)(maxv >> 16));   /* :470-471 */
    minv = (int32_t)ssub16_sel((uint32_t)(minv >> 16), (uint32_t)minv, (uint32_t)minv, (uint32_t)(minv >> 16));   /* :472-473 */
    minv = (int32_t)(minv < 0 ? 0u - (uint32_t)minv : (uint32_t)minv);      /* abs(), :475 */
    maxv = (int32_t)(maxv < 0 ? 0u - (uint32_t)maxv : (uint32_t)maxv);      /* :476 */
    uint16_t absmax = (uint16_t)ssub16_sel((uint32_t)maxv, (uint32_t)minv, (uint32_t)maxv, (uint32_t)minv);       /* :477-478 */

    --a->agc_idx;                                                    /* :480 */
    if (a->agc_idx >= 0) a->agc_buffer[a->agc_idx] = (int16_t)absmax;
    if (a->agc_idx < 0) a->agc_idx = ORC_AGCBUF_SIZE;                /* :481 */

    int m = 0;
    for (int i = 0; i < ORC_AGCBUF_SIZE; i++) m += a->agc_buffer[i];
    int d = m / ORC_AGCBUF_SIZE;                                     /* :485 */
    const float x = 16000;
    float f = x / (float)d;                                          /* :488 (d == 0 -> +inf) */
    float AGC_val = a->AGC_val;
    int set = 0;
    if ((double)f > 1.3) {
        float fagc = AGC_val + (AGC_val * f / 1500);                 /* :489 */
        if (fagc < 40.0f) { AGC_val = fagc; set = 1; }               /* :490-493, AGC_Max :95 */
    } else if ((double)AGC_val > 0.1) {
        if ((double)f < 0.6) { AGC_val = AGC_val - (AGC_val * f / 50); set = 1; }
        else if ((double)f < 0.7) { AGC_val = AGC_val - (AGC_val * f / 200); set = 1; }
        else if ((double)f < 0.8) { AGC_val = AGC_val - (AGC_val * f / 2000); set = 1; }
        else if ((double)f < 0.9) { AGC_val = AGC_val - (AGC_val * f / 4000); set = 1; }
    }
    if (set) { a->AGC_val = AGC_val; a->multiplier = orc_amp_multiplier(AGC_val); }
}

void orc_frontend_init(orc_frontend *f, uint16_t first_conversion)
{
    orc_dcblock_init(&f->dc, first_conversion);
    orc_agc_init(&f->agc);
}

void orc_frontend_run(orc_frontend *f, const uint16_t *adc, int16_t *out, uint32_t n_blocks)
{
    for (uint32_t b = 0; b < n_blocks; b++) {
        int16_t *o = out + (size_t)b * ORC_BLOCK;
        orc_dcblock_update(&f->dc, adc + (size_t)b * ORC_BLOCK, o, ORC_BLOCK);
        if (!orc_amp_update(f->agc.multiplier, o, ORC_BLOCK)) { memset(o, 0, ORC_BLOCK * sizeof(int16_t)); continue; }
        orc_agc_block(&f->agc, o);
    }
}

/* ======================================================================================
 * Row f2: SYNCAM PLL, Minimal-SDR.ino:631-688.  See msdr_oracle.h.
 * ====================================================================================== */
#define ORC_PI_ARDUINO 3.1415926535897932384626433832795   /* Arduino.h's PI (a double literal) */
#define ORC_SAMPLE_RATE 24000                              /* Minimal-SDR.ino:85 */

void orc_syncam_init(orc_syncam *s) { s->fil_out = 0.0f; s->omega2 = 0.0f; s->phzerror = 0.0f; }

void orc_syncam_constants(float c[4])
{
    static const float omegaN = 400.0;                                                     /* :637 */
    static const float zeta = 0.45;                                                        /* :638 */
    const float omega_min = 2.0 * ORC_PI_ARDUINO * -4000.0 / ORC_SAMPLE_RATE;              /* :639 */
    const float omega_max = 2.0 * ORC_PI_ARDUINO * 4000.0 / ORC_SAMPLE_RATE;               /* :640 */
    const float g1 = 1.0 - exp(-2.0 * omegaN * zeta / ORC_SAMPLE_RATE);                    /* :641 */
    const float g2 = -g1 + 2.0 * (1 - exp(-omegaN * zeta / ORC_SAMPLE_RATE) * cosf(omegaN / ORC_SAMPLE_RATE * sqrtf(1.0 - zeta * zeta)));   /* :642 */
    c[0] = omega_min; c[1] = omega_max; c[2] = g1; c[3] = g2;
}

void orc_syncam_q15(orc_syncam *s, const int16_t *I, const int16_t *Q, int16_t *out, uint32_t n)
{
    float c[4];
    orc_syncam_constants(c);
    const float omega_min = c[0], omega_max = c[1], g1 = c[2], g2 = c[3];
    float fil_out = s->fil_out, omega2 = s->omega2, phzerror = s->phzerror;
    for (uint32_t i = 0; i < n; i++) {
        const float Sin = (float)sin((double)phzerror);                 /* sinf, correctly rounded   :660 */
        const float Cos = (float)cos((double)phzerror);                 /* cosf                      :661 */
        const float ai = Cos * (float)I[i], bi = Sin * (float)I[i];     /* :662-663 */
        const float aq = Cos * (float)Q[i], bq = Sin * (float)Q[i];     /* :664-665 */
        const float corr0 = +ai + bq;                                   /* :667 */
        const float corr1 = -bi + aq;                                   /* :668 */
        out[i] = (int16_t)(uint16_t)(uint32_t)(int32_t)corr0;           /* :670 */
        const float det = (float)atan2((double)corr1, (double)corr0);   /* atan2f                    :674 */
        const float del_out = fil_out;                                  /* :677 */
        omega2 = omega2 + g2 * det;                                     /* :678 */
        if (omega2 < omega_min) omega2 = omega_min;                     /* :679 */
        else if (omega2 > omega_max) omega2 = omega_max;                /* :680 */
        fil_out = g1 * det + omega2;                                    /* :681 */
        phzerror = phzerror + del_out;                                  /* :682 */
        while ((double)phzerror >= 2 * ORC_PI_ARDUINO) phzerror = (float)((double)phzerror - 2.0 * ORC_PI_ARDUINO);   /* :685 */
        while ((double)phzerror < 0.0) phzerror = (float)((double)phzerror + 2.0 * ORC_PI_ARDUINO);                  /* :686 */
    }
    s->fil_out = fil_out; s->omega2 = omega2; s->phzerror = phzerror;
}

/* the same loop body on float samples (fp32 chain; msdr_oracle.h): out = corr[0], not truncated */
float orc_syncam_step_f32(orc_syncam *s, float I, float Q)
{
    float c[4];
    orc_syncam_constants(c);
    const float omega_min = c[0], omega_max = c[1], g1 = c[2], g2 = c[3];
    float fil_out = s->fil_out, omega2 = s->omega2, phzerror = s->phzerror;
    const float Sin = (float)sin((double)phzerror);                 /* :660 */
    const float Cos = (float)cos((double)phzerror);                 /* :661 */
    const float ai = Cos * I, bi = Sin * I;                         /* :662-663 */
    const float aq = Cos * Q, bq = Sin * Q;                         /* :664-665 */
    const float corr0 = +ai + bq;                                   /* :667 */
    const float corr1 = -bi + aq;                                   /* :668 */
    const float det = (float)atan2((double)corr1, (double)corr0);   /* :674 */
    const float del_out = fil_out;                                  /* :677 */
    omega2 = omega2 + g2 * det;                                     /* :678 */
    if (omega2 < omega_min) omega2 = omega_min;                     /* :679 */
    else if (omega2 > omega_max) omega2 = omega_max;                /* :680 */
    fil_out = g1 * det + omega2;                                    /* :681 */
    phzerror = phzerror + del_out;                                  /* :682 */
    while ((double)phzerror >= 2 * ORC_PI_ARDUINO) phzerror = (float)((double)phzerror - 2.0 * ORC_PI_ARDUINO);   /* :685 */
    while ((double)phzerror < 0.0) phzerror = (float)((double)phzerror + 2.0 * ORC_PI_ARDUINO);                  /* :686 */
    s->fil_out = fil_out; s->omega2 = omega2; s->phzerror = phzerror;
    return corr0;                                                   /* :670 without the int16 store */
}
void orc_syncam_f32(orc_syncam *s, const float *I, const float *Q, float *out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) out[i] = orc_syncam_step_f32(s, I[i], Q[i]);
}

/* ======================================================================================
 * Row f3: LMS automatic notch / noise reduction, Minimal-SDR.ino:702-770.  See msdr_oracle.h.
 * ====================================================================================== */
void orc_anr_init(orc_anr *a)
{
    memset(a, 0, sizeof *a);
    a->lidx = 120.0f;          /* :715 */
    a->ngamma = 0.001f;        /* :718 */
    a->in_idx = 0;             /* :723 */
}

void orc_anr_q15(orc_anr *a, int ANR_on, int16_t *p_dac, uint32_t n)
{
    if (!(ANR_on > 0)) return;                                    /* :702 */
    static const int ANR_taps = ORC_ANR_TAPS, ANR_delay = ORC_ANR_DELAY;
    static const float ANR_two_mu = 0.001, ANR_gamma = 0.1;
    static const float ANR_lidx_min = 0.0, ANR_lidx_max = 200.0;
    static const float ANR_den_mult = 6.25e-10, ANR_lincr = 1.0, ANR_ldecr = 3.0;
    const int ANR_mask = ORC_ANR_DLINE - 1;
    float ANR_lidx = a->lidx, ANR_ngamma = a->ngamma;
    int ANR_in_idx = a->in_idx;
    float *ANR_d = a->d, *ANR_w = a->w;
    for (uint32_t i = 0; i < n; i++) {
        int j, idx;
        float c0, c1, y, error, sigma, inv_sigp, nel, nev;
        ANR_d[ANR_in_idx] = p_dac[i];                             /* :734 */
        y = 0; sigma = 0;
        for (j = 0; j < ANR_taps; j++) {                          /* :739-744 */
            idx = (ANR_in_idx + j + ANR_delay) & ANR_mask;
            y += ANR_w[j] * ANR_d[idx];
            sigma += ANR_d[idx] * ANR_d[idx];
        }
        inv_sigp = 1.0 / (sigma + 1e-10);                         /* :745 */
        error = ANR_d[ANR_in_idx] - y;                            /* :746 */
        if (ANR_on == 1) p_dac[i] = (int16_t)(uint16_t)(uint32_t)(int32_t)error;     /* :749 notch filter */
        else p_dac[i] = (int16_t)(uint16_t)(uint32_t)(int32_t)y;                     /* :750 noise reduction */
        if ((nel = error * (1.0 - ANR_two_mu * sigma * inv_sigp)) < 0.0) nel = -nel;                                           /* :752 */
        if ((nev = ANR_d[ANR_in_idx] - (1.0 - ANR_two_mu * ANR_ngamma) * y - ANR_two_mu * error * sigma * inv_sigp) < 0.0) nev = -nev;   /* :753 */
        if (nev < nel) {                                          /* :754-757, as written */
            if ((ANR_lidx += ANR_lincr) > ANR_lidx_max) ANR_lidx = ANR_lidx_max;
            else if ((ANR_lidx -= ANR_ldecr) < ANR_lidx_min) ANR_lidx = ANR_lidx_min;
        }
        ANR_ngamma = ANR_gamma * (ANR_lidx * ANR_lidx) * (ANR_lidx * ANR_lidx) * ANR_den_mult;   /* :758 */
        c0 = 1.0 - ANR_two_mu * ANR_ngamma;                       /* :760 */
        c1 = ANR_two_mu * error * inv_sigp;                       /* :761 */
        for (j = 0; j < ANR_taps; j++) {                          /* :763-767 */
            idx = (ANR_in_idx + j + ANR_delay) & ANR_mask;
            ANR_w[j] = c0 * ANR_w[j] + c1 * ANR_d[idx];
        }
        ANR_in_idx = (ANR_in_idx + ANR_mask) & ANR_mask;          /* :768 */
    }
    a->lidx = ANR_lidx; a->ngamma = ANR_ngamma; a->in_idx = ANR_in_idx;
}

/* the same loop body on one float sample (fp32 chain; msdr_oracle.h): int16 units inside, nothing truncated */
float orc_anr_step_f32(orc_anr *a, int ANR_on, float sample)
{
    if (!(ANR_on > 0)) return sample;                             /* :702 */
    static const int ANR_taps = ORC_ANR_TAPS, ANR_delay = ORC_ANR_DELAY;
    static const float ANR_two_mu = 0.001, ANR_gamma = 0.1;
    static const float ANR_lidx_min = 0.0, ANR_lidx_max = 200.0;
    static const float ANR_den_mult = 6.25e-10, ANR_lincr = 1.0, ANR_ldecr = 3.0;
    const int ANR_mask = ORC_ANR_DLINE - 1;
    float *ANR_d = a->d, *ANR_w = a->w;
    const int ANR_in_idx = a->in_idx;
    int j, idx;
    float c0, c1, y, error, sigma, inv_sigp, nel, nev, out;
    ANR_d[ANR_in_idx] = sample * 32768.0f;                        /* :734, in the reference's sample units */
    y = 0; sigma = 0;
    for (j = 0; j < ANR_taps; j++) {                              /* :739-744 */
        idx = (ANR_in_idx + j + ANR_delay) & ANR_mask;
        y += ANR_w[j] * ANR_d[idx];
        sigma += ANR_d[idx] * ANR_d[idx];
    }
    inv_sigp = 1.0 / (sigma + 1e-10);                             /* :745 */
    error = ANR_d[ANR_in_idx] - y;                                /* :746 */
    out = (ANR_on == 1 ? error : y) * (1.0f / 32768.0f);          /* :749-750 without the int16 store */
    if ((nel = error * (1.0 - ANR_two_mu * sigma * inv_sigp)) < 0.0) nel = -nel;                                           /* :752 */
    if ((nev = ANR_d[ANR_in_idx] - (1.0 - ANR_two_mu * a->ngamma) * y - ANR_two_mu * error * sigma * inv_sigp) < 0.0) nev = -nev;   /* :753 */
    if (nev < nel) {                                              /* :754-757, as written */
        if ((a->lidx += ANR_lincr) > ANR_lidx_max) a->lidx = ANR_lidx_max;
        else if ((a->lidx -= ANR_ldecr) < ANR_lidx_min) a->lidx = ANR_lidx_min;
    }
    a->ngamma = ANR_gamma * (a->lidx * a->lidx) * (a->lidx * a->lidx) * ANR_den_mult;   /* :758 */
    c0 = 1.0 - ANR_two_mu * a->ngamma;                            /* :760 */
    c1 = ANR_two_mu * error * inv_sigp;                           /* :761 */
    for (j = 0; j < ANR_taps; j++) {                              /* :763-767 */
        idx = (ANR_in_idx + j + ANR_delay) & ANR_mask;
        ANR_w[j] = c0 * ANR_w[j] + c1 * ANR_d[idx];
    }
    a->in_idx = (ANR_in_idx + ANR_mask) & ANR_mask;               /* :768 */
    return out;
}
void orc_anr_f32(orc_anr *a, int anr_on, float *data, uint32_t n)
{
    if (!(anr_on > 0)) return;
    for (uint32_t i = 0; i < n; i++) data[i] = orc_anr_step_f32(a, anr_on, data[i]);
}

/* Row f4: output_dac.cpp:139-151 */
void orc_dac_format(const int16_t *src, int16_t *dest, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) dest[i] = src ? (int16_t)(((int32_t)src[i] + 32768) >> 4) : (int16_t)2048;
}

/* ======================================================================================
 * Row f4 (second half): the spectrum display's FFT, UI.cpp:520-592.
 *   initSpectrum()  : arm_rfft_init_q15(&FFT, 128, 0, 1)            (UI.cpp:522-524)
 *   showSpectrum()  : every 25th call (:534-536) arm_rfft_q15(&FFT, data, FFT_out) (:551),
 *                     then 127 column heights y_new = abs(FFT_out[127 - x]) / 200, clipped to
 *                     16, y1_new = 15 - y_new (:557-572).  The drawing calls are the OLED driver.
 * arm_rfft_q15 forward (arm_rfft_q15.c:74-112) = arm_cfft_q15(len 64) in place on the 128 reals
 * taken as 64 complex (re = even sample) + arm_split_rfft_q15.  arm_cfft_q15 for 64 points
 * (arm_cfft_q15.c:106-127) = arm_radix4_butterfly_q15(p, 64, twiddleCoef_64_q15, 1) +
 * arm_bitreversal_16 (ARM assembly, arm_bitreversal2.S:117-139: for each pair of table entries
 * (a, b) swap the 32-bit words at byte offsets a/2 and b/2).  The Cortex-M4 (ARM_MATH_DSP)
 * branches are the ones restated: arm_cfft_radix4_q15.c:156-611, arm_rfft_q15.c:152-222.
 * Every packed 2x16-bit intrinsic is written out per half with the semantics of the
 * reference header's generic-C definitions (arm_math.h:721-991).
 * PINNED: butterfly+bit reversal against the compiled arm_cfft_radix4_q15 (the reference's all-C
 * entry to the same butterfly), the split against the compiled arm_split_rfft_q15, the tables
 * against the compiled arm_common_tables.c / arm_rfft_init_q15.c (tests/test_oracle_pinned.py).
 * ====================================================================================== */
typedef struct { int32_t re, im; } cq15;   /* one packed word: re = low half, im = high half */
static inline int32_t top16(int32_t v) { return (int32_t)(int16_t)((uint32_t)v >> 16); }
static inline int32_t mul_wrap(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
/* re' = (c*re + s*im) >> 16, im' = (c*im - s*re) >> 16: __SMUAD(C, X) >> 16 and the high half of __SMUSDX(C, X) */
static inline cq15 twid(cq15 x, int32_t c, int32_t s)
{
    cq15 r;
    r.re = top16(wrap_add32(mul_wrap(c, x.re), mul_wrap(s, x.im)));
    r.im = top16((int32_t)((uint32_t)mul_wrap(c, x.im) - (uint32_t)mul_wrap(s, x.re)));
    return r;
}
static inline cq15 cq_ld(const int16_t *p, uint32_t i) { cq15 r = { p[2 * i], p[2 * i + 1] }; return r; }
static inline void cq_st(int16_t *p, uint32_t i, cq15 v) { p[2 * i] = (int16_t)v.re; p[2 * i + 1] = (int16_t)v.im; }
static inline cq15 cq_sar(cq15 a, int k) { cq15 r = { a.re >> k, a.im >> k }; return r; }
static inline cq15 cq_qadd(cq15 a, cq15 b) { cq15 r = { ssat16(a.re + b.re), ssat16(a.im + b.im) }; return r; }
static inline cq15 cq_qsub(cq15 a, cq15 b) { cq15 r = { ssat16(a.re - b.re), ssat16(a.im - b.im) }; return r; }

void orc_fft_tables(int16_t twiddle64[96], int16_t coefA[128], int16_t coefB[128])
{
    /* twiddleCoef_64_q15 (arm_common_tables.c:12914): {cos, sin}(2 pi i / 64), i < 48, floor(x * 2^15) clipped;
     * realCoefAQ15 / realCoefBQ15 (arm_rfft_init_q15.c:44-56, :1086-1098) at the 128-point stride 64:
     * A = {0.5 (1 - sin), -0.5 cos}, B = {0.5 (1 + sin), 0.5 cos} of 2 pi i / 128, round(x * 2^15). */
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < 48; i++) {
        double c = floor(cos(2.0 * pi * i / 64.0) * 32768.0 + 1e-6), s = floor(sin(2.0 * pi * i / 64.0) * 32768.0 + 1e-6);
        twiddle64[2 * i] = (int16_t)(c > 32767.0 ? 32767.0 : c);
        twiddle64[2 * i + 1] = (int16_t)(s > 32767.0 ? 32767.0 : s);
    }
    for (int i = 0; i < 64; i++) {
        double sn = sin(2.0 * pi * i / 128.0), cs = cos(2.0 * pi * i / 128.0);
        const double v[4] = { 0.5 * (1.0 - sn), -0.5 * cs, 0.5 * (1.0 + sn), 0.5 * cs };
        int16_t q[4];
        for (int k = 0; k < 4; k++) { double r = floor(v[k] * 32768.0 + 0.5); q[k] = (int16_t)(r > 32767.0 ? 32767.0 : r); }   /* B[64] = 0x7fff */
        coefA[2 * i] = q[0]; coefA[2 * i + 1] = q[1]; coefB[2 * i] = q[2]; coefB[2 * i + 1] = q[3];
    }
}

/* armBitRevIndexTable_fixed_64 (arm_common_tables.c:17129) as a set: 28 pairs of byte offsets 8*i, 8*rev6(i). */
uint32_t orc_bitrev_table64(uint16_t table[56])
{
    uint32_t n = 0;
    for (uint32_t i = 0; i < 64; i++) {
        uint32_t r = 0;
        for (int b = 0; b < 6; b++) r |= ((i >> b) & 1u) << (5 - b);
        if (i < r) { table[n++] = (uint16_t)(8 * i); table[n++] = (uint16_t)(8 * r); }
    }
    return n;
}
/* arm_bitreversal2.S:117-139 */
void orc_bitreversal_16(int16_t *buf, uint32_t bitRevLen, const uint16_t *table)
{
    uint32_t *w = (uint32_t *)buf;
    for (uint32_t k = 0; k < (bitRevLen + 1) / 2; k++) {
        uint32_t a = (table[2 * k] >> 1) / 4, b = (table[2 * k + 1] >> 1) / 4, t = w[a];
        w[a] = w[b]; w[b] = t;
    }
}

/* arm_cfft_radix4_q15.c:156-611 for fftLen = 64, twidCoefModifier = 1, in place, output in bit-reversed order */
void orc_radix4_butterfly64_q15(int16_t *p, const int16_t *tw)
{
    /* first stage (:168-365): inputs >> 2, 16 butterflies over (j, j+16, j+32, j+48), twiddle index j */
    for (uint32_t j = 0; j < 16; j++) {
        cq15 a = cq_sar(cq_ld(p, j), 2), b = cq_sar(cq_ld(p, j + 16), 2), c = cq_sar(cq_ld(p, j + 32), 2), d = cq_sar(cq_ld(p, j + 48), 2);
        cq15 R = cq_qadd(a, c), S = cq_qsub(a, c), T = cq_qadd(b, d), o, V;
        o.re = (R.re + T.re) >> 1; o.im = (R.im + T.im) >> 1;                /* __SHADD16(R, T) :239 */
        cq_st(p, j, o);
        R = cq_qsub(R, T);
        cq_st(p, j + 16, twid(R, tw[4 * j], tw[4 * j + 1]));                /* co2/si2 :246-275 */
        T = cq_qsub(b, d);
        V.re = ssat16(S.re - T.im); V.im = ssat16(S.im + T.re);             /* __QASX(S, T) :290 */
        S.re = ssat16(S.re + T.im); S.im = ssat16(S.im - T.re);             /* __QSAX(S, T) :292 */
        cq_st(p, j + 32, twid(S, tw[2 * j], tw[2 * j + 1]));                /* co1/si1 :300-323 */
        cq_st(p, j + 48, twid(V, tw[6 * j], tw[6 * j + 1]));                /* co3/si3 :326-349 */
    }
    /* middle stage (:371-517), once for 64 points: n1 = 16, n2 = 4, twiddle index 4 j */
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t ic = 4 * j;
        for (uint32_t i0 = j; i0 < 64; i0 += 16) {
            cq15 a = cq_ld(p, i0), b = cq_ld(p, i0 + 4), c = cq_ld(p, i0 + 8), d = cq_ld(p, i0 + 12);
            cq15 R = cq_qadd(a, c), S = cq_qsub(a, c), T = cq_qadd(b, d), o, V;
            o.re = ((R.re + T.re) >> 1) >> 1; o.im = ((R.im + T.im) >> 1) >> 1;   /* __SHADD16 twice :421-423 */
            cq_st(p, i0, o);
            R.re = (R.re - T.re) >> 1; R.im = (R.im - T.im) >> 1;           /* __SHSUB16 :427 */
            cq_st(p, i0 + 4, twid(R, tw[4 * ic], tw[4 * ic + 1]));
            T = cq_qsub(b, d);
            V.re = (S.re - T.im) >> 1; V.im = (S.im + T.re) >> 1;           /* __SHASX(S, T) :465 */
            S.re = (S.re + T.im) >> 1; S.im = (S.im - T.re) >> 1;           /* __SHSAX(S, T) :468 */
            cq_st(p, i0 + 8, twid(S, tw[2 * ic], tw[2 * ic + 1]));
            cq_st(p, i0 + 12, twid(V, tw[6 * ic], tw[6 * ic + 1]));
        }
    }
    /* last stage (:530-607): 16 groups of four neighbours, no twiddles; stores a', c', b', d' */
    for (uint32_t g = 0; g < 16; g++) {
        cq15 a = cq_ld(p, 4 * g), b = cq_ld(p, 4 * g + 1), c = cq_ld(p, 4 * g + 2), d = cq_ld(p, 4 * g + 3);
        cq15 R = cq_qadd(a, c), T = cq_qadd(b, d), S = cq_qsub(a, c), U = cq_qsub(b, d), o;
        o.re = (R.re + T.re) >> 1; o.im = (R.im + T.im) >> 1; cq_st(p, 4 * g, o);
        o.re = (R.re - T.re) >> 1; o.im = (R.im - T.im) >> 1; cq_st(p, 4 * g + 1, o);
        o.re = (S.re + U.im) >> 1; o.im = (S.im - U.re) >> 1; cq_st(p, 4 * g + 2, o);   /* __SHSAX(S, U) :585 */
        o.re = (S.re - U.im) >> 1; o.im = (S.im + U.re) >> 1; cq_st(p, 4 * g + 3, o);   /* __SHASX(S, U) :589 */
    }
}

/* arm_rfft_q15.c:128-222 (ARM_MATH_DSP branch), fftLen = 64 complex points, modifier applied by the caller's tables:
 * A/B = 64 {re, im} pairs (index i = the reference's pATable[2 * modifier * i]).  pDst gets 256 int16. */
void orc_split_rfft64_q15(const int16_t *X, const int16_t *A, const int16_t *B, int16_t *dst)
{
    for (uint32_t i = 1; i < 64; i++) {
        const int32_t r1 = X[2 * i], i1 = X[2 * i + 1], r2 = X[2 * (64 - i)], i2 = X[2 * (64 - i) + 1];
        const int32_t a0 = A[2 * i], a1 = A[2 * i + 1], b0 = B[2 * i], b1 = B[2 * i + 1];
        uint32_t outR = (uint32_t)mul_wrap(r1, a0) - (uint32_t)mul_wrap(i1, a1);                      /* __SMUSD :173 */
        outR = (outR + (uint32_t)mul_wrap(r2, b0) + (uint32_t)mul_wrap(i2, b1)) >> 16;                /* __SMLAD ... >> 16U :186 */
        uint32_t outI = (uint32_t)mul_wrap(r2, b1) - (uint32_t)mul_wrap(i2, b0);                      /* __SMUSDX :193 */
        outI = outI + (uint32_t)mul_wrap(r1, a1) + (uint32_t)mul_wrap(i1, a0);                        /* __SMLADX :202 */
        const int32_t im = (int32_t)outI >> 16;
        dst[2 * i] = (int16_t)outR; dst[2 * i + 1] = (int16_t)im;                                     /* :205-206 */
        dst[256 - 2 * i] = (int16_t)outR; dst[256 - 2 * i + 1] = (int16_t)(-im);                      /* :209-211 */
    }
    dst[128] = (int16_t)(((int32_t)X[0] - X[1]) >> 1); dst[129] = 0;                                  /* :218-219 */
    dst[0] = (int16_t)(((int32_t)X[0] + X[1]) >> 1); dst[1] = 0;                                      /* :221-222 */
}

/* arm_rfft_q15(&FFT, data, FFT_out) with FFT = (128, forward, bit reversal on).  `data` is left untouched here
 * (the reference transforms it in place, a latent bug: the mixer reads the block afterwards);
 * work[128] receives what the reference leaves in `data`. */
void orc_rfft128_q15(const int16_t *data, int16_t *fft_out /* 256 */, int16_t *work /* 128 or NULL */)
{
    int16_t tw[96], A[128], B[128], buf[128];
    uint16_t brt[56];
    orc_fft_tables(tw, A, B);
    const uint32_t nbr = orc_bitrev_table64(brt);
    memcpy(buf, data, sizeof buf);
    orc_radix4_butterfly64_q15(buf, tw);
    orc_bitreversal_16(buf, nbr, brt);
    orc_split_rfft64_q15(buf, A, B, fft_out);
    if (work) memcpy(work, buf, sizeof buf);
}

/* UI.cpp:557-572: y_new[x] = min(abs(FFT_out[127 - x]) / 200, 16) for x = 0..126 (int arithmetic: abs(-32768) = 32768) */
void orc_spectrum_columns(const int16_t *fft_out, uint8_t *y_new /* 127 */)
{
    for (int x = 0; x < 127; x++) {
        int v = fft_out[127 - x];
        int y = (v < 0 ? -v : v) / 200;
        y_new[x] = (uint8_t)(y > 16 ? 16 : y);
    }
}

/* showSpectrum's cadence (UI.cpp:534-536): `if (!Spectrum_on) return; if (--spectrumCounter > 0) return; spectrumCounter = 25;`
 * returns 1 when this call computes a spectrum. */
int orc_spectrum_tick(int spectrum_on, int *counter)
{
    if (!spectrum_on) return 0;
    if (--*counter > 0) return 0;
    *counter = 25;
    return 1;
}
