// msdr_design.cpp -- see msdr_design.h.  The results are truncated to integers, so operand types
// (float vs double) follow the reference expression by expression; compile with -ffp-contract=off.
#include "msdr_design.h"

#include <cmath>

namespace msdr {
namespace design {

namespace {
// `PI` comes from outside the sketch.  Two sources exist: the vendored CMSIS header's float fallback (src/CMSIS_5/arm_math.h:365-367:
// what the reference's files alone give) and, on the real target, Arduino.h's double literal, which wins there because it is
// defined first.  With the double, `m * PIH`, `PIH * (float)(ii - nc / 2)` and `PIH * (2 * jj - nc) * fc` are evaluated in double
// before they are rounded to float, and since the results are truncated to int16 single taps can differ by 1 LSB.  Both are
// built (P = float / double), both are checked against the compiled reference (oracle/build_ref.sh: calc_FIR_coeffs / _pid).
template <typename P> struct Pi;
template <> struct Pi<float>  { static constexpr float half = 3.14159265358979f / 2; };
template <> struct Pi<double> { static constexpr double half = 3.1415926535897932384626433832795 / 2; };

inline int16_t to_q15(float v) { return (int16_t)(int32_t)v; }
inline int16_t to_q15(double v) { return (int16_t)(int32_t)v; }

// Kaiser window value at normalised position x in [-1,1)  (.ino:852-853)
inline float kaiser(float beta, float x, float izb) { return izero(beta * sqrtf(1.0f - x * x)) / izb; }

// Kaiser beta from the stop-band attenuation (.ino:801-806); the literals are doubles in the sketch
float kaiser_beta(float astop)
{
    if (astop < 20.96) return 0.0f;
    if (astop >= 50.0) return (float)(0.1102 * (astop - 8.71));
    return (float)(0.5842 * powf((float)(astop - 20.96), (float)0.4) + 0.07886 * (astop - 20.96));
}
}  // namespace

// modified Bessel function I0 by its power series (.ino:883-899)
float izero(float x)
{
    const float errorlimit = 1e-9;
    const float half = (float)(x / 2.0);
    float sum = 1.0f, term = 1.0f, k = 1.0f;
    do {
        float q = half / k;
        q *= q;
        term *= q;
        sum += term;
        k = (float)(k + 1.0);
    } while (term >= errorlimit * sum);
    return sum;
}

// sin(m*pi/2*fc) / (m*pi/2*fc), m stepping by 2 (.ino:874-881)
template <typename P> static float sinc_half_pi_t(int m, float fc)
{
    if (m == 0) return 1.0f;
    const float x = (float)(m * Pi<P>::half);
    return sinf(x * fc) / (fc * x);
}
float sinc_half_pi(int m, float fc) { return sinc_half_pi_t<float>(m, fc); }

template <typename P> static void calc_fir_coeffs_t(int16_t *coeffs, int num_coeffs, float fc, float astop, int type, float dfc, float fsamp)
{
    fc = fc / fsamp;
    dfc = dfc / fsamp;
    const float beta = kaiser_beta(astop);
    const float izb = izero(beta);
    const int even = 2 * (num_coeffs / 2);
    float fcf;
    int nc;
    switch (type) {
    case 0: fcf = (float)(fc * 2.0); nc = num_coeffs; break;     // low-pass (.ino:812-815)
    case 1: fcf = -fc; nc = even; break;                          // high-pass
    case 2:
    case 3: fcf = dfc; nc = even; break;                          // band-pass / notch
    case 4: {                                                     // Hilbert (.ino:828-845), interleaved 2*nc layout
        nc = even;
        for (int i = 0; i < 2 * (nc - 1); i++) coeffs[i] = 0;
        coeffs[nc] = 1;
        for (int i = 1; i < nc + 1; i += 2) {
            if (2 * i == nc) continue;
            const float w = kaiser(beta, (float)(2 * i - nc) / (float)nc, izb);
            coeffs[2 * i + 1] = to_q15(32767 * (1.0f / (Pi<P>::half * (float)(i - nc / 2)) * w));
        }
        return;
    }
    default: return;
    }
    int j = 0;
    for (int i = -nc; i < nc; i += 2, j++) {                      // windowed sinc (.ino:850-855)
        const float w = kaiser(beta, (float)i / (float)nc, izb);
        coeffs[j] = to_q15(fcf * sinc_half_pi_t<P>(i, fcf) * w * 32767);
    }
    if (type == 1) {
        coeffs[nc / 2] += 1;
    } else if (type == 2 || type == 3) {                          // modulate to the centre frequency (.ino:861-869)
        const float gain = (type == 2) ? 2.0f : -2.0f;
        for (j = 0; j < nc + 1; j++) coeffs[j] = to_q15(coeffs[j] * (gain * cosf((float)(Pi<P>::half * (2 * j - nc) * fc))));
        if (type == 3) coeffs[nc / 2] += 1;
    }
}

void calc_fir_coeffs(int16_t *coeffs, int num_coeffs, float fc, float astop, int type, float dfc, float fsamp, bool pi_double)
{
    if (pi_double) calc_fir_coeffs_t<double>(coeffs, num_coeffs, fc, astop, type, dfc, fsamp);
    else calc_fir_coeffs_t<float>(coeffs, num_coeffs, fc, astop, type, dfc, fsamp);
}

void biquad_design(int kind, float frequency, float q_or_gain, float slope, double sample_rate, int32_t coef[5])
{
    const double w0 = frequency * (2 * 3.141592654 / sample_rate);
    const double sn = sin(w0), cs = cos(w0);
    if (kind <= 3) {                                              // low/high/band-pass, notch: h:56-111
        const double alpha = sn / ((double)q_or_gain * 2.0);
        const double scale = 1073741824.0 / (1.0 + alpha);
        double b0, b1, b2;
        if (kind == 0)      { b0 = ((1.0 - cs) / 2.0) * scale; b1 = (1.0 - cs) * scale;  b2 = NAN; }
        else if (kind == 1) { b0 = ((1.0 + cs) / 2.0) * scale; b1 = -(1.0 + cs) * scale; b2 = NAN; }
        else if (kind == 2) { b0 = alpha * scale;              b1 = 0.0;                 b2 = (-alpha) * scale; }
        else                { b0 = scale;                      b1 = (-2.0 * cs) * scale; b2 = NAN; }
        coef[0] = (int32_t)b0;
        coef[1] = (int32_t)b1;
        coef[2] = std::isnan(b2) ? coef[0] : (int32_t)b2;         // the header copies coef[0] (already truncated)
        coef[3] = (int32_t)((-2.0 * cs) * scale);
        coef[4] = (int32_t)((1.0 - alpha) * scale);
        return;
    }
    const double a = pow(10.0, q_or_gain / 40.0);                  // shelves: h:112-149
    const double sinsq = sn * sqrt((pow(a, 2.0) + 1.0) * (1.0 / slope - 1.0) + 2.0 * a);
    const double am = (a - 1.0) * cs, ap = (a + 1.0) * cs;
    if (kind == 4) {
        const double scale = 1073741824.0 / ((a + 1.0) + am + sinsq);
        coef[0] = (int32_t)(a * ((a + 1.0) - am + sinsq) * scale);
        coef[1] = (int32_t)(2.0 * a * ((a - 1.0) - ap) * scale);
        coef[2] = (int32_t)(a * ((a + 1.0) - am - sinsq) * scale);
        coef[3] = (int32_t)(-2.0 * ((a - 1.0) + ap) * scale);
        coef[4] = (int32_t)(((a + 1.0) + am - sinsq) * scale);
    } else {
        const double scale = 1073741824.0 / ((a + 1.0) - am + sinsq);
        coef[0] = (int32_t)(a * ((a + 1.0) + am + sinsq) * scale);
        coef[1] = (int32_t)(-2.0 * a * ((a - 1.0) + ap) * scale);
        coef[2] = (int32_t)(a * ((a + 1.0) + am - sinsq) * scale);
        coef[3] = (int32_t)(2.0 * ((a - 1.0) - ap) * scale);
        coef[4] = (int32_t)(((a + 1.0) - am - sinsq) * scale);
    }
}

// The three constant tables behind arm_rfft_q15 for 128 real points, regenerated from their documented formulas (the
// reference ships them as literals): twiddleCoef_64_q15 = {cos, sin}(2 pi i / 64), i < 48, floor(x * 2^15) clipped to Q15
// (arm_common_tables.c:12914); realCoefAQ15 / realCoefBQ15 = {0.5 (1 -/+ sin), -/+ 0.5 cos}(2 pi i / 128), i < 64,
// round(x * 2^15) clipped (arm_rfft_init_q15.c:44-56, :1086-1098, read at twidCoefRModifier = 64, :2206).  Checked
// entry by entry against the reference's literals (tests/golden: fft/*).
void fft128_tables(int16_t tables[352])
{
    const double pi = 3.14159265358979323846;
    auto q15 = [](double r) { return (int16_t)(r > 32767.0 ? 32767.0 : (r < -32768.0 ? -32768.0 : r)); };
    for (int i = 0; i < 48; i++) {
        tables[2 * i] = q15(floor(cos(2.0 * pi * i / 64.0) * 32768.0 + 1e-6));      // +1e-6: cos(pi/2) etc. are not exact zeros
        tables[2 * i + 1] = q15(floor(sin(2.0 * pi * i / 64.0) * 32768.0 + 1e-6));
    }
    int16_t *a = tables + 96, *b = tables + 96 + 128;
    for (int i = 0; i < 64; i++) {
        const double sn = sin(2.0 * pi * i / 128.0), cs = cos(2.0 * pi * i / 128.0);
        a[2 * i] = q15(floor(0.5 * (1.0 - sn) * 32768.0 + 0.5));
        a[2 * i + 1] = q15(floor(-0.5 * cs * 32768.0 + 0.5));
        b[2 * i] = q15(floor(0.5 * (1.0 + sn) * 32768.0 + 0.5));
        b[2 * i + 1] = q15(floor(0.5 * cs * 32768.0 + 0.5));
    }
}

}  // namespace design
}  // namespace msdr
