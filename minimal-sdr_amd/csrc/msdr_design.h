// msdr_design.h -- host-side filter designers of the demodulation chain (setup path, CPU arithmetic).
//   calc_fir_coeffs : Minimal-SDR.ino:782-872 (m_sinc :874-881, Izero :883-899)
//   biquad_design   : src/Audio/filter_biquad.h:56-149 (Audio-EQ-cookbook, coefficients * 2^30)
//   fft128_tables   : the constant tables arm_rfft_q15 uses for 128 points (arm_common_tables.c:12914, arm_rfft_init_q15.c:44-56, :1086-1098)
#pragma once
#include <stdint.h>

namespace msdr {
namespace design {

float izero(float x);
float sinc_half_pi(int m, float fc);
void calc_fir_coeffs(int16_t *coeffs, int num_coeffs, float fc, float astop, int type, float dfc, float fsamp, bool pi_double = false);
void fft128_tables(int16_t tables[352]);   // twiddleCoef_64_q15[96] | realCoefAQ15[::64 pairs][128] | realCoefBQ15[::64 pairs][128]
void biquad_design(int kind, float frequency, float q_or_gain, float slope, double sample_rate, int32_t coef[5]);

}  // namespace design
}  // namespace msdr
