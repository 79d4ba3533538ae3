// msdr_design.h -- host-side filter designers of the demodulation chain (setup path, CPU arithmetic).
//   calc_fir_coeffs : Minimal-SDR.ino:782-872 (m_sinc :874-881, Izero :883-899)
//   biquad_design   : src/Audio/filter_biquad.h:56-149 (Audio-EQ-cookbook, coefficients * 2^30)
#pragma once
#include <stdint.h>

namespace msdr {
namespace design {

float izero(float x);
float sinc_half_pi(int m, float fc);
void calc_fir_coeffs(int16_t *coeffs, int num_coeffs, float fc, float astop, int type, float dfc, float fsamp);
void biquad_design(int kind, float frequency, float q_or_gain, float slope, double sample_rate, int32_t coef[5]);

}  // namespace design
}  // namespace msdr
