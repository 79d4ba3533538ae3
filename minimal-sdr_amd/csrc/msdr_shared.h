// msdr_shared.h -- definitions shared by the host C-ABI layer and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace msdr {

constexpr int kThreads   = 256;                 // one workgroup = 4 wave64, one per SIMD of a CU
constexpr int kWaves     = kThreads / 64;
// Outputs per lane in the fused chain.  The LDS window holds {I,Q} pairs (8 B); a lane's window
// starts R pairs = R/2 16-byte slots after its neighbour's.  R/2 must be ODD so that the 16 lanes
// ds_read_b128 services per LDS cycle hit 16 different 16-byte slots (no bank conflict, no padding).
constexpr int kChainR    = 10;
constexpr int kChainTile = kThreads * kChainR;  // 2560 output samples per workgroup tile
// Single-stream FIR (arm_fir_f32 / arm_fir_fast_q15 mirrors): 4-byte elements, lane stride R/4 slots.
constexpr int kFirR      = 12;                  // 3 slots: odd
constexpr int kFirTile   = kThreads * kFirR;    // 3072
// Folded fp32 chain: one float per sample in LDS, 12 outputs per lane (3 slots: odd), and 12 is a
// multiple of every supported NCO period (1, 2, 4), so a lane's r-th output always has phase r mod P.
constexpr int kFoldR     = 12;
constexpr int kFoldTile  = kThreads * kFoldR;   // 3072
// Biquad-only kernel (arm_biquad_cascade_df1_f32 mirror)
constexpr int kBqR       = 8;
constexpr int kBqTile    = kThreads * kBqR;     // 2048

constexpr int kMaxStages = 4;

// Host-precomputed tables for the in-workgroup parallel evaluation of one df1 biquad stage whose
// lanes each own L consecutive samples (see DESIGN.md "IIR along time").
//   y[n] = u[n] + a1 y[n-1] + a2 y[n-2],  u[n] = b0 x[n] + b1 x[n-1] + b2 x[n-2]
//   homogeneous solution from (p,q) = (y[-1], y[-2]):  h[j] = alpha[j] p + beta[j] q
//   lane transition M = [[alpha[L-1], beta[L-1]], [alpha[L-2], beta[L-2]]]
template <int L>
struct BiquadStageTables {
    float b0, b1, b2, a1, a2;
    float pad0[3];
    float alpha[L], beta[L];
    float mpow[6][4];     // M^(2^d), d = 0..5, row-major 2x2
    float m64[4];         // M^64  (one whole wave)
    float mlane[64][4];   // M^(l+1), l = lane in wave
};

struct ChainParams {
    const int16_t *x;          // [channels][n] IF samples
    void *out;                 // [channels][n] float (F32) or int16 (Q15)
    const int16_t *hist_in;    // [channels][hist_len] raw IF history, oldest first
    int16_t *hist_out;         // written by the state kernel
    long long n;               // samples per channel in this call
    int channels;
    int nseg;                  // time segments per channel
    long long seg_len;         // multiple of the tile
    int warm;                  // IIR warm-up samples for segments > 0 (multiple of the tile)
    int ntaps_pad;             // taps per filter, front-padded with zeros to a multiple of 4
    int hist_len;              // = ntaps_pad - 1
    const void *taps;          // [tapsets][ntaps_pad] pairs {hI[k], hQ[k]}: float2 (F32) / int2 (Q15)
    const int *chan_mode;      // [channels]
    const int *chan_tapset;    // [channels]
    int mixer;                 // MSDR_MIXER_*
    const void *osc;           // [osc_len] pairs {osc_q ("cos"), osc_i ("sin")}: float2 / int2
    int osc_len;
    int phase0;                // (absolute index of sample 0 of this call) mod osc_len (mod 4 for FS4)
    float in_scale;
    int sqrt_kind;
    int nstages;               // F32 biquad stages
    const BiquadStageTables<kChainR> *bq;   // [nstages]
    float *bq_state;           // [channels][kMaxStages][4] = x1,x2,y1,y2
    // folded F32 kernel (mixer folded into the taps; see DESIGN.md "Tap folding")
    const float *ftaps;        // [fsets][P rotations][steps][PE rows][2][2] folded tap pairs, in_scale included
    const int *chan_fset;      // [channels] folded-set index
    int fold_period;           // P: NCO period in samples (1, 2 or 4)
    int fold_rot;              // (absolute index of sample 0 of this call) mod P
    const BiquadStageTables<kFoldR> *bq_fold;   // [nstages], lanes own kFoldR samples
};

}  // namespace msdr
