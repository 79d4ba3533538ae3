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

// Host-precomputed tables for the in-workgroup parallel evaluation of a df1 biquad CASCADE whose lanes
// each own L consecutive samples (DESIGN.md "IIR along time").  The cascade
//     H(z) = prod_s (b0 + b1 z^-1 + b2 z^-2)_s / (1 - a1 z^-1 - a2 z^-2)_s
// is evaluated as ONE numerator FIR  v = C(z) d,  C = prod_s B_s  (2S+1 taps, in registers), followed by
// S all-pole sections  w_s[n] = w_(s-1)[n] + a1 w_s[n-1] + a2 w_s[n-2].  For an all-pole section:
//   end state of a lane run from zero state   Z = sum_j (g[L-1-j], g[L-2-j]) v_j      (g = impulse response)
//   lane transition                           (y[L-1], y[L-2]) = M (p, q) + Z
template <int L>
struct BiquadCascadeTables {
    int nstages;
    int pad0[3];
    float num[2 * kMaxStages + 1 + 3];   // c_0 .. c_2S, zero padded to 12
    struct Section {
        float a1, a2;
        float pad1[2];
        float g[L][2];        // g[j] = (g_(L-1-j), g_(L-2-j))
        float mcol[4][2][2];  // M^(2^k), k = 0..3, stored by columns: [k][col][row]
        float m64[4];         // M^64 row-major 2x2 (one whole wave)
        float mlane[64][4];   // M^(l+1) stored {M00, M10, M01, M11} (columns), l = lane in wave
    } sec[kMaxStages];
};
// per-channel cascade state in HBM: 16 floats = d[n-1..n-8] (numerator history) | (w1, w2) per section
constexpr int kBqStateFloats = 16;

// Oscillator tables that were in force before a live change, oldest first (device memory, behind ONE pointer of ChainParams: the
// matrix-core kernels are at their register budget, every kernel argument costs them scalar registers).  A carried history sample at
// time t < sw[k] (t relative to the call, so sw <= 0) was mixed with table k when it arrived (freq_conv.cpp:70-103 mixes each block with
// the tables as they are at that update()), and chain_kernel<Arith> mixes it with that table again.
constexpr int kOscHistMax = 16;      // oscillator tables that can still have samples in one FIR history (msdr_chain_set_osc)
struct OscHistory { const void *tab[kOscHistMax]; long long sw[kOscHistMax]; int n; };

constexpr int kChainOutI16 = 0x40000000;

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
    const BiquadCascadeTables<kChainR> *bq;
    float *bq_state;           // [channels][kBqStateFloats]
    short *syncam_q;           // Q15, MSDR_CHAIN_SYNCAM_PLL: [channels][n] post-FIR Q of SYNCAM channels (their I goes to `out`); else null
    // folded F32 kernel (mixer folded into the taps; see DESIGN.md "Tap folding")
    const float *ftaps;        // [fsets][P rotations][steps][PE rows][2][2] folded tap pairs, in_scale included
    const int *chan_fset;      // [channels] folded-set index
    int fold_period;           // P: NCO period in samples (1, 2 or 4)
    int fold_rot;              // (absolute index of sample 0 of this call) mod P
    const BiquadCascadeTables<kFoldR> *bq_fold;   // lanes own kFoldR samples
    // matrix-core kernel (msdr_chain_mfma.hiph); uses chan_fset, fold_period, fold_rot
    const void *mf_tab;        // [fsets][P rotations] tables: MfmaTableHeader (1 KB) + B fragments, mf_stride bytes apart
    int mf_stride;
    int mf_halo;               // H: window samples before the tile, a multiple of 32 >= ntaps - 1
    int mf_bsteps;             // k-steps the LDS B region holds (max n0 + n1 over the tables)
    const void *bq_mf;         // BiquadCascadeTables<16>: 16-sample chunks
    const void *bq_mf32;       // BiquadCascadeTables<32>: 32-sample blocks (msdr_biquad_paired.hiph)
    int mf_waves;              // waves per workgroup: 4 or 8
    // wave-stream matrix-core kernel (msdr_chain_mfw.hiph)
    const int *mf_units;       // [workgroups][mf_nw] pairs (channel, segment); channel < 0 = idle wave
    int mf_nw;                 // waves per workgroup (1..16)
    unsigned long long *dbg_buf;   // diagnostic build (-DMSDR_STAMPS): per-phase cycle sums, 8 per unit
    int dbg;                   // diagnostic bits: read from MSDR_DBG by the -DMSDR_STAMPS build only; the product build looks at ONE bit of it:
                               // kChainOutI16 -- chain_mfw_kernel / chain_amtr_kernel, F32: `out` is int16 [channels][n], written as arm_float_to_q15
                               // converts (MSDR_CHAIN_OUT_I16).  (A bit of an existing field: these kernels are at their register budget and
                               // every kernel argument costs them scalar registers -- a field of its own made the headline flavour spill.)
    const float *mw_iir;       // wave-stream kernel, folded IIR: MwIirConsts block (scan matrices, response fragments), or null
    float *bq_state_out;       // [channels][kBqStateFloats] cascade state after this call (ping-pong partner of bq_state)
    // chain_kernel<Arith> only: oscillator tables that were in force before a live change (msdr_chain_set_osc), or null -- see OscHistory
    const struct OscHistory *osc_hist;
};

}  // namespace msdr
