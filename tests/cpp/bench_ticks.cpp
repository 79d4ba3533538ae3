// tests/cpp/bench_ticks.cpp -- the reference's tick through the reference's own operator API: AudioStream::update_all() once per
// AUDIO_BLOCK_SAMPLES = 128 samples (input_adc.cpp:122 -> every node's update() in construction order, Minimal-SDR.ino:66-74), on the
// MI355X runtime (minimal-sdr_amd/host).  Graph: source (an IF block batch resident in HBM, handed on by a device-to-device copy: what the
// ADC's DMA does on the Teensy) -> AudioSDRDemodulator (demodulation() + biquad1_dac + biquad2_dac as the reference writes them: Q15,
// bit-exact -- tests/cpp/test_graph.cpp checks that; this program only times) -> sink.
// Prints one JSON line: {"channels": C, "ticks": K, "tick_us": T, "Msamples_per_s": R, ...}.  usage: bench_ticks [channels] [ticks] [taps]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../minimal-sdr_amd/host/msdr_nodes.h"

q15_t Osc_Q_buffer_i[AUDIO_BLOCK_SAMPLES];
q15_t Osc_I_buffer_i[AUDIO_BLOCK_SAMPLES];

class ResidentSource : public AudioStream {
public:
    ResidentSource() : AudioStream(0, nullptr), d_if(nullptr) {}
    const int16_t *d_if;       // device [channels][AUDIO_BLOCK_SAMPLES]
    virtual void update(void)
    {
        if (!d_if) return;
        audio_block_t *b = allocate();
        if (!b) return;
        msdr_memcpy_d2d(AudioGPU.context(), b->data, d_if, AudioGPU.block_bytes());
        transmit(b);
        release(b);
    }
};
class NullSink : public AudioStream {
public:
    NullSink() : AudioStream(1, inputQueueArray), blocks(0) {}
    unsigned long blocks;
    std::vector<int16_t> first;     // the first block that arrives: channels 0 .. 3 (for the caller's parity check against the oracle)
    virtual void update(void)
    {
        audio_block_t *b = receiveReadOnly();
        if (!b) return;
        if (blocks == 0) {
            first.resize((size_t)4 * AUDIO_BLOCK_SAMPLES);
            msdr_memcpy_d2h(AudioGPU.context(), first.data(), b->data, first.size() * sizeof(int16_t));
        }
        blocks++;
        release(b);
    }
private:
    audio_block_t *inputQueueArray[1];
};

static ResidentSource adc;
static AudioSDRDemodulator demod;
static NullSink dac;
static AudioConnection patch1(adc, demod);
static AudioConnection patch2(demod, dac);

int main(int argc, char **argv)
{
    const uint32_t CH = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096;
    const int ticks = argc > 2 ? atoi(argv[2]) : 2000;
    const int taps = argc > 3 ? atoi(argv[3]) : 256;
    if (CH == 0 || CH > (1u << 20) || ticks <= 0 || ticks > 1000000 || taps < 2 || taps > 512 || (taps & 1)) { fprintf(stderr, "bad arguments\n"); return 2; }
    if (AudioGPU.begin(0, CH) != 0) { fprintf(stderr, "AudioGPU.begin failed: %s\n", msdr_last_error()); return 2; }
    if (AudioMemory(20) != 0) { fprintf(stderr, "AudioMemory failed: %s\n", msdr_last_error()); return 2; }
    msdr_ctx *ctx = AudioGPU.context();
    // filters as the sketch configures them: calc_demod_filter() (Minimal-SDR.ino:221-223; `taps` instead of 102), biquad1_dac / biquad2_dac (:391-393, :356)
    const double CORR_FACT = AUDIO_SAMPLE_RATE_EXACT / 24000.0;
    std::vector<int16_t> am(taps + 2);
    msdr_calc_FIR_coeffs(am.data(), taps, 2800, 70, 0, 0.0, 24000);
    int32_t lp[5], nt[5];
    msdr_biquad_design(MSDR_BQ_LOWPASS, (float)(6000 * 0.9 * CORR_FACT), 0.54f, 1.0f, AUDIO_SAMPLE_RATE_EXACT, lp);
    msdr_biquad_design(MSDR_BQ_NOTCH, (float)(24000 / 8 * CORR_FACT), 15.0f, 1.0f, AUDIO_SAMPLE_RATE_EXACT, nt);
    msdr_chain_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg; cfg.arith = MSDR_ARITH_Q15; cfg.channels = CH; cfg.mixer = MSDR_MIXER_FS4; cfg.default_mode = MSDR_MODE_AM;
    cfg.num_taps = (uint32_t)taps; cfg.num_tapsets = 1; cfg.coeffs_i[0] = am.data(); cfg.coeffs_q[0] = am.data();
    cfg.num_biquad_nodes = 2; cfg.node_stages[0] = 1; cfg.node_stages[1] = 1; cfg.node_coefs[0] = lp; cfg.node_coefs[1] = nt;
    if (demod.begin(cfg) != 0) { fprintf(stderr, "demod.begin failed: %s\n", msdr_last_error()); return 2; }

    std::vector<int16_t> h((size_t)CH * AUDIO_BLOCK_SAMPLES);
    unsigned s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (int16_t)((int)(s >> 16) % 16001 - 8000); }
    void *d_if = nullptr;
    if (msdr_malloc(ctx, h.size() * 2, &d_if) != 0 || msdr_memcpy_h2d(ctx, d_if, h.data(), h.size() * 2) != 0) { fprintf(stderr, "IF buffer: %s\n", msdr_last_error()); return 2; }
    adc.d_if = (const int16_t *)d_if;

    for (int k = 0; k < 200; k++) AudioStream::update_all();
    AudioGPU.synchronize();
    const unsigned long before = dac.blocks;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < ticks; k++) AudioStream::update_all();
    AudioGPU.synchronize();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const unsigned long got = dac.blocks - before;
    msdr_chain_info info;
    memset(&info, 0, sizeof info);
    printf("{\"first_block_ch0_3\": [");
    for (size_t k = 0; k < dac.first.size(); k++) printf("%s%d", k ? "," : "", (int)dac.first[k]);
    printf("], \"channels\": %u, \"ticks\": %d, \"blocks_through\": %lu, \"taps\": %d, \"tick_us\": %.3f, \"Msamples_per_s\": %.1f, "
           "\"graph\": \"ResidentSource (d2d copy) -> AudioSDRDemodulator (Q15 demodulation() + biquad1_dac + biquad2_dac) -> sink, AudioStream::update_all() per 128 samples\"}\n",
           CH, ticks, got, taps, dt / ticks * 1e6, (double)CH * AUDIO_BLOCK_SAMPLES * ticks / dt / 1e6);
    msdr_free(ctx, d_if);
    return got == (unsigned long)ticks ? 0 : 1;
}
