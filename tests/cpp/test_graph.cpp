// tests/cpp/test_graph.cpp -- the reference's operator graph, rebuilt on the MI355X runtime, checked
// against the oracle (TEST code: links oracle/liboracle.so; the product libraries never do).
//
//   graph A (as the sketch wires it, Minimal-SDR.ino:66-81):
//       source -> queue_adc            queue_dac -> biquad1_dac -> biquad2_dac -> capture
//     and between the queues a demodulation() written like the reference's (:518-775) with the
//     kernel-function API: Fs/4 mix -> arm_fir_fast_q15 x2 -> demod switch.
//   graph B: source -> AudioSDRDemodulator (fused kernel, biquads inside) -> capture
//   graph C: two sources -> AudioEffectFreqConv -> two captures
//   graph D: source -> AudioAmplifier (amp_adc / amp_dac, mixer.cpp:134-159) -> capture
// Exit code 0 = every sample bit-exact against the oracle.  `--no-gpu` checks the failure path only.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../minimal-sdr_amd/host/msdr_nodes.h"
#include "../../oracle/msdr_oracle.h"

q15_t Osc_Q_buffer_i[AUDIO_BLOCK_SAMPLES];
q15_t Osc_I_buffer_i[AUDIO_BLOCK_SAMPLES];

static const uint32_t CH = 5;
static const int B = AUDIO_BLOCK_SAMPLES;

// a source node the test feeds with host data (stands in for adc1 + amp_adc)
class TestSource : public AudioStream {
public:
    TestSource() : AudioStream(0, nullptr), next(nullptr) {}
    const int16_t *next;    // host [CH][B]
    virtual void update(void)
    {
        if (!next) return;
        audio_block_t *b = allocate();
        if (!b) return;
        msdr_memcpy_h2d(AudioGPU.context(), b->data, next, AudioGPU.block_bytes());
        transmit(b);
        release(b);
        next = nullptr;
    }
};

// ---- graph A ---------------------------------------------------------------------------------
static TestSource adc1;
static AudioRecordQueue queue_adc;
static AudioPlayQueue queue_dac;
static AudioFilterBiquad biquad1_dac;
static AudioFilterBiquad biquad2_dac;
static AudioRecordQueue capture_a;
static AudioConnection patchCord1(adc1, queue_adc);
static AudioConnection patchCord2(queue_dac, biquad1_dac);
static AudioConnection patchCord4a(biquad1_dac, biquad2_dac);
static AudioConnection patchCord4(biquad2_dac, capture_a);
// ---- graph B ---------------------------------------------------------------------------------
static TestSource adc_b;
static AudioSDRDemodulator demod_b;
static AudioRecordQueue capture_b;
static AudioConnection patchB1(adc_b, demod_b);
static AudioConnection patchB2(demod_b, capture_b);
// ---- graph C ---------------------------------------------------------------------------------
static TestSource src_i, src_q;
static AudioEffectFreqConv freqconv;
static AudioRecordQueue cap_i, cap_q;
static AudioConnection patchC1(src_i, 0, freqconv, 0);
static AudioConnection patchC2(src_q, 0, freqconv, 1);
static AudioConnection patchC3(freqconv, 0, cap_i, 0);
static AudioConnection patchC4(freqconv, 1, cap_q, 0);

// ---- graph D ---------------------------------------------------------------------------------
static TestSource src_d;
static AudioAmplifier amp_d;
static AudioRecordQueue cap_d;
static AudioConnection patchD1(src_d, amp_d);
static AudioConnection patchD2(amp_d, cap_d);

static int mode = ORC_AM;
static msdr_fir_q15 *FIR_I, *FIR_Q;
static int16_t *d_I, *d_Q, *d_If, *d_Qf;

// the reference's demodulation() (Minimal-SDR.ino:518-775) with the batched kernel functions
static unsigned long demodulation(void)
{
    if (queue_dac.available() == false) return 0;
    if (queue_adc.available() < 1) return 0;
    msdr_ctx *ctx = AudioGPU.context();
    int16_t *p_adc = queue_adc.readBuffer();
    msdr_mix_fs4_q15(ctx, p_adc, d_I, d_Q, CH, B);                       // :546-558
    queue_adc.freeBuffer();                                              // :560
    msdr_fir_q15_process(FIR_I, d_I, d_If, B);                           // :574
    msdr_fir_q15_process(FIR_Q, d_Q, d_Qf, B);                           // :575
    int16_t *p_dac = queue_dac.getBuffer();                              // :587
    msdr_demod_q15(ctx, mode, nullptr, MSDR_SQRT_F32, d_If, d_Qf, p_dac, CH, B);   // :589-627
    queue_dac.playBuffer();                                              // :773
    return 1;
}

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fails++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static void fetch(AudioRecordQueue &q, std::vector<int16_t> &out)
{
    int16_t *d = q.readBuffer();
    out.assign((size_t)CH * B, 0);
    if (d) msdr_memcpy_d2h(AudioGPU.context(), out.data(), d, AudioGPU.block_bytes());
    else CHECK(false, "capture queue empty");
    q.freeBuffer();
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "--no-gpu")) {
        {   // the queue nodes' ring (host logic only): order, capacity, wrap-around, full / empty
            BlockRing<3> ring;
            audio_block_t blk[5];
            CHECK(ring.size() == 0 && ring.pop() == nullptr, "a new ring is empty");
            CHECK(ring.push(&blk[0]) && ring.push(&blk[1]) && ring.push(&blk[2]), "three blocks fit");
            CHECK(!ring.push(&blk[3]) && ring.size() == 3, "the fourth is refused and nothing changes");
            CHECK(ring.pop() == &blk[0] && ring.pop() == &blk[1] && ring.size() == 1, "first in, first out");
            for (int round = 0; round < 1000; round++) {                     // many laps around the three slots
                CHECK(ring.push(&blk[round % 5]), "push lap %d", round);
                CHECK(ring.pop() == &blk[round ? (round - 1) % 5 : 2], "pop lap %d", round);
            }
            CHECK(ring.size() == 1 && ring.pop() == &blk[999 % 5] && ring.pop() == nullptr, "drained");
        }
        int rc = AudioGPU.begin(0, CH);
        if (msdr_device_count() == 0) {
            CHECK(rc == MSDR_STATUS_NO_DEVICE, "begin() without a GPU returned %d", rc);
            CHECK(AudioMemory(20) != 0, "AudioMemory must fail without a context");
            // without a pool the queues hand out nothing and accept nothing, and never spin
            AudioPlayQueue pq;
            AudioRecordQueue rq;
            CHECK(!pq.available() && pq.getBuffer() == nullptr && !pq.playBuffer(), "play queue without a pool");
            rq.begin();
            CHECK(rq.available() == 0 && rq.readBuffer() == nullptr, "record queue without data");
            rq.freeBuffer(); rq.clear(); rq.end();
            printf("no-gpu path: %s (%s)\n", fails ? "FAILED" : "OK", msdr_last_error());
        }
        return fails ? 1 : 0;
    }
    if (AudioGPU.begin(0, CH) != 0) { printf("AudioGPU.begin failed: %s\n", msdr_last_error()); return 2; }
    CHECK(AudioMemory(20) == 0, "AudioMemory");                          // Minimal-SDR.ino:83,373
    msdr_ctx *ctx = AudioGPU.context();

    // setup(): filters as the sketch configures them (.ino:391-393, 356, 115, 222)
    const double CORR_FACT = AUDIO_SAMPLE_RATE_EXACT / 24000.0;
    biquad1_dac.setLowpass(0, 6000 * 0.9 * CORR_FACT, 0.54);
    biquad2_dac.setNotch(0, 24000 / 8 * CORR_FACT, 15.0);
    int16_t FIR_AM_coeffs[102];
    msdr_calc_FIR_coeffs(FIR_AM_coeffs, 102, 2800, 70, 0, 0.0, 24000);
    CHECK(msdr_fir_q15_create(ctx, 102, FIR_AM_coeffs, CH, &FIR_I) == 0, "FIR_I");
    CHECK(msdr_fir_q15_create(ctx, 102, FIR_AM_coeffs, CH, &FIR_Q) == 0, "FIR_Q");
    for (int16_t **p : {&d_I, &d_Q, &d_If, &d_Qf}) msdr_malloc(ctx, AudioGPU.block_bytes(), (void **)p);
    queue_adc.begin(); capture_a.begin(); capture_b.begin(); cap_i.begin(); cap_q.begin(); cap_d.begin();

    // fused node: same chain, biquads inside
    int32_t lp[5], nt[5];
    msdr_biquad_design(MSDR_BQ_LOWPASS, (float)(6000 * 0.9 * CORR_FACT), 0.54f, 1.0f, AUDIO_SAMPLE_RATE_EXACT, lp);
    msdr_biquad_design(MSDR_BQ_NOTCH, (float)(24000 / 8 * CORR_FACT), 15.0f, 1.0f, AUDIO_SAMPLE_RATE_EXACT, nt);
    msdr_chain_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg; cfg.arith = MSDR_ARITH_Q15; cfg.channels = CH; cfg.mixer = MSDR_MIXER_FS4;
    cfg.num_taps = 102; cfg.num_tapsets = 1; cfg.coeffs_i[0] = FIR_AM_coeffs; cfg.coeffs_q[0] = FIR_AM_coeffs;
    cfg.default_mode = MSDR_MODE_AM; cfg.num_biquad_nodes = 2; cfg.node_stages[0] = 1; cfg.node_stages[1] = 1;
    cfg.node_coefs[0] = lp; cfg.node_coefs[1] = nt;
    CHECK(demod_b.begin(cfg) == 0, "demod_b.begin: %s", msdr_last_error());

    // oracle state per channel
    std::vector<orc_chain_q15_state> st(CH);
    std::vector<std::vector<int16_t>> si(CH, std::vector<int16_t>(102 + B, 0)), sq(CH, std::vector<int16_t>(102 + B, 0));
    for (uint32_t c = 0; c < CH; c++) {
        st[c].state_i = si[c].data(); st[c].state_q = sq[c].data();
        for (int k = 0; k < 2; k++) orc_biquad_teensy_init(&st[c].bq[k]);
        orc_biquad_teensy_set_coefficients(&st[c].bq[0], 0, lp);
        orc_biquad_teensy_set_coefficients(&st[c].bq[1], 0, nt);
    }
    orc_chain_q15_cfg ocfg;
    memset(&ocfg, 0, sizeof ocfg);
    ocfg.mode = ORC_AM; ocfg.sqrt_kind = ORC_SQRT_F32; ocfg.mixer = 0; ocfg.num_taps = 102;
    ocfg.coeffs_i = FIR_AM_coeffs; ocfg.coeffs_q = FIR_AM_coeffs; ocfg.n_biquad_nodes = 2;

    srand(7);
    std::vector<int16_t> x((size_t)CH * B), want((size_t)CH * B), got;
    for (int tick = 0; tick < 12; tick++) {
        for (auto &v : x) v = (int16_t)((rand() % 24001) - 12000);
        if (tick == 5) for (auto &v : x) v = (int16_t)((rand() & 1) ? 32767 : -32768);   // full scale
        if (tick == 6) {
            // the bandwidth menu: calc_demod_filter() rewrites FIR_AM_coeffs IN PLACE, no init_FIR() (UI.cpp:332-345, Minimal-SDR.ino:221-223);
            // the oracle's instances hold the pointer, like the reference's (arm_fir_init_q15.c:100-109)
            msdr_calc_FIR_coeffs(FIR_AM_coeffs, 102, 2025, 70, 0, 0.0, 24000);
            CHECK(msdr_fir_q15_set_coeffs(FIR_I, FIR_AM_coeffs) == 0 && msdr_fir_q15_set_coeffs(FIR_Q, FIR_AM_coeffs) == 0, "set_coeffs: %s", msdr_last_error());
            CHECK(demod_b.setTaps(0, FIR_AM_coeffs, FIR_AM_coeffs) == 0, "demod_b.setTaps: %s", msdr_last_error());
        }
        if (tick == 9) {
            // tune(): biquad2_dac.setNotch(0, pdb_freq_actual / 8.0 * CORR_FACT, 15.0) on the running graph (Minimal-SDR.ino:356)
            const float f = (float)(23200.0 / 8 * CORR_FACT);
            biquad2_dac.setNotch(0, f, 15.0);
            CHECK(demod_b.setNodeNotch(1, 0, f, 15.0f) == 0, "demod_b.setNodeNotch: %s", msdr_last_error());
            int32_t nt2[5];
            msdr_biquad_design(MSDR_BQ_NOTCH, f, 15.0f, 1.0f, AUDIO_SAMPLE_RATE_EXACT, nt2);
            for (uint32_t c = 0; c < CH; c++) orc_biquad_teensy_set_coefficients(&st[c].bq[1], 0, nt2);
        }
        adc1.next = x.data(); adc_b.next = x.data();
        AudioStream::update_all();           // tick 1: source -> queue_adc, fused node runs
        demodulation();                      // main loop
        AudioStream::update_all();           // tick 2: queue_dac -> biquads -> capture
        for (uint32_t c = 0; c < CH; c++)
            orc_chain_q15(&ocfg, &st[c], x.data() + (size_t)c * B, want.data() + (size_t)c * B, nullptr, nullptr, 1);
        fetch(capture_a, got);
        CHECK(got == want, "graph A (queues + demodulation() + biquad nodes) differs from the oracle at tick %d", tick);
        fetch(capture_b, got);
        CHECK(got == want, "graph B (fused AudioSDRDemodulator) differs from the oracle at tick %d", tick);
    }
    CHECK(AudioMemoryUsageMax() > 0 && AudioMemoryUsageMax() <= 20, "AudioMemoryUsageMax = %d", (int)AudioMemoryUsageMax());
    CHECK(AudioMemoryUsage() == 0, "blocks leaked: %d still in use", (int)AudioMemoryUsage());
    CHECK(AudioProcessorUsageMax() >= 0.0f, "usage");

    // ---- graph C: freq_conv node, both directions, bypass, and a missing input -------------------------------
    for (int i = 0; i < B; i++) {
        Osc_I_buffer_i[i] = (q15_t)(32767.0 * __builtin_sin(2 * 3.14159265358979 * 5 * i / B));
        Osc_Q_buffer_i[i] = (q15_t)(32767.0 * __builtin_cos(2 * 3.14159265358979 * 5 * i / B));
    }
    std::vector<int16_t> xi((size_t)CH * B), xq((size_t)CH * B), gi, gq;
    for (int variant = 0; variant < 4; variant++) {
        const bool dir = variant & 1, pass = !(variant & 2);
        freqconv.direction(dir); freqconv.passthrough(pass);
        for (auto &v : xi) v = (int16_t)((rand() % 65536) - 32768);
        for (auto &v : xq) v = (int16_t)((rand() % 65536) - 32768);
        src_i.next = xi.data(); src_q.next = xq.data();
        AudioStream::update_all();
        fetch(cap_i, gi); fetch(cap_q, gq);
        for (uint32_t c = 0; c < CH; c++)
            orc_freqconv_q15(xi.data() + (size_t)c * B, xq.data() + (size_t)c * B, Osc_I_buffer_i, Osc_Q_buffer_i, dir, pass, B);
        CHECK(gi == xi && gq == xq, "graph C (AudioEffectFreqConv dir=%d pass=%d) differs from the oracle", (int)dir, (int)pass);
    }
    src_i.next = xi.data(); src_q.next = nullptr;       // only one input arrives: nothing is transmitted (freq_conv.cpp:40-47)
    AudioStream::update_all();
    CHECK(cap_i.available() == 0 && cap_q.available() == 0, "freq_conv transmitted with a missing input");
    CHECK(AudioMemoryUsage() == 0, "blocks leaked after graph C: %d", (int)AudioMemoryUsage());

    // ---- graph D: AudioAmplifier: gain, unity pass-through, zero gain transmits nothing (mixer.cpp:139-157) ----
    for (float g : {0.37f, 1.0f, 3.0f, -1.0f, 0.0f}) {
        amp_d.gain(g);
        for (auto &v : xi) v = (int16_t)((rand() % 65536) - 32768);
        src_d.next = xi.data();
        AudioStream::update_all();
        std::vector<int16_t> w = xi;
        const int sent = orc_amp_update(orc_amp_multiplier(g), w.data(), (uint32_t)w.size());
        if (!sent) CHECK(cap_d.available() == 0, "AudioAmplifier with zero gain transmitted a block");
        else { fetch(cap_d, gi); CHECK(gi == w, "graph D (AudioAmplifier gain %g) differs from the oracle", (double)g); }
    }
    CHECK(AudioMemoryUsage() == 0, "blocks leaked after graph D: %d", (int)AudioMemoryUsage());

    // pool exhaustion: a node that cannot allocate drops the tick's data, nothing crashes (record_queue.cpp:91-92)
    AudioMemory(1);
    adc1.next = x.data(); adc_b.next = x.data();
    AudioStream::update_all();
    CHECK(AudioMemoryUsageMax() <= 1, "pool of 1 exceeded");

    AudioGPU.synchronize();
    msdr_fir_q15_destroy(FIR_I); msdr_fir_q15_destroy(FIR_Q);
    printf("test_graph: %s\n", fails ? "FAILED" : "OK (graphs A, B, C, D bit-exact vs oracle)");
    return fails ? 1 : 0;
}
