// msdr_nodes.h -- the graph nodes on the demodulation path, as AudioStream subclasses over the C ABI.
// Class and method names follow the reference so a sketch's wiring reads the same:
//   AudioFilterBiquad    src/Audio/filter_biquad.{h,cpp}
//   AudioEffectFreqConv  freq_conv.{h,cpp}
//   AudioRecordQueue     src/Audio/record_queue.{h,cpp}     (graph -> main loop)
//   AudioPlayQueue       src/Audio/play_queue.{h,cpp}       (main loop -> graph)
// plus AudioSDRDemodulator, the fused node (one kernel for demodulation() + the biquad nodes behind it).
// Every block is a block batch in HBM (AudioStream.h); update() enqueues kernels, never touches samples.
#pragma once
#include <atomic>
#include <vector>

#include "AudioStream.h"

// ------------------------------------------------------------------------------------------------
// AudioFilterBiquad -- up to 4 cascaded biquads, Q2.30 coefficients, int16 data (filter_biquad.h:33-155)
// ------------------------------------------------------------------------------------------------
class AudioFilterBiquad : public AudioStream {
public:
    AudioFilterBiquad(void) : AudioStream(1, inputQueueArray), node(nullptr) {}
    ~AudioFilterBiquad() { if (node) msdr_biquad_q15_destroy(node); }
    virtual void update(void)
    {
        audio_block_t *block = receiveWritable();
        if (!block) return;                                    // filter_biquad.cpp:41
        if (ensure()) msdr_biquad_q15_update(node, block->data, AUDIO_BLOCK_SAMPLES);
        transmit(block);
        release(block);
    }
    void setCoefficients(uint32_t stage, const int *coefficients)
    {
        if (stage >= 4) return;                                // filter_biquad.cpp:86
        pending.push_back({stage, {coefficients[0], coefficients[1], coefficients[2], coefficients[3], coefficients[4]}});
        if (AudioGPU.context()) ensure();
    }
    void setCoefficients(uint32_t stage, const double *coefficients)      // filter_biquad.h:43-51
    {
        int coef[5];
        for (int i = 0; i < 5; i++) coef[i] = (int)(coefficients[i] * 1073741824.0);
        setCoefficients(stage, coef);
    }
    void setLowpass(uint32_t stage, float frequency, float q = 0.7071f) { design(stage, MSDR_BQ_LOWPASS, frequency, q, 1.0f); }
    void setHighpass(uint32_t stage, float frequency, float q = 0.7071f) { design(stage, MSDR_BQ_HIGHPASS, frequency, q, 1.0f); }
    void setBandpass(uint32_t stage, float frequency, float q = 1.0f) { design(stage, MSDR_BQ_BANDPASS, frequency, q, 1.0f); }
    void setNotch(uint32_t stage, float frequency, float q = 1.0f) { design(stage, MSDR_BQ_NOTCH, frequency, q, 1.0f); }
    void setLowShelf(uint32_t stage, float frequency, float gain, float slope = 1.0f) { design(stage, MSDR_BQ_LOWSHELF, frequency, gain, slope); }
    void setHighShelf(uint32_t stage, float frequency, float gain, float slope = 1.0f) { design(stage, MSDR_BQ_HIGHSHELF, frequency, gain, slope); }
    // definition[32] of one channel, as filter_biquad.h:152 lays it out (test/debug aid)
    int getDefinition(uint32_t channel, int32_t definition[32]) { return ensure() ? msdr_biquad_q15_get_definition(node, channel, definition) : MSDR_STATUS_NO_DEVICE; }

private:
    struct Pending { uint32_t stage; int32_t coef[5]; };
    void design(uint32_t stage, int kind, float f, float q, float slope)
    {
        int32_t coef[5];
        if (msdr_biquad_design(kind, f, q, slope, AUDIO_SAMPLE_RATE_EXACT, coef) == 0) setCoefficients(stage, (const int *)coef);
    }
    bool ensure(void)
    {
        if (!node) {
            if (!AudioGPU.context() || msdr_biquad_q15_create(AudioGPU.context(), AudioGPU.channels(), &node) != 0) { node = nullptr; return false; }
        }
        for (const Pending &p : pending) msdr_biquad_q15_set_coefficients(node, p.stage, p.coef);
        pending.clear();
        return true;
    }
    msdr_biquad_q15 *node;
    std::vector<Pending> pending;
    audio_block_t *inputQueueArray[1];
};

// ------------------------------------------------------------------------------------------------
// AudioAmplifier -- gain stage, Q16.16 multiplier, saturating (mixer.h:67-82, mixer.cpp:34-47, :134-159).
// amp_adc / amp_dac of the sketch (Minimal-SDR.ino:67, :73); AGC() drives amp_adc.gain() (:446-515).
// ------------------------------------------------------------------------------------------------
class AudioAmplifier : public AudioStream {
public:
    AudioAmplifier(void) : AudioStream(1, inputQueueArray), multiplier(65536) {}
    virtual void update(void)
    {
        const int32_t mult = multiplier;
        if (mult == 0) {                                       // mixer.cpp:139-142: discard, transmit nothing
            audio_block_t *block = receiveReadOnly(0);
            if (block) release(block);
        } else if (mult == 65536) {                            // :143-149: unity, pass the block on untouched
            audio_block_t *block = receiveReadOnly(0);
            if (block) { transmit(block); release(block); }
        } else {                                               // :150-157
            audio_block_t *block = receiveWritable(0);
            if (block) {
                if (AudioGPU.context()) msdr_amp_q15(AudioGPU.context(), mult, block->data, AudioGPU.channels(), AUDIO_BLOCK_SAMPLES, nullptr);
                transmit(block);
                release(block);
            }
        }
    }
    void gain(float n) { multiplier = msdr_amp_multiplier(n); } // mixer.h:75-79

private:
    int32_t multiplier;
    audio_block_t *inputQueueArray[1];
};

// ------------------------------------------------------------------------------------------------
// AudioEffectFreqConv -- complex (I,Q) x oscillator mix in q15 (freq_conv.h:36-56, freq_conv.cpp:30-116).
// The oscillator tables are globals of the application, exactly as in the reference (freq_conv.h:33-34).
// Unlike the reference the kernel needs no temporary blocks, so the node cannot run out of pool memory.
// ------------------------------------------------------------------------------------------------
extern q15_t Osc_Q_buffer_i[AUDIO_BLOCK_SAMPLES];
extern q15_t Osc_I_buffer_i[AUDIO_BLOCK_SAMPLES];

class AudioEffectFreqConv : public AudioStream {
public:
    AudioEffectFreqConv() : AudioStream(2, inputQueueArray), dir(0), pass(1) {}
    void direction(bool d) { dir = d; }
    void passthrough(bool p) { pass = p; }
    virtual void update(void)
    {
        audio_block_t *blockI = receiveWritable(0);
        audio_block_t *blockQ = receiveWritable(1);
        if (!blockI) { if (blockQ) release(blockQ); return; }                 // freq_conv.cpp:40-47
        if (!blockQ) { release(blockI); return; }
        // `pass` keeps the reference's inverted meaning: pass == false forwards untouched (:49-56)
        msdr_freqconv_q15(AudioGPU.context(), blockI->data, blockQ->data, Osc_I_buffer_i, Osc_Q_buffer_i, AUDIO_BLOCK_SAMPLES,
                          dir ? 1 : 0, pass ? 1 : 0, AudioGPU.channels(), AUDIO_BLOCK_SAMPLES);
        transmit(blockI, 0);
        transmit(blockQ, 1);
        release(blockI);
        release(blockQ);
    }

private:
    audio_block_t *inputQueueArray[2];
    bool dir, pass;
};

// ------------------------------------------------------------------------------------------------
// BlockRing -- a single-producer / single-consumer ring of block pointers.  One side is the graph tick (update(), the software
// interrupt of the reference), the other the main loop; each side owns one free-running counter, so neither ever writes what
// the other one writes and a full ring is told from an empty one without giving up a slot.
// ------------------------------------------------------------------------------------------------
template <unsigned CAPACITY>
class BlockRing {
public:
    unsigned size(void) const { return pushed.load(std::memory_order_acquire) - popped.load(std::memory_order_acquire); }
    bool push(audio_block_t *block)                   // producer side; false = full
    {
        const unsigned w = pushed.load(std::memory_order_relaxed);
        if (w - popped.load(std::memory_order_acquire) >= CAPACITY) return false;
        slot[w % CAPACITY] = block;
        pushed.store(w + 1, std::memory_order_release);
        return true;
    }
    audio_block_t *pop(void)                          // consumer side; NULL = empty
    {
        const unsigned r = popped.load(std::memory_order_relaxed);
        if (r == pushed.load(std::memory_order_acquire)) return nullptr;
        audio_block_t *block = slot[r % CAPACITY];
        popped.store(r + 1, std::memory_order_release);
        return block;
    }

private:
    audio_block_t *slot[CAPACITY];
    std::atomic<unsigned> pushed{0}, popped{0};
};

// ------------------------------------------------------------------------------------------------
// AudioRecordQueue -- hands graph blocks to the main loop (interface of record_queue.h:34-58; the reference keeps 53 slots of which
// 52 can be occupied, and drops the incoming block when they are: record_queue.cpp:91-92).
// readBuffer() returns a DEVICE pointer to [channels][128] int16.
// ------------------------------------------------------------------------------------------------
class AudioRecordQueue : public AudioStream {
public:
    AudioRecordQueue(void) : AudioStream(1, inputQueueArray), lent(nullptr), recording(false) {}
    void begin(void) { clear(); recording.store(true); }
    void end(void) { recording.store(false); }
    int available(void) { return (int)ring.size(); }
    void clear(void)
    {
        freeBuffer();
        while (audio_block_t *b = ring.pop()) release(b);
    }
    int16_t *readBuffer(void)                         // one block at a time is out with the main loop
    {
        if (lent) return nullptr;
        lent = ring.pop();
        return lent ? lent->data : nullptr;
    }
    void freeBuffer(void)
    {
        if (lent) release(lent);
        lent = nullptr;
    }
    virtual void update(void)
    {
        audio_block_t *block = receiveReadOnly();
        if (!block) return;
        if (!recording.load() || !ring.push(block)) release(block);
    }

private:
    audio_block_t *inputQueueArray[1];
    BlockRing<52> ring;
    audio_block_t *lent;
    std::atomic<bool> recording;
};

// ------------------------------------------------------------------------------------------------
// AudioPlayQueue -- main loop feeds blocks into the graph (interface of play_queue.h:34-52; 32 slots of which 31 can be occupied).
// Where the reference busy-waits (allocate() in getBuffer, a full ring in playBuffer: play_queue.cpp:42-46, :56) these return
// NULL / false: there is no interrupt here that could end the wait.
// ------------------------------------------------------------------------------------------------
class AudioPlayQueue : public AudioStream {
public:
    AudioPlayQueue(void) : AudioStream(0, nullptr), filling(nullptr) {}
    bool available(void) { return getBuffer() != nullptr; }
    int16_t *getBuffer(void)
    {
        if (!filling) filling = allocate();
        return filling ? filling->data : nullptr;
    }
    bool playBuffer(void)
    {
        if (!filling || !ring.push(filling)) return false;
        filling = nullptr;
        return true;
    }
    virtual void update(void)
    {
        if (audio_block_t *block = ring.pop()) {
            transmit(block);
            release(block);
        }
    }

private:
    BlockRing<31> ring;
    audio_block_t *filling;
};

// ------------------------------------------------------------------------------------------------
// AudioSDRDemodulator -- the whole demodulation() step (+ optional biquad nodes) as ONE graph node and ONE
// kernel per tick: IF block batch in, audio block batch out.  Configure with a msdr_chain_config whose
// `channels` equals AudioGPU.channels() and whose arith is MSDR_ARITH_Q15 (int16 audio blocks).
// ------------------------------------------------------------------------------------------------
class AudioSDRDemodulator : public AudioStream {
public:
    AudioSDRDemodulator(void) : AudioStream(1, inputQueueArray), chain(nullptr) {}
    ~AudioSDRDemodulator() { if (chain) msdr_chain_destroy(chain); }
    int begin(const msdr_chain_config &cfg)
    {
        if (chain) { msdr_chain_destroy(chain); chain = nullptr; }
        if (cfg.arith != MSDR_ARITH_Q15 || cfg.channels != AudioGPU.channels()) return MSDR_STATUS_ARGUMENT_ERROR;
        return msdr_chain_create(AudioGPU.context(), &cfg, &chain);
    }
    int init_FIR(void) { return chain ? msdr_chain_init_fir(chain) : MSDR_STATUS_ARGUMENT_ERROR; }     // Minimal-SDR.ino:901-930
    int setMode(uint32_t channel, int mode, int tapset) { return chain ? msdr_chain_set_mode(chain, channel, mode, tapset) : MSDR_STATUS_ARGUMENT_ERROR; }
    // what the sketch changes while audio plays, filter state kept (include/msdr.h "live updates"):
    // calc_demod_filter() rewriting FIR_AM_coeffs (Minimal-SDR.ino:221-223), tune()'s biquad2_dac.setNotch (:356), new oscillator tables (freq_conv.h:33-34)
    int setTaps(uint32_t tapset, const void *coeffs_i, const void *coeffs_q) { return chain ? msdr_chain_set_taps(chain, tapset, coeffs_i, coeffs_q) : MSDR_STATUS_ARGUMENT_ERROR; }
    int setNodeCoefficients(uint32_t node, uint32_t stage, const int *coef) { return chain ? msdr_chain_set_node_coefficients(chain, node, stage, (const int32_t *)coef) : MSDR_STATUS_ARGUMENT_ERROR; }
    int setNodeNotch(uint32_t node, uint32_t stage, float frequency, float q = 1.0f)
    {
        int32_t coef[5];
        if (int rc = msdr_biquad_design(MSDR_BQ_NOTCH, frequency, q, 1.0f, AUDIO_SAMPLE_RATE_EXACT, coef)) return rc;
        return setNodeCoefficients(node, stage, (const int *)coef);
    }
    int setOsc(const void *osc_i, const void *osc_q) { return chain ? msdr_chain_set_osc(chain, osc_i, osc_q) : MSDR_STATUS_ARGUMENT_ERROR; }
    virtual void update(void)
    {
        audio_block_t *in = receiveReadOnly();
        if (!in) return;
        audio_block_t *out = chain ? allocate() : nullptr;
        if (out) {
            msdr_chain_process(chain, in->data, out->data, AUDIO_BLOCK_SAMPLES);
            transmit(out);
            release(out);
        }
        release(in);
    }

private:
    audio_block_t *inputQueueArray[1];
    msdr_chain *chain;
};
