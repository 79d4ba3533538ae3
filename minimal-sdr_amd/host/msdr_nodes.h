// msdr_nodes.h -- the graph nodes on the demodulation path, as AudioStream subclasses over the C ABI.
// Class and method names follow the reference so a sketch's wiring reads the same:
//   AudioFilterBiquad    src/Audio/filter_biquad.{h,cpp}
//   AudioEffectFreqConv  freq_conv.{h,cpp}
//   AudioRecordQueue     src/Audio/record_queue.{h,cpp}     (graph -> main loop)
//   AudioPlayQueue       src/Audio/play_queue.{h,cpp}       (main loop -> graph)
// plus AudioSDRDemodulator, the fused node (one kernel for demodulation() + the biquad nodes behind it).
// Every block is a block batch in HBM (AudioStream.h); update() enqueues kernels, never touches samples.
#pragma once
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
// AudioRecordQueue -- hands graph blocks to the main loop (record_queue.h:34-58, record_queue.cpp:31-97).
// readBuffer() returns a DEVICE pointer to [channels][128] int16.
// ------------------------------------------------------------------------------------------------
class AudioRecordQueue : public AudioStream {
public:
    AudioRecordQueue(void) : AudioStream(1, inputQueueArray), userblock(nullptr), head(0), tail(0), enabled(0) {}
    void begin(void) { clear(); enabled = 1; }
    int available(void)
    {
        uint32_t h = head, t = tail;
        return (h >= t) ? (int)(h - t) : (int)(kSlots + h - t);
    }
    void clear(void)
    {
        if (userblock) { release(userblock); userblock = nullptr; }
        uint32_t t = tail;
        while (t != head) { if (++t >= kSlots) t = 0; release(queue[t]); }
        tail = t;
    }
    int16_t *readBuffer(void)
    {
        if (userblock) return nullptr;
        uint32_t t = tail;
        if (t == head) return nullptr;
        if (++t >= kSlots) t = 0;
        userblock = queue[t];
        tail = t;
        return userblock->data;
    }
    void freeBuffer(void)
    {
        if (userblock == nullptr) return;
        release(userblock);
        userblock = nullptr;
    }
    void end(void) { enabled = 0; }
    virtual void update(void)
    {
        audio_block_t *block = receiveReadOnly();
        if (!block) return;
        if (!enabled) { release(block); return; }
        uint32_t h = head + 1;
        if (h >= kSlots) h = 0;
        if (h == tail) release(block);             // ring full: the block is dropped (record_queue.cpp:91-92)
        else { queue[h] = block; head = h; }
    }

private:
    static const uint32_t kSlots = 53;             // record_queue.h:52
    audio_block_t *inputQueueArray[1];
    audio_block_t *volatile queue[kSlots];
    audio_block_t *userblock;
    volatile uint32_t head, tail, enabled;
};

// ------------------------------------------------------------------------------------------------
// AudioPlayQueue -- main loop feeds blocks into the graph (play_queue.h:34-52, play_queue.cpp:31-75).
// ------------------------------------------------------------------------------------------------
class AudioPlayQueue : public AudioStream {
public:
    AudioPlayQueue(void) : AudioStream(0, nullptr), userblock(nullptr), head(0), tail(0) {}
    bool available(void)
    {
        if (userblock) return true;
        userblock = allocate();
        return userblock != nullptr;
    }
    int16_t *getBuffer(void)                         // the reference spins on allocate(); here: NULL when the pool is empty
    {
        if (userblock) return userblock->data;
        userblock = allocate();
        return userblock ? userblock->data : nullptr;
    }
    bool playBuffer(void)                            // false when the 32-slot ring is full (the reference busy-waits, play_queue.cpp:56)
    {
        if (!userblock) return false;
        uint32_t h = head + 1;
        if (h >= kSlots) h = 0;
        if (tail == h) return false;
        queue[h] = userblock;
        head = h;
        userblock = nullptr;
        return true;
    }
    virtual void update(void)
    {
        uint32_t t = tail;
        if (t == head) return;
        if (++t >= kSlots) t = 0;
        audio_block_t *block = queue[t];
        tail = t;
        transmit(block);
        release(block);
    }

private:
    static const uint32_t kSlots = 32;              // play_queue.h:47
    audio_block_t *volatile queue[kSlots];
    audio_block_t *userblock;
    volatile uint32_t head, tail;
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
