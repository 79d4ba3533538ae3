// AudioStream.h -- the operator-graph runtime the reference's nodes plug into, rebuilt for MI355X.
//
// The reference #includes "AudioStream.h" from the un-vendored Teensyduino core (freq_conv.h:30,
// src/Audio/filter_biquad.h:31); its contract is reconstructed from the call sites:
//   ctor   AudioStream(n_inputs, audio_block_t **inputQueueArray)        freq_conv.h:39, filter_biquad.h:36
//   node   virtual void update(void)                                      freq_conv.h:50
//   inside update(): receiveReadOnly(i) record_queue.cpp:83, receiveWritable(i) freq_conv.cpp:37-38,
//          allocate() freq_conv.cpp:58-61, transmit(block, out) :107-108, release(block) :109-113
//   wiring AudioConnection(src, dst) / (src, srcPort, dst, dstPort)       Minimal-SDR.ino:76-81
//   pool   AudioMemory(n)                                                 Minimal-SDR.ino:83,373
//   tick   AudioStream::update_all(): every node's update(), construction order, once per 128 samples
//          (input_adc.cpp:122, output_dac.cpp:153; order Minimal-SDR.ino:66-74)
//   stats  AudioProcessorUsageMax(), AudioMemoryUsageMax(), ...Reset()    Minimal-SDR.ino:365,406,424-426
//
// MI355X-first difference: one audio_block_t is a BLOCK BATCH -- `channels` x AUDIO_BLOCK_SAMPLES int16
// in HBM, `data` is a DEVICE pointer -- and every node's update() enqueues kernels on the context's HIP
// stream through the C ABI (include/msdr.h) instead of looping over samples on the CPU.  Ownership rules
// are the core's: receive* hands the slot's reference to the node, transmit adds one reference per
// connected destination, receiveWritable yields a private copy when the block is shared.
#pragma once
#include <stdint.h>

#include "../../include/msdr.h"

#ifndef AUDIO_BLOCK_SAMPLES
#define AUDIO_BLOCK_SAMPLES MSDR_AUDIO_BLOCK_SAMPLES
#endif
#ifndef AUDIO_SAMPLE_RATE_EXACT
#define AUDIO_SAMPLE_RATE_EXACT MSDR_AUDIO_SAMPLE_RATE_EXACT
#endif

typedef struct audio_block_struct {
    uint8_t ref_count;
    uint8_t reserved1;
    uint16_t memory_pool_index;
    int16_t *data;      // DEVICE pointer: [channels][AUDIO_BLOCK_SAMPLES]
} audio_block_t;

class AudioStream;

class AudioConnection {
public:
    AudioConnection(AudioStream &source, AudioStream &destination)
        : src(source), dst(destination), src_index(0), dest_index(0), next_dest(nullptr) { connect(); }
    AudioConnection(AudioStream &source, unsigned char sourceOutput, AudioStream &destination, unsigned char destinationInput)
        : src(source), dst(destination), src_index(sourceOutput), dest_index(destinationInput), next_dest(nullptr) { connect(); }
    friend class AudioStream;

protected:
    void connect(void);
    AudioStream &src;
    AudioStream &dst;
    unsigned char src_index;
    unsigned char dest_index;
    AudioConnection *next_dest;
};

// Binds the graph runtime to one GPU / one channel count before AudioMemory().
struct AudioGPUClass {
    // device: HIP ordinal; channels: receiver channels carried by every block batch; stream: hipStream_t or NULL
    int begin(int device, uint32_t channels, void *hip_stream = nullptr);
    void end(void);
    msdr_ctx *context(void) const { return ctx; }
    uint32_t channels(void) const { return nchannels; }
    size_t block_bytes(void) const { return (size_t)nchannels * AUDIO_BLOCK_SAMPLES * sizeof(int16_t); }
    int synchronize(void);
    msdr_ctx *ctx = nullptr;
    uint32_t nchannels = 0;
};
extern AudioGPUClass AudioGPU;

#define AudioMemory(num) AudioStream::initialize_memory(num)
#define AudioProcessorUsage() (AudioStream::cpu_usage_percent())
#define AudioProcessorUsageMax() (AudioStream::cpu_usage_max_percent())
#define AudioProcessorUsageMaxReset() (AudioStream::cpu_usage_max_reset())
#define AudioMemoryUsage() (AudioStream::memory_used)
#define AudioMemoryUsageMax() (AudioStream::memory_used_max)
#define AudioMemoryUsageMaxReset() (AudioStream::memory_used_max = AudioStream::memory_used)
// src/Audio/Audio.h:55-56: graph changes from the control thread are bracketed by these; here the tick runs on
// the caller's thread, so they only mark the critical section.
#define AudioNoInterrupts() (AudioStream::update_locked = true)
#define AudioInterrupts() (AudioStream::update_locked = false)

class AudioStream {
public:
    AudioStream(unsigned char ninput, audio_block_t **iqueue);
    virtual ~AudioStream() {}
    static int initialize_memory(unsigned int num);   // AudioMemory(num): num block batches in HBM
    static void release_memory(void);
    // one graph tick: every node with a connection runs update() in construction order
    static void update_all(void);
    static float cpu_usage_percent(void);
    static float cpu_usage_max_percent(void);
    static void cpu_usage_max_reset(void);
    static uint16_t memory_used;
    static uint16_t memory_used_max;
    static bool update_locked;
    bool isActive(void) const { return active; }

protected:
    bool active;
    unsigned char num_inputs;
    static audio_block_t *allocate(void);
    static void release(audio_block_t *block);
    void transmit(audio_block_t *block, unsigned char index = 0);
    audio_block_t *receiveReadOnly(unsigned int index = 0);
    audio_block_t *receiveWritable(unsigned int index = 0);
    virtual void update(void) = 0;
    friend class AudioConnection;

private:
    AudioConnection *destination_list;
    audio_block_t **inputQueue;
    AudioStream *next_update;
    static AudioStream *first_update;
    static audio_block_t *pool;
    static unsigned int pool_size;
    static int16_t *pool_data;
    static uint32_t *free_mask;
    static double tick_seconds_last, tick_seconds_max;
};
