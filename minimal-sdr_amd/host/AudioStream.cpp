// AudioStream.cpp -- see AudioStream.h.  Host-side bookkeeping only; all sample data stays in HBM.
#include "AudioStream.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

AudioGPUClass AudioGPU;

AudioStream *AudioStream::first_update = nullptr;
audio_block_t *AudioStream::pool = nullptr;
unsigned int AudioStream::pool_size = 0;
int16_t *AudioStream::pool_data = nullptr;
uint32_t *AudioStream::free_mask = nullptr;
uint16_t AudioStream::memory_used = 0;
uint16_t AudioStream::memory_used_max = 0;
bool AudioStream::update_locked = false;
double AudioStream::tick_seconds_last = 0.0;
double AudioStream::tick_seconds_max = 0.0;

int AudioGPUClass::begin(int device, uint32_t channels, void *hip_stream)
{
    if (ctx) end();
    if (channels == 0) return MSDR_STATUS_ARGUMENT_ERROR;
    int rc = msdr_ctx_create(device, hip_stream, &ctx);
    if (rc != 0) { ctx = nullptr; return rc; }     // no GPU: the graph cannot run (there is no CPU path)
    nchannels = channels;
    return 0;
}

void AudioGPUClass::end(void)
{
    AudioStream::release_memory();
    if (ctx) msdr_ctx_destroy(ctx);
    ctx = nullptr;
    nchannels = 0;
}

int AudioGPUClass::synchronize(void) { return ctx ? msdr_ctx_synchronize(ctx) : MSDR_STATUS_NO_DEVICE; }

AudioStream::AudioStream(unsigned char ninput, audio_block_t **iqueue)
    : active(false), num_inputs(ninput), destination_list(nullptr), inputQueue(iqueue), next_update(nullptr)
{
    for (int i = 0; i < num_inputs; i++) inputQueue[i] = nullptr;
    // append to the update list: nodes run in construction order (Minimal-SDR.ino:66-74)
    if (first_update == nullptr) first_update = this;
    else {
        AudioStream *p = first_update;
        while (p->next_update) p = p->next_update;
        p->next_update = this;
    }
}

int AudioStream::initialize_memory(unsigned int num)
{
    release_memory();
    if (!AudioGPU.ctx || num == 0 || num > 896) return MSDR_STATUS_ARGUMENT_ERROR;
    void *d = nullptr;
    int rc = msdr_malloc(AudioGPU.ctx, (size_t)num * AudioGPU.block_bytes(), &d);
    if (rc != 0) return rc;
    pool_data = (int16_t *)d;
    pool = (audio_block_t *)calloc(num, sizeof(audio_block_t));
    free_mask = (uint32_t *)calloc((num + 31) / 32, sizeof(uint32_t));
    pool_size = num;
    for (unsigned int i = 0; i < num; i++) {
        pool[i].memory_pool_index = (uint16_t)i;
        pool[i].data = pool_data + (size_t)i * AudioGPU.nchannels * AUDIO_BLOCK_SAMPLES;
        free_mask[i >> 5] |= (1u << (i & 31));
    }
    memory_used = memory_used_max = 0;
    return 0;
}

void AudioStream::release_memory(void)
{
    if (pool_data && AudioGPU.ctx) msdr_free(AudioGPU.ctx, pool_data);
    free(pool);
    free(free_mask);
    pool = nullptr; free_mask = nullptr; pool_data = nullptr; pool_size = 0;
    memory_used = 0;
}

// Returns a block batch with ref_count 1, or NULL when the pool is exhausted (nodes then drop the tick's data,
// record_queue.cpp:91-92, freq_conv.cpp:64).
audio_block_t *AudioStream::allocate(void)
{
    for (unsigned int w = 0; w < (pool_size + 31) / 32; w++) {
        if (free_mask[w]) {
            unsigned int bit = (unsigned int)__builtin_ctz(free_mask[w]);
            unsigned int idx = w * 32 + bit;
            if (idx >= pool_size) return nullptr;
            free_mask[w] &= ~(1u << bit);
            audio_block_t *b = pool + idx;
            b->ref_count = 1;
            if (++memory_used > memory_used_max) memory_used_max = memory_used;
            return b;
        }
    }
    return nullptr;
}

void AudioStream::release(audio_block_t *block)
{
    if (!block) return;
    if (block->ref_count > 1) { block->ref_count--; return; }
    block->ref_count = 0;
    unsigned int idx = block->memory_pool_index;
    free_mask[idx >> 5] |= (1u << (idx & 31));
    memory_used--;
}

// Hand `block` to every destination wired to output `index`; each takes one reference.  A destination whose
// input slot is still occupied keeps the older block (the core's behaviour).
void AudioStream::transmit(audio_block_t *block, unsigned char index)
{
    for (AudioConnection *c = destination_list; c != nullptr; c = c->next_dest) {
        if (c->src_index == index && c->dst.inputQueue[c->dest_index] == nullptr) {
            c->dst.inputQueue[c->dest_index] = block;
            block->ref_count++;
        }
    }
}

audio_block_t *AudioStream::receiveReadOnly(unsigned int index)
{
    if (index >= num_inputs) return nullptr;
    audio_block_t *in = inputQueue[index];
    inputQueue[index] = nullptr;
    return in;
}

audio_block_t *AudioStream::receiveWritable(unsigned int index)
{
    if (index >= num_inputs) return nullptr;
    audio_block_t *in = inputQueue[index];
    inputQueue[index] = nullptr;
    if (in && in->ref_count > 1) {          // shared: work on a private copy (device-to-device, stream ordered)
        audio_block_t *p = allocate();
        if (p) msdr_memcpy_d2d(AudioGPU.ctx, p->data, in->data, AudioGPU.block_bytes());
        in->ref_count--;
        in = p;
    }
    return in;
}

void AudioConnection::connect(void)
{
    if (dest_index >= dst.num_inputs) return;
    AudioConnection **pp = &src.destination_list;
    while (*pp) pp = &(*pp)->next_dest;
    *pp = this;
    src.active = true;
    dst.active = true;
}

void AudioStream::update_all(void)
{
    if (update_locked) return;
    auto t0 = std::chrono::steady_clock::now();
    for (AudioStream *p = first_update; p; p = p->next_update)
        if (p->active) p->update();
    tick_seconds_last = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (tick_seconds_last > tick_seconds_max) tick_seconds_max = tick_seconds_last;
}

// host time spent enqueueing one tick, relative to the real-time budget of one block
static double block_seconds(void) { return AUDIO_BLOCK_SAMPLES / (double)AUDIO_SAMPLE_RATE_EXACT; }
float AudioStream::cpu_usage_percent(void) { return (float)(100.0 * tick_seconds_last / block_seconds()); }
float AudioStream::cpu_usage_max_percent(void) { return (float)(100.0 * tick_seconds_max / block_seconds()); }
void AudioStream::cpu_usage_max_reset(void) { tick_seconds_max = tick_seconds_last; }
