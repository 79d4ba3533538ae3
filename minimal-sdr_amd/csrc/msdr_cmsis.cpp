// msdr_cmsis.cpp -- include/msdr_cmsis.h: the reference's CMSIS-DSP argument lists over the batched C ABI (no device code here).
#include "../../include/msdr_cmsis.h"

#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

// msdr_api.hip (not part of the C ABI): a pinned host buffer mapped into the device's address space
int msdr_mapped_alloc(msdr_ctx *ctx, size_t bytes, void **host, void **dev);
void msdr_mapped_free(msdr_ctx *ctx, void *host);

namespace {

// kind: 0 fir q15, 1 fir f32, 2 biquad df1 f32.  coef: the bytes of pCoeffs the device tables were last built from -- CMSIS reads
// the caller's array on every call (the instance holds a pointer, arm_fir_init_q15.c:100-109), and the reference rewrites it in place
// under the running filter (UI.cpp:337-345, Minimal-SDR.ino:221-223), so every process call compares and, on a change, rebuilds the
// tables with the filter state kept (msdr_fir_*_set_coeffs, msdr_biquad_df1_f32_set_coeffs).
struct Entry { int kind; void *handle; std::vector<char> coef; };
struct Binding {
    std::mutex mu;
    msdr_ctx *ctx = nullptr;
    uint32_t channels = 0;
    std::unordered_map<const void *, Entry> inst;  // keyed by the caller's instance struct, as CMSIS identifies a filter
    // msdr_cmsis_bind_host: pSrc / pDst are HOST arrays (the sketch's stack buffers, Minimal-SDR.ino:525-526, 574-575), staged through two
    // PINNED host buffers the device reads and writes in place (they grow with the largest block seen): a call is memcpy in, the kernel,
    // one stream synchronisation, memcpy out -- no copy command on the stream (two of them, each with a synchronisation of its own, were
    // 37 of the 41 us of a 128-sample call)
    bool host = false;
    void *h_in = nullptr, *h_out = nullptr;      // host addresses
    void *d_in = nullptr, *d_out = nullptr;      // the same buffers as the device sees them
    size_t cap = 0;
};
Binding &binding() { static Binding b; return b; }

void destroy(const Entry &e)
{
    if (e.kind == 0) msdr_fir_q15_destroy((msdr_fir_q15 *)e.handle);
    else if (e.kind == 1) msdr_fir_f32_destroy((msdr_fir_f32 *)e.handle);
    else msdr_biquad_df1_f32_destroy((msdr_biquad_df1_f32 *)e.handle);
}
// registers (or replaces: a re-init of the same instance) the device object behind S
void remember(const void *S, int kind, void *handle, const void *coef, size_t coef_bytes)
{
    Binding &b = binding();
    std::vector<char> snap((const char *)coef, (const char *)coef + (coef ? coef_bytes : 0));
    auto it = b.inst.find(S);
    if (it != b.inst.end()) { destroy(it->second); it->second = Entry{kind, handle, std::move(snap)}; }
    else b.inst.emplace(S, Entry{kind, handle, std::move(snap)});
}
// a failed (re-)init: the instance struct already shows the new filter, so the OLD device object must not answer for it any more --
// later process calls on S do nothing and msdr_last_error() holds the create's text
void forget(const void *S)
{
    Binding &b = binding();
    auto it = b.inst.find(S);
    if (it != b.inst.end()) { destroy(it->second); b.inst.erase(it); }
}
// (caller holds the lock, and keeps it across the process call: a concurrent re-init or msdr_cmsis_bind cannot destroy the object
//  under a running enqueue -- process calls only queue work on the context's stream, so the lock is held for microseconds)
Entry *lookup_locked(const void *S, int kind)
{
    Binding &b = binding();
    auto it = b.inst.find(S);
    return (it != b.inst.end() && it->second.kind == kind) ? &it->second : nullptr;
}
// The caller's coefficient array against the snapshot.  On a change the tables are rebuilt through `set` (state kept); the snapshot takes
// the new bytes only once that has SUCCEEDED -- after a failed rebuild (a cascade whose state cannot cross to the new coefficients, an
// allocation that failed) the old snapshot stays, so the next call compares unequal again and retries, and this call runs the filter with
// the tables it still has (pDst is written either way; msdr_last_error() has the text of the failure).
template <typename Set>
void follow_coeffs(Entry *e, const void *pCoeffs, Set &&set)
{
    if (!pCoeffs || e->coef.empty() || memcmp(e->coef.data(), pCoeffs, e->coef.size()) == 0) return;
    if (set() == 0) memcpy(e->coef.data(), pCoeffs, e->coef.size());
}
void drop_staging(Binding &b)
{
    if (b.ctx) { msdr_mapped_free(b.ctx, b.h_in); msdr_mapped_free(b.ctx, b.h_out); }
    b.h_in = b.h_out = b.d_in = b.d_out = nullptr; b.cap = 0;
}
// host-array binding: the block batch [channels][blockSize] of `esz`-byte samples goes into the pinned input buffer, `run` reads it and writes
// the pinned output buffer over PCIe, and the call returns when pDst holds the result, as the CMSIS function does.  Device-pointer binding:
// `run` on the caller's pointers.
template <typename Run>
void with_buffers(Binding &b, const void *pSrc, void *pDst, uint32_t blockSize, size_t esz, Run &&run)
{
    if (!b.host) { (void)run(pSrc, pDst); return; }
    const size_t bytes = (size_t)b.channels * blockSize * esz;
    if (bytes == 0 || !pSrc || !pDst) return;
    if (bytes > b.cap) {
        drop_staging(b);
        if (msdr_mapped_alloc(b.ctx, bytes, &b.h_in, &b.d_in) != 0 || msdr_mapped_alloc(b.ctx, bytes, &b.h_out, &b.d_out) != 0) { drop_staging(b); return; }
        b.cap = bytes;
    }
    memcpy(b.h_in, pSrc, bytes);
    if (run(b.d_in, b.d_out) != 0) return;
    if (msdr_ctx_synchronize(b.ctx) != 0) return;
    memcpy(pDst, b.h_out, bytes);
}

}  // namespace

// msdr_ctx_destroy calls this (msdr_api.hip): a context that goes away takes its binding and the objects made through it along,
// so that no msdr_arm_* call can reach a dangling handle.  Not part of the C ABI.
__attribute__((visibility("hidden"))) void msdr_cmsis_ctx_gone(msdr_ctx *ctx)
{
    Binding &b = binding();
    std::lock_guard<std::mutex> g(b.mu);
    if (b.ctx != ctx) return;
    for (auto &kv : b.inst) destroy(kv.second);
    b.inst.clear();
    drop_staging(b);
    b.ctx = nullptr; b.channels = 0; b.host = false;
}

static int bind_common(msdr_ctx *ctx, uint32_t channels, bool host)
{
    Binding &b = binding();
    std::lock_guard<std::mutex> g(b.mu);
    for (auto &kv : b.inst) destroy(kv.second);    // objects of the previous binding go with it
    b.inst.clear();
    drop_staging(b);
    b.ctx = ctx; b.channels = ctx ? channels : 0; b.host = ctx ? host : false;
    return (ctx && channels == 0) ? MSDR_STATUS_ARGUMENT_ERROR : MSDR_STATUS_SUCCESS;
}
extern "C" int msdr_cmsis_bind(msdr_ctx *ctx, uint32_t channels) { return bind_common(ctx, channels, false); }
extern "C" int msdr_cmsis_bind_host(msdr_ctx *ctx, uint32_t channels) { return bind_common(ctx, channels, true); }

extern "C" msdr_arm_status msdr_arm_fir_init_q15(msdr_arm_fir_instance_q15 *S, uint16_t numTaps, q15_t *pCoeffs, q15_t *pState, uint32_t blockSize)
{
    if (numTaps & 1u) return MSDR_ARM_MATH_ARGUMENT_ERROR;                       // arm_fir_init_q15.c:93-96: status only, S untouched
    S->numTaps = numTaps; S->pCoeffs = pCoeffs; S->pState = pState;              // :100-109
    if (pState) memset(pState, 0, ((size_t)numTaps + blockSize) * sizeof(q15_t));   // :106
    Binding &b = binding();
    std::lock_guard<std::mutex> g(b.mu);
    msdr_fir_q15 *h = nullptr;
    if (!b.ctx || msdr_fir_q15_create(b.ctx, numTaps, pCoeffs, b.channels, &h) != 0) { forget(S); return MSDR_ARM_MATH_ARGUMENT_ERROR; }
    remember(S, 0, h, pCoeffs, (size_t)numTaps * sizeof(q15_t));
    return MSDR_ARM_MATH_SUCCESS;
}
extern "C" void msdr_arm_fir_fast_q15(const msdr_arm_fir_instance_q15 *S, q15_t *pSrc, q15_t *pDst, uint32_t blockSize)
{
    std::lock_guard<std::mutex> g(binding().mu);
    Entry *e = lookup_locked(S, 0);
    if (!e) return;
    follow_coeffs(e, S->pCoeffs, [&] { return msdr_fir_q15_set_coeffs((msdr_fir_q15 *)e->handle, S->pCoeffs); });
    with_buffers(binding(), pSrc, pDst, blockSize, sizeof(q15_t), [&](const void *src, void *dst) { return msdr_fir_q15_process((msdr_fir_q15 *)e->handle, (const q15_t *)src, (q15_t *)dst, blockSize); });
}

extern "C" void msdr_arm_fir_init_f32(msdr_arm_fir_instance_f32 *S, uint16_t numTaps, float32_t *pCoeffs, float32_t *pState, uint32_t blockSize)
{
    S->numTaps = numTaps; S->pCoeffs = pCoeffs; S->pState = pState;
    if (pState && numTaps) memset(pState, 0, ((size_t)numTaps + blockSize - 1u) * sizeof(float32_t));     // state length arm_math.h:1050
    Binding &b = binding();
    std::lock_guard<std::mutex> g(b.mu);
    msdr_fir_f32 *h = nullptr;
    if (b.ctx && msdr_fir_f32_create(b.ctx, numTaps, pCoeffs, b.channels, &h) == 0) remember(S, 1, h, pCoeffs, (size_t)numTaps * sizeof(float32_t));
    else forget(S);
}
extern "C" void msdr_arm_fir_f32(const msdr_arm_fir_instance_f32 *S, float32_t *pSrc, float32_t *pDst, uint32_t blockSize)
{
    std::lock_guard<std::mutex> g(binding().mu);
    Entry *e = lookup_locked(S, 1);
    if (!e) return;
    follow_coeffs(e, S->pCoeffs, [&] { return msdr_fir_f32_set_coeffs((msdr_fir_f32 *)e->handle, S->pCoeffs); });
    with_buffers(binding(), pSrc, pDst, blockSize, sizeof(float32_t), [&](const void *src, void *dst) { return msdr_fir_f32_process((msdr_fir_f32 *)e->handle, (const float32_t *)src, (float32_t *)dst, blockSize); });
}

extern "C" void msdr_arm_biquad_cascade_df1_init_f32(msdr_arm_biquad_casd_df1_inst_f32 *S, uint8_t numStages, float32_t *pCoeffs, float32_t *pState)
{
    S->numStages = numStages; S->pCoeffs = pCoeffs; S->pState = pState;
    if (pState) memset(pState, 0, (size_t)4 * numStages * sizeof(float32_t));    // 4 state values per stage, arm_math.h:1233
    Binding &b = binding();
    std::lock_guard<std::mutex> g(b.mu);
    msdr_biquad_df1_f32 *h = nullptr;
    if (b.ctx && msdr_biquad_df1_f32_create(b.ctx, numStages, pCoeffs, b.channels, &h) == 0) remember(S, 2, h, pCoeffs, (size_t)5 * numStages * sizeof(float32_t));
    else forget(S);
}
extern "C" void msdr_arm_biquad_cascade_df1_f32(const msdr_arm_biquad_casd_df1_inst_f32 *S, float32_t *pSrc, float32_t *pDst, uint32_t blockSize)
{
    std::lock_guard<std::mutex> g(binding().mu);
    Entry *e = lookup_locked(S, 2);
    if (!e) return;
    follow_coeffs(e, S->pCoeffs, [&] { return msdr_biquad_df1_f32_set_coeffs((msdr_biquad_df1_f32 *)e->handle, S->pCoeffs); });
    with_buffers(binding(), pSrc, pDst, blockSize, sizeof(float32_t), [&](const void *src, void *dst) { return msdr_biquad_df1_f32_process((msdr_biquad_df1_f32 *)e->handle, (const float32_t *)src, (float32_t *)dst, blockSize); });
}
