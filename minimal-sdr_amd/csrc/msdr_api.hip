// msdr_api.hip -- C ABI (include/msdr.h) over the gfx950 kernels in msdr_kernels.hiph.
// Host logic only: argument checks mirroring the reference's error behaviour, coefficient/table
// preparation, state ownership in HBM, launch geometry.  No CPU compute path exists here: every
// process/update entry point launches HIP kernels or fails.
#include <dlfcn.h>
#include "../../include/msdr.h"
#include "msdr_kernels.hiph"
#include "msdr_chain_fold.hiph"
#include <type_traits>
#include "msdr_chain_mfma.hiph"
#include "msdr_chain_mfw.hiph"
#include "msdr_frontend.hiph"
#include "msdr_spectrum.hiph"
#include "msdr_chain_q15mf.hiph"
#include "msdr_fir_f32mf.hiph"
#include "msdr_fir_f32tr.hiph"
#include "msdr_fir_f32tq.hiph"
#include "msdr_chain_amtr.hiph"
#include "msdr_chain_mfb.hiph"
#include "msdr_chain_q15mb.hiph"
#include "msdr_block.h"
#include "msdr_design.h"
#include "msdr_cascade_state.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <vector>

using namespace msdr;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? MSDR_STATUS_OUT_OF_MEMORY : MSDR_STATUS_HIP_ERROR, \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *msdr_last_error(void) { return g_err.c_str(); }
extern "C" const char *msdr_version(void) { return "msdr 0.1 (gfx950, hip)"; }

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct msdr_ctx {
    int device;
    hipStream_t stream;
    bool owns_stream;
    int num_cus;
    void *scratch;          // small device buffer reused for per-call host tables (oscillator tables)
    size_t scratch_bytes;
    int16_t *d_fft_tables;  // twiddle / split tables of the 128-point q15 real FFT, created on first use
    // msdr_ctx_enable_kernel_timing: HIP events on `stream` around the MAIN kernel of the stage mirrors (msdr_fir_*_process)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double timed_ms = 0.0;
    uint64_t timed_launches = 0;
};

// brackets one kernel launch with events when the context's timing is on (at most 8192 pending pairs)
struct KernelTimer {
    msdr_ctx *ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    explicit KernelTimer(msdr_ctx *c) : ctx(c)
    {
        if (!c->timing || c->events.size() >= 8192) return;
        if (hipEventCreate(&e0) != hipSuccess) { e0 = nullptr; return; }
        if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); e0 = e1 = nullptr; return; }
        if (hipEventRecord(e0, c->stream) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); e0 = e1 = nullptr; }
    }
    ~KernelTimer()
    {
        if (!e0) return;
        if (hipEventRecord(e1, ctx->stream) == hipSuccess) ctx->events.emplace_back(e0, e1);
        else { hipEventDestroy(e0); hipEventDestroy(e1); }
    }
};

static int bind(msdr_ctx *ctx)
{
    if (!ctx) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    (void)hipGetLastError();      // launch_check() reads the sticky last error: start every entry point with a clean slate
    return 0;
}

extern "C" int msdr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int msdr_ctx_create(int device, void *hip_stream, msdr_ctx **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(MSDR_STATUS_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n) return fail(MSDR_STATUS_ARGUMENT_ERROR, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MSDR_STATUS_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    msdr_ctx *c = new (std::nothrow) msdr_ctx();
    if (!c) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    c->device = device; c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->scratch = nullptr; c->scratch_bytes = 0; c->d_fft_tables = nullptr;
    c->owns_stream = (hip_stream == nullptr);
    c->stream = (hipStream_t)hip_stream;
    if (c->owns_stream) {
        hipError_t e2 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e2 != hipSuccess) { delete c; return fail(MSDR_STATUS_HIP_ERROR, "hipStreamCreate: %s", hipGetErrorString(e2)); }
    }
    *out = c;
    return 0;
}

void msdr_cmsis_ctx_gone(msdr_ctx *ctx);          // msdr_cmsis.cpp: drops a CMSIS binding (and its objects) that points at this context

extern "C" int msdr_ctx_destroy(msdr_ctx *ctx)
{
    if (!ctx) return 0;
    if (int rc = bind(ctx)) return rc;
    msdr_cmsis_ctx_gone(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->d_fft_tables) (void)hipFree(ctx->d_fft_tables);
    for (auto &e : ctx->events) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

extern "C" int msdr_ctx_enable_kernel_timing(msdr_ctx *ctx, int on)
{
    if (!ctx) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null context");
    ctx->timing = on != 0;
    return 0;
}
extern "C" int msdr_ctx_get_kernel_time(msdr_ctx *ctx, double *total_ms, uint64_t *launches, int reset)
{
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto &e : ctx->events) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
        ctx->timed_ms += ms; ctx->timed_launches++;
        hipEventDestroy(e.first); hipEventDestroy(e.second);
    }
    ctx->events.clear();
    if (total_ms) *total_ms = ctx->timed_ms;
    if (launches) *launches = ctx->timed_launches;
    if (reset) { ctx->timed_ms = 0; ctx->timed_launches = 0; }
    return 0;
}

extern "C" int msdr_ctx_synchronize(msdr_ctx *ctx)
{
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" void *msdr_ctx_stream(msdr_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int msdr_malloc(msdr_ctx *ctx, size_t bytes, void **d_ptr)
{
    if (int rc = bind(ctx)) return rc;
    if (!d_ptr) return fail(MSDR_STATUS_ARGUMENT_ERROR, "d_ptr is null");
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 16));
    return 0;
}
extern "C" int msdr_free(msdr_ctx *ctx, void *d_ptr)
{
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d_ptr));
    return 0;
}
extern "C" int msdr_memcpy_h2d(msdr_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int msdr_memcpy_d2h(msdr_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}
// Not part of the C ABI (msdr_cmsis.cpp's host-array binding): a pinned host buffer the device reads and writes in place -- the shim's block
// batch then needs no copy command on either side of the kernel, only the stream synchronisation CMSIS semantics ask for anyway.
__attribute__((visibility("hidden"))) int msdr_mapped_alloc(msdr_ctx *ctx, size_t bytes, void **host, void **dev)
{
    if (int rc = bind(ctx)) return rc;
    *host = nullptr; *dev = nullptr;
    if (hipHostMalloc(host, bytes, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return fail(MSDR_STATUS_OUT_OF_MEMORY, "hipHostMalloc(%zu) failed", bytes); }
    if (hipHostGetDevicePointer(dev, *host, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(*host); *host = nullptr; return fail(MSDR_STATUS_HIP_ERROR, "hipHostGetDevicePointer failed"); }
    return 0;
}
__attribute__((visibility("hidden"))) void msdr_mapped_free(msdr_ctx *ctx, void *host)
{
    if (host && bind(ctx) == 0) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(host); }
}
extern "C" int msdr_memcpy_d2d(msdr_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
extern "C" int msdr_memset(msdr_ctx *ctx, void *d_dst, int value, size_t bytes)
{
    if (int rc = bind(ctx)) return rc;
    HIP_TRY(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return 0;
}

// small RAII-free helper: device buffer filled from a host vector
template <typename T>
static int upload(msdr_ctx *ctx, const std::vector<T> &h, T **d)
{
    *d = nullptr;
    HIP_TRY(hipMalloc((void **)d, std::max<size_t>(h.size() * sizeof(T), 16)));
    if (!h.empty()) {
        // (a failing copy must not leave the buffer behind: callers treat a non-zero return as "nothing was allocated")
        hipError_t e = hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(*d); *d = nullptr;
            return fail(MSDR_STATUS_HIP_ERROR, "upload of %zu bytes failed: %s", h.size() * sizeof(T), hipGetErrorString(e));
        }
    }
    return 0;
}
template <typename T>
static int dzalloc(msdr_ctx *ctx, size_t count, T **d)
{
    *d = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    HIP_TRY(hipMalloc((void **)d, bytes));
    const hipError_t e = hipMemsetAsync(*d, 0, bytes, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(*d); *d = nullptr;
        return fail(MSDR_STATUS_HIP_ERROR, "clearing %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    return 0;
}
static int grid_1d(long long total, int block = 256)
{
    long long g = (total + block - 1) / block;
    return (int)std::max<long long>(1, std::min<long long>(g, 256LL * 8));   // grid-stride above 2048 blocks
}
static int launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSDR_STATUS_HIP_ERROR, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// RCCL gather of demodulated audio: the one exchange step of the path, from C.
// RCCL is loaded on first use (dlopen: a process that already carries an RCCL -- torch's -- shares it; nothing is linked in).
// ------------------------------------------------------------------------------------------------
namespace {
struct RcclUniqueId { char internal[128]; };
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi *rccl_api()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.lib ? &api : nullptr;
    tried = true;
    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
    }
    if (!api.lib) return nullptr;
#define MSDR_RCCL_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, sym)); if (!api.field) { api.lib = nullptr; return nullptr; }
    MSDR_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") MSDR_RCCL_SYM(CommInitRank, "ncclCommInitRank") MSDR_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    MSDR_RCCL_SYM(GroupStart, "ncclGroupStart") MSDR_RCCL_SYM(GroupEnd, "ncclGroupEnd") MSDR_RCCL_SYM(Send, "ncclSend") MSDR_RCCL_SYM(Recv, "ncclRecv")
    MSDR_RCCL_SYM(AllGather, "ncclAllGather") MSDR_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef MSDR_RCCL_SYM
    return &api;
}
constexpr int kRcclChar = 0;             // ncclInt8 / ncclChar: the audio travels as bytes, whatever its sample type
}  // namespace

struct msdr_comm {
    msdr_ctx *ctx;
    RcclApi *api;
    void *comm;
    int rank, world;
    hipStream_t stream;                  // the gathers' own stream: they overlap the next block's kernels on the context's stream
    hipEvent_t ready[MSDR_GATHER_SLOTS], done[MSDR_GATHER_SLOTS];
    bool pending[MSDR_GATHER_SLOTS];
};
#define RCCL_TRY(call) do { int r_ = (call); if (r_ != 0) return fail(MSDR_STATUS_HIP_ERROR, "%s: %s", #call, C->api->GetErrorString(r_)); } while (0)

extern "C" int msdr_comm_get_unique_id(void *id128)
{
    RcclApi *api = rccl_api();
    if (!api) return fail(MSDR_STATUS_NO_DEVICE, "librccl.so.1 could not be loaded");
    if (!id128) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null id");
    RcclUniqueId id;
    if (int r = api->GetUniqueId(&id)) return fail(MSDR_STATUS_HIP_ERROR, "ncclGetUniqueId: %s", api->GetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return 0;
}
extern "C" int msdr_comm_create(msdr_ctx *ctx, const void *id128, int rank, int world, msdr_comm **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (int rc = bind(ctx)) return rc;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad communicator arguments");
    RcclApi *api = rccl_api();
    if (!api) return fail(MSDR_STATUS_NO_DEVICE, "librccl.so.1 could not be loaded");
    msdr_comm *C = new (std::nothrow) msdr_comm();
    if (!C) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    C->ctx = ctx; C->api = api; C->comm = nullptr; C->rank = rank; C->world = world; C->stream = nullptr;
    for (int k = 0; k < MSDR_GATHER_SLOTS; k++) { C->ready[k] = C->done[k] = nullptr; C->pending[k] = false; }
    RcclUniqueId id;
    memcpy(&id, id128, sizeof id);
    if (int r = api->CommInitRank(&C->comm, world, id, rank)) { delete C; return fail(MSDR_STATUS_HIP_ERROR, "ncclCommInitRank: %s", api->GetErrorString(r)); }
    hipError_t e = hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking);
    for (int k = 0; k < MSDR_GATHER_SLOTS && e == hipSuccess; k++) {
        e = hipEventCreateWithFlags(&C->ready[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&C->done[k], hipEventDisableTiming);
    }
    if (e != hipSuccess) { msdr_comm_destroy(C); return fail(MSDR_STATUS_HIP_ERROR, "communicator stream / events: %s", hipGetErrorString(e)); }
    *out = C;
    return 0;
}
extern "C" int msdr_gather_audio_begin(msdr_comm *C, int slot, const void *d_local, size_t local_bytes, void *d_recv, int root)
{
    if (!C || slot < 0 || slot >= MSDR_GATHER_SLOTS) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad communicator / slot");
    if (int rc = bind(C->ctx)) return rc;
    if (root >= C->world) return fail(MSDR_STATUS_ARGUMENT_ERROR, "root out of range");
    const bool receives = root < 0 || root == C->rank;
    if (!d_local || (receives && !d_recv)) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    // the audio of this block is whatever the context's stream has queued so far
    HIP_TRY(hipEventRecord(C->ready[slot], C->ctx->stream));
    HIP_TRY(hipStreamWaitEvent(C->stream, C->ready[slot], 0));
    if (local_bytes) {
        if (root < 0) RCCL_TRY(C->api->AllGather(d_local, d_recv, local_bytes, kRcclChar, C->comm, C->stream));
        else {
            // gather to one rank: every peer sends its shard over its own link, the root posts one receive per peer (one group)
            RCCL_TRY(C->api->GroupStart());
            int r = 0;
            if (C->rank == root) {
                for (int peer = 0; peer < C->world && r == 0; peer++)
                    if (peer != root) r = C->api->Recv((char *)d_recv + (size_t)peer * local_bytes, local_bytes, kRcclChar, peer, C->comm, C->stream);
            } else r = C->api->Send(d_local, local_bytes, kRcclChar, root, C->comm, C->stream);
            const int r2 = C->api->GroupEnd();
            if (r || r2) return fail(MSDR_STATUS_HIP_ERROR, "RCCL gather: %s", C->api->GetErrorString(r ? r : r2));
            if (C->rank == root) HIP_TRY(hipMemcpyAsync((char *)d_recv + (size_t)root * local_bytes, d_local, local_bytes, hipMemcpyDeviceToDevice, C->stream));
        }
    }
    HIP_TRY(hipEventRecord(C->done[slot], C->stream));
    C->pending[slot] = true;
    return 0;
}
extern "C" int msdr_gather_audio_wait(msdr_comm *C, int slot, int host_wait)
{
    if (!C || slot < 0 || slot >= MSDR_GATHER_SLOTS) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad communicator / slot");
    if (int rc = bind(C->ctx)) return rc;
    if (!C->pending[slot]) return 0;
    if (host_wait) HIP_TRY(hipEventSynchronize(C->done[slot]));
    else HIP_TRY(hipStreamWaitEvent(C->ctx->stream, C->done[slot], 0));     // later work on the context's stream may reuse the buffers
    C->pending[slot] = false;
    return 0;
}
extern "C" int msdr_comm_destroy(msdr_comm *C)
{
    if (!C) return 0;
    (void)bind(C->ctx);
    if (C->stream) (void)hipStreamSynchronize(C->stream);
    for (int k = 0; k < MSDR_GATHER_SLOTS; k++) { if (C->ready[k]) hipEventDestroy(C->ready[k]); if (C->done[k]) hipEventDestroy(C->done[k]); }
    if (C->comm) (void)C->api->CommDestroy(C->comm);
    if (C->stream) (void)hipStreamDestroy(C->stream);
    delete C;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// designers (host)
// ------------------------------------------------------------------------------------------------
extern "C" void msdr_calc_FIR_coeffs(int16_t *coeffs, int numCoeffs, float32_t fc, float32_t Astop, int type, float dfc,
                                     float Fsamprate)
{
    msdr::design::calc_fir_coeffs(coeffs, numCoeffs, fc, Astop, type, dfc, Fsamprate, false);
}
extern "C" void msdr_calc_FIR_coeffs_pid(int16_t *coeffs, int numCoeffs, float32_t fc, float32_t Astop, int type, float dfc,
                                         float Fsamprate)
{
    msdr::design::calc_fir_coeffs(coeffs, numCoeffs, fc, Astop, type, dfc, Fsamprate, true);
}
extern "C" int msdr_biquad_design(int kind, float frequency, float q_or_gain, float slope, double sample_rate, int32_t coef[5])
{
    if (!coef || kind < MSDR_BQ_LOWPASS || kind > MSDR_BQ_HIGHSHELF) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad biquad kind");
    msdr::design::biquad_design(kind, frequency, q_or_gain, slope, sample_rate, coef);
    return 0;
}

// tables for the parallel df1 cascade (see BiquadCascadeTables in msdr_shared.h)
template <int L>
static void make_cascade_tables(const float *coeffs, int stages, BiquadCascadeTables<L> *T)
{
    memset(T, 0, sizeof *T);
    T->nstages = stages;
    // combined numerator C(z) = prod_s (b0 + b1 z^-1 + b2 z^-2)
    std::vector<double> num(1, 1.0);
    for (int s = 0; s < stages; s++) {
        std::vector<double> nx(num.size() + 2, 0.0);
        for (size_t i = 0; i < num.size(); i++)
            for (int k = 0; k < 3; k++) nx[i + k] += num[i] * (double)coeffs[5 * s + k];
        num.swap(nx);
    }
    if (stages == 0) num.assign(1, 1.0);
    for (size_t i = 0; i < num.size() && i < 12; i++) T->num[i] = (float)num[i];
    auto mul = [](const double *A, const double *B, double *C) {
        double r[4] = {A[0] * B[0] + A[1] * B[2], A[0] * B[1] + A[1] * B[3], A[2] * B[0] + A[3] * B[2], A[2] * B[1] + A[3] * B[3]};
        memcpy(C, r, sizeof r);
    };
    for (int s = 0; s < stages; s++) {
        auto &Q = T->sec[s];
        Q.a1 = coeffs[5 * s + 3]; Q.a2 = coeffs[5 * s + 4];
        const double a1 = Q.a1, a2 = Q.a2;
        // impulse response g and homogeneous responses alpha/beta of  y[n] = a1 y[n-1] + a2 y[n-2]
        double g[L + 1], gm1 = 1.0, gm2 = 0.0;      // g[0] = 1
        g[0] = 1.0;
        for (int k = 1; k <= L; k++) { g[k] = a1 * gm1 + a2 * gm2; gm2 = gm1; gm1 = g[k]; }
        for (int j = 0; j < L; j++) {
            Q.g[j][0] = (float)g[L - 1 - j];
            Q.g[j][1] = (L - 2 - j >= 0) ? (float)g[L - 2 - j] : 0.0f;
        }
        double am1 = 1, am2 = 0, bm1 = 0, bm2 = 1, al[L], be[L];
        for (int j = 0; j < L; j++) {
            al[j] = a1 * am1 + a2 * am2; be[j] = a1 * bm1 + a2 * bm2;
            am2 = am1; am1 = al[j]; bm2 = bm1; bm1 = be[j];
        }
        double M[4] = {al[L - 1], be[L - 1], al[L - 2], be[L - 2]};      // row-major
        double P[4];
        memcpy(P, M, sizeof P);
        for (int k = 0; k < 6; k++) {
            if (k < 4) {
                Q.mcol[k][0][0] = (float)P[0]; Q.mcol[k][0][1] = (float)P[2];   // column 0 = (M00, M10)
                Q.mcol[k][1][0] = (float)P[1]; Q.mcol[k][1][1] = (float)P[3];   // column 1 = (M01, M11)
            }
            mul(P, P, P);
        }
        for (int i = 0; i < 4; i++) Q.m64[i] = (float)P[i];
        double W[4];
        memcpy(W, M, sizeof W);
        for (int l = 0; l < 64; l++) {
            Q.mlane[l][0] = (float)W[0]; Q.mlane[l][1] = (float)W[2]; Q.mlane[l][2] = (float)W[1]; Q.mlane[l][3] = (float)W[3];
            mul(W, M, W);
        }
    }
}

// largest pole radius of the cascade (y = ... + a1 y1 + a2 y2  =>  z^2 - a1 z - a2)
static double max_pole_radius(const float *coeffs, int stages)
{
    double r = 0;
    for (int s = 0; s < stages; s++) {
        std::complex<double> a1 = coeffs[5 * s + 3], a2 = coeffs[5 * s + 4];
        std::complex<double> disc = std::sqrt(a1 * a1 + 4.0 * a2);
        r = std::max(r, std::max(std::abs((a1 + disc) / 2.0), std::abs((a1 - disc) / 2.0)));
    }
    return r;
}

// Conditioning of the "numerators first, all-pole sections afterwards" evaluation of a df1 cascade (msdr_biquad.hiph): kappa =
// ||c||_1 ||g||_1 / ||h||_1 with c = product of the numerators, g = impulse response of the all-pole cascade, h = c * g the whole
// filter's.  ~1 for low-pass sections; thousands when zeros and poles crowd the same spot (high-pass, narrow notches), and then the
// parallel evaluation's error grows like kappa x 2.5e-8 on real signals (measured: LP + Q15 notch, kappa 19: 5e-7; one 300 Hz
// high-pass section, kappa 300: 8e-6; two of them, kappa 1.7e5: 8e-3; tests/debug/iir_conditioning.py).  Above the limit the cascade runs
// in CMSIS order (biquad_df1_seq_kernel).
static double cascade_condition(const float *coeffs, int stages)
{
    if (stages <= 0) return 1.0;
    std::vector<double> c(1, 1.0);
    for (int s = 0; s < stages; s++) {
        std::vector<double> nx(c.size() + 2, 0.0);
        for (size_t i = 0; i < c.size(); i++)
            for (int k = 0; k < 3; k++) nx[i + k] += c[i] * (double)coeffs[5 * s + k];
        c.swap(nx);
    }
    const double r = max_pole_radius(coeffs, stages);
    if (!(r < 0.99999)) return 1e30;
    int len = r > 0 ? (int)std::min(200000.0, std::ceil(std::log(1e-12) / std::log(r)) + 64.0 * stages) : 16;
    std::vector<double> g(len, 0.0);
    g[0] = 1.0;
    for (int s = 0; s < stages; s++) {                        // all-pole sections in series
        const double a1 = coeffs[5 * s + 3], a2 = coeffs[5 * s + 4];
        double y1 = 0, y2 = 0;
        for (int k = 0; k < len; k++) { const double yv = g[k] + a1 * y1 + a2 * y2; y2 = y1; y1 = yv; g[k] = yv; }
    }
    double c1 = 0, g1 = 0, h1 = 0;
    for (double v : c) c1 += std::fabs(v);
    for (double v : g) g1 += std::fabs(v);
    for (int k = 0; k < len; k++) {
        double hv = 0;
        for (size_t i = 0; i < c.size() && (int)i <= k; i++) hv += c[i] * g[k - i];
        h1 += std::fabs(hv);
    }
    return h1 > 0 ? c1 * g1 / h1 : 1e30;
}
// The limit: judged against a float64 evaluation (tests/debug/fuzz_f32_truth.py, random 1-4 section LP / HP / notch cascades) the
// parallel form between kappa 30 and 100 was 1e-5 ... 9e-5 from the truth where the sequential fp32 order is at 1e-6; below 30 the two
// are within a small factor of each other.  The reference's own cascade (LP Q 0.54 + notch Q 15) has kappa 19.
constexpr double kCascadeConditionLimit = 30.0;

// How noisy the cascade is in fp32 whatever the order of evaluation: the distance of the sequential df1 evaluation (CMSIS order,
// separately rounded products and sums) from a double evaluation on a fixed test signal (white noise on a DC term, 8192 samples).
// 4e-7 for the reference's cascade; 1e-5 and more when narrow resonances ring for hundreds of samples.  The block-parallel
// solver's dot products with the all-pole impulse response lose kappa times that (measured, tests/debug/fuzz_stage_df1.py:
// kappa 20 x 1.4e-5 -> 4.5e-4), so the product of the two is what decides -- see cascade_needs_cmsis_order().
static double cascade_fp32_noise(const float *coeffs, int stages)
{
    const int N = 8192;
    float sf[kMaxStages][4] = {{0}};
    double sd[kMaxStages][4] = {{0}};
    double num = 0.0, den = 0.0;
    uint32_t lcg = 12345u;
    for (int n = 0; n < N; n++) {
        double u = 0.0;
        for (int k = 0; k < 4; k++) { lcg = lcg * 1664525u + 1013904223u; u += (double)(lcg >> 8) * (1.0 / 16777216.0); }
        const float xin = (float)(1.0 + (u - 2.0) * 1.7320508);          // unit variance (sum of four uniforms), mean 1
        volatile float df = xin;                                          // volatile: every product and sum rounded to fp32, no contraction
        double dd = (double)xin;
        for (int s = 0; s < stages; s++) {
            const float *c = coeffs + 5 * s;
            volatile float p0 = c[0] * df, p1 = c[1] * sf[s][0], p2 = c[2] * sf[s][1], p3 = c[3] * sf[s][2], p4 = c[4] * sf[s][3];
            volatile float a = p0 + p1; a = a + p2; a = a + p3; a = a + p4;
            sf[s][1] = sf[s][0]; sf[s][0] = df; sf[s][3] = sf[s][2]; sf[s][2] = a;
            df = a;
            const double yd = (double)c[0] * dd + (double)c[1] * sd[s][0] + (double)c[2] * sd[s][1] + (double)c[3] * sd[s][2] + (double)c[4] * sd[s][3];
            sd[s][1] = sd[s][0]; sd[s][0] = dd; sd[s][3] = sd[s][2]; sd[s][2] = yd;
            dd = yd;
        }
        if (!std::isfinite(dd) || !std::isfinite((double)df)) return 1e30;
        num += ((double)df - dd) * ((double)df - dd); den += dd * dd;
    }
    return den > 0 ? std::sqrt(num / den) : 0.0;
}
// What the block-parallel evaluation costs by itself: the kernels' algorithm at small scale on the host, every product and sum rounded to
// fp32 -- the combined numerator C(z) = prod B_s as one FIR, then every all-pole section solved in blocks of 16 samples (zero-state end
// state of a block as two dot products with the section's impulse response, block entry states by the affine recurrence with M^16, the
// recursion inside the block from its entry state) -- against a double evaluation, on the same test signal as cascade_fp32_noise.
// For the reference's cascade (and the test suites' 3- and 4-section ones) it equals the sequential order's noise (3.2e-7 ... 4.4e-7).
// Round 4's fuzzers found cascades every older criterion passed (kappa 3 ... 18, sequential noise 3e-7: three or four resonant sections,
// pole radii 0.95 ... 0.997; fuzz_f32_truth seed 4106 case 46411, fuzz_stage_df1 seed 4107) on which the kernels were 1.5e-5 ... 4.6e-5
// from float64, 30 ... 120 x the oracle's distance; this emulation reads 1.1e-6 ... 1.8e-5 on them, 3 ... 48 x the sequential figure,
// and 0.6 ... 1.4 x on the cascades the kernels handle well.  The contract (include/msdr.h) allows 2 x: such cascades run in CMSIS order.
static double cascade_parallel_noise(const float *coeffs, int stages)
{
    const int N = 8192, L = 16, NB = N / L, NC = 2 * stages + 1;
    std::vector<double> cn(1, 1.0);
    for (int s = 0; s < stages; s++) {
        std::vector<double> nx(cn.size() + 2, 0.0);
        for (size_t i = 0; i < cn.size(); i++)
            for (int k = 0; k < 3; k++) nx[i + k] += cn[i] * (double)coeffs[5 * s + k];
        cn.swap(nx);
    }
    std::vector<float> x(N), u(N), y(N);
    std::vector<double> ref(N);
    {
        uint32_t lcg = 12345u;
        for (int n = 0; n < N; n++) {
            double t = 0.0;
            for (int k = 0; k < 4; k++) { lcg = lcg * 1664525u + 1013904223u; t += (double)(lcg >> 8) * (1.0 / 16777216.0); }
            x[n] = (float)(1.0 + (t - 2.0) * 1.7320508);
        }
        double sd[kMaxStages][4] = {{0}};
        for (int n = 0; n < N; n++) {
            double dd = (double)x[n];
            for (int s = 0; s < stages; s++) {
                const float *c = coeffs + 5 * s;
                const double yd = (double)c[0] * dd + (double)c[1] * sd[s][0] + (double)c[2] * sd[s][1] + (double)c[3] * sd[s][2] + (double)c[4] * sd[s][3];
                sd[s][1] = sd[s][0]; sd[s][0] = dd; sd[s][3] = sd[s][2]; sd[s][2] = yd;
                dd = yd;
            }
            ref[n] = dd;
        }
    }
    for (int n = 0; n < N; n++) {                                   // the numerator FIR
        volatile float v = 0.0f;
        for (int k = 0; k < NC && k <= n; k++) { volatile float pr = (float)cn[k] * x[n - k]; v = v + pr; }
        u[n] = v;
    }
    for (int s = 0; s < stages; s++) {
        const float a1 = coeffs[5 * s + 3], a2 = coeffs[5 * s + 4];
        double g[L + 1], gm1 = 1.0, gm2 = 0.0, al[L], be[L], am1 = 1, am2 = 0, bm1 = 0, bm2 = 1;
        g[0] = 1.0;
        for (int k = 1; k <= L; k++) { g[k] = (double)a1 * gm1 + (double)a2 * gm2; gm2 = gm1; gm1 = g[k]; }
        for (int j = 0; j < L; j++) { al[j] = (double)a1 * am1 + (double)a2 * am2; be[j] = (double)a1 * bm1 + (double)a2 * bm2; am2 = am1; am1 = al[j]; bm2 = bm1; bm1 = be[j]; }
        const float m00 = (float)al[L - 1], m01 = (float)be[L - 1], m10 = (float)al[L - 2], m11 = (float)be[L - 2];
        float e0 = 0.0f, e1 = 0.0f;                                 // state entering the block: (w[-1], w[-2])
        for (int b = 0; b < NB; b++) {
            const float *ub = u.data() + b * L;
            volatile float z0 = 0.0f, z1 = 0.0f;                   // zero-state (w[L-1], w[L-2]) of the block
            for (int j = 0; j < L; j++) {
                volatile float p0 = (float)g[L - 1 - j] * ub[j]; z0 = z0 + p0;
                if (j < L - 1) { volatile float p1 = (float)g[L - 2 - j] * ub[j]; z1 = z1 + p1; }
            }
            volatile float y1 = e0, y2 = e1;                       // the recursion inside the block, from its entry state
            for (int j = 0; j < L; j++) {
                volatile float q1 = a1 * y1, q2 = a2 * y2;
                volatile float w = ub[j] + q1; w = w + q2;
                y[b * L + j] = w; y2 = y1; y1 = w;
            }
            volatile float t0 = m00 * e0, t1 = m01 * e1, t2 = m10 * e0, t3 = m11 * e1;      // the next block's entry state by the affine recurrence
            volatile float n0 = z0 + t0; n0 = n0 + t1;
            volatile float n1 = z1 + t2; n1 = n1 + t3;
            e0 = n0; e1 = n1;
        }
        u.swap(y);
    }
    double num = 0.0, den = 0.0;
    for (int n = 0; n < N; n++) {
        if (!std::isfinite((double)u[n]) || !std::isfinite(ref[n])) return 1e30;
        num += ((double)u[n] - ref[n]) * ((double)u[n] - ref[n]); den += ref[n] * ref[n];
    }
    return den > 0 ? std::sqrt(num / den) : 0.0;
}
constexpr double kCascadeParallelErrorLimit = 2e-5;     // kappa x fp32 noise; the reference's cascade: 19 x 4e-7 = 8e-6
// A cascade that is this noisy in the sequential order already (5 x the reference's) leaves no room: the parallel solver was 2-10 x
// the oracle's distance from float64 on such filters (resonant sections below 1 kHz), whatever kappa says.
constexpr double kCascadeNoiseLimit = 2e-6;

static bool cascade_needs_cmsis_order(const float *coeffs, int stages)
{
    if (stages <= 0) return false;
    const double kappa = cascade_condition(coeffs, stages);
    const double noise = cascade_fp32_noise(coeffs, stages);
    // three and four sections: the stage fuzz still found 2e-5 ... 6e-5 between kappa 20 and 30 (the test cascades of that size: 13-14)
    const double klimit = stages >= 3 ? 20.0 : kCascadeConditionLimit;
    bool seq = kappa > klimit || kappa * noise > kCascadeParallelErrorLimit || noise > kCascadeNoiseLimit;
    // ... and the block-parallel evaluation itself must not cost more than half of what the contract allows over the sequential order
    if (!seq && cascade_parallel_noise(coeffs, stages) > 1.5 * noise + 2e-7) seq = true;
    return seq;
}

// ------------------------------------------------------------------------------------------------
// Q15 on the integer matrix cores (msdr_chain_q15mf.hiph): byte-split Toeplitz fragments per (tap set, phase mod 4).
// par[o] = parity of the mixer phases that feed accumulator o (I, Q); fir_only: one filter over every sample (the FIR stage).
// ------------------------------------------------------------------------------------------------
struct QmTables { std::vector<char> blob; int stride = 0, halo = 0, bsteps = 0; std::vector<char> set_ok; };
// fr: full-rate layout -- accumulator o runs over stream o (I, Q) at every sample, the oscillator is not in the tables.
static void qm_build_tables(int N, uint32_t tapsets, const int16_t *const *coef_i, const int16_t *const *coef_q, const int par[2], bool fir_only, QmTables &out, bool fr = false)
{
    const int H = qm_halo(N), NE = fr ? H + 32 : (H + 32) / 2, NC = (NE + 31) / 32;      // source-array elements / 32-element chunks
    const int SP = fr ? 1 : 2;                                                           // window samples per array element
    struct QTab { Q15MfHeader h; std::vector<int8_t> frags; };
    std::vector<QTab> tabs((size_t)tapsets * 4);
    out.set_ok.assign(tapsets, 1);
    int bsteps = 0;
    for (uint32_t s = 0; s < tapsets; s++) {
        const int16_t *cf[2] = {coef_i[s], fir_only ? coef_i[s] : coef_q[s]};
        for (int o = 0; o < (fir_only ? 1 : 2); o++)
            for (int k = 0; k < N; k++) if (cf[o][k] >= 32640) out.set_ok[s] = 0;          // ch = (c - (int8)c) >> 8 would be 128
        for (int rot = 0; rot < 4; rot++) {
            QTab &T = tabs[(size_t)s * 4 + rot];
            memset(&T.h, 0, sizeof T.h);
            T.h.ok = out.set_ok[s];
            if (!T.h.ok) continue;
            int total = 0;
            for (int o = 0; o < (fir_only ? 1 : 2); o++) {
                unsigned bsum[2] = {0u, 0u};
                for (int r = 0; r < (fir_only ? 2 : 1); r++) {
                    // accumulator o is fed by the window samples i whose mixer phase (rot + i) mod 4 has parity par[o]; the FIR stage
                    // takes both parities as two runs of the same accumulator
                    const int src = fr ? o : fir_only ? r : ((par[o] + 4 - rot) & 1);
                    // B[e][b] = tap at delay H + b - (2 e + src) (CMSIS keeps the taps time-reversed: delay d is pCoeffs[N - 1 - d]);
                    // full rate: element e is window sample e
                    auto Bv = [&](int e, int b) -> int { const int d = H + b - (fr ? e : 2 * e + src); return (d >= 0 && d < N && e < NE) ? (int)cf[o][N - 1 - d] : 0; };
                    int jlo = NC, jhi = -1;
                    for (int j = 0; j < NC; j++) {
                        bool any = false;
                        for (int e = 32 * j; e < 32 * j + 32 && !any; e++)
                            for (int b = 0; b < 32 && !any; b++) any = Bv(e, b) != 0;
                        if (any) { jlo = std::min(jlo, j); jhi = std::max(jhi, j); }
                    }
                    if (jhi < 0) { jlo = 0; jhi = 0; }                                   // an all-zero filter: one chunk of zeros (the kernel's first k-step is unconditional)
                    T.h.run[o][r].src = src; T.h.run[o][r].j0 = jlo; T.h.run[o][r].cnt = jhi - jlo + 1;
                    total += T.h.run[o][r].cnt;
                    for (int pb = 0; pb < 2; pb++)
                        for (int e = 0; e < NE; e++) bsum[pb] += (unsigned)Bv(e, pb);
                    for (int j = jlo; j <= jhi; j++) {
                        const size_t base = T.frags.size();
                        T.frags.resize(base + 2048);                                  // ch piece [64 lanes][16], then cl piece
                        for (int l = 0; l < 64; l++)
                            for (int jj = 0; jj < 16; jj++) {
                                const int v = Bv(32 * j + 16 * (l >> 5) + jj, l & 31);
                                const int lo = (int)(int8_t)(v & 0xFF), hi = (v - lo) >> 8;
                                T.frags[base + l * 16 + jj] = (int8_t)hi;
                                T.frags[base + 1024 + l * 16 + jj] = (int8_t)lo;
                            }
                    }
                }
                T.h.bias[o][0] = (int)(128u * bsum[0]); T.h.bias[o][1] = (int)(128u * bsum[1]);
            }
            bsteps = std::max(bsteps, total);
        }
    }
    (void)SP;
    if (bsteps > 0 && qm_lds_bytes(H, bsteps, 1, fr) <= 160 * 1024) {
        const int stride = kQmHdrBytes + bsteps * 2048;
        out.blob.assign((size_t)stride * tabs.size(), 0);
        for (size_t t = 0; t < tabs.size(); t++) {
            memcpy(out.blob.data() + t * stride, &tabs[t].h, sizeof(Q15MfHeader));
            if (!tabs[t].frags.empty()) memcpy(out.blob.data() + t * stride + kQmHdrBytes, tabs[t].frags.data(), tabs[t].frags.size());
        }
        out.stride = stride; out.halo = H; out.bsteps = bsteps;
    }
}

// ------------------------------------------------------------------------------------------------
// single-stream FIR instances (arm_fir_fast_q15 / arm_fir_f32 mirrors)
// ------------------------------------------------------------------------------------------------
template <typename In, typename El>
struct FirInst {
    msdr_ctx *ctx;
    uint32_t channels, ntaps, ntaps_pad, hist_len;
    El *d_taps;
    In *d_hist[2];
    int cur;
};
struct msdr_fir_q15 : FirInst<int16_t, int32_t> {
    // matrix-core path (chain_q15mf_kernel<3>): tables, channel list (identity), or null when a tap does not split into signed bytes
    char *d_qm_tab = nullptr;
    int *d_qm_order = nullptr;
    int qm_stride = 0, qm_halo = 0, qm_bsteps = 0;
    // block cadence (chain_q15mb_kernel<3>: calls of 32 .. 512 samples, the reference's 128 among them): the tile table of the last block
    // length seen (identity order: one tap set), waves per workgroup, tiles per wave
    int *d_btiles = nullptr;
    size_t btiles_cap = 0;
    uint32_t bt_n = 0, bt_wgs = 0, bt_nw = 0, bt_tpw = 0;
};
struct msdr_fir_f32 : FirInst<float, float> {
    // matrix-core path (msdr_fir_f32mf.hiph): header + split-fp16 Toeplitz fragments, or null (then fir_kernel<FirF32> runs)
    char *d_fm_tab = nullptr;
    int fm_halo_ = 0, fm_bsteps = 0, fm_ex = 0;
    // taps-in-registers path (msdr_fir_f32tr.hiph, <= ~290 taps): header + the two families of A fragments, or null
    char *d_tr_tab = nullptr;
    int tr_ns = 0, tr_skip1 = 0;
    // tile queue of fir_f32tq_kernel (msdr_fir_f32tq.hiph): two sets of front counters that alternate between launches, or null
    unsigned *d_tq_ctr = nullptr;
    int tq_fronts = 32;     // 32 fronts: 24 ... 64 within 1 %, 16 and fewer saturate the counters (profiles/r03/fir_ab.txt)
    // tiles per draw = 2^shift (round 4): the second .. last tile of a run take their 1 KB halo from the registers that hold the tile before.
    // Runs of 4 / 8: HBM traffic / algorithmic 1.049 -> 1.037 / ~1.02 (the halos that are still fetched all miss L2), a fifth of the loads gone;
    // time -0.5 % only (the kernel sits at the power cap: the clock rises by what the traffic saves), runs of 2 .. 8 alike
    // (profiles/r04/README.md).  MSDR_TQ_RUN_SHIFT overrides (tests: 0 .. 3).
    int tq_run_shift_cap = getenv("MSDR_TQ_RUN_SHIFT") ? std::max(0, std::min(6, atoi(getenv("MSDR_TQ_RUN_SHIFT")))) : 3;
    int last_launch = 0;                 // what the last msdr_fir_f32_process launched: 0 none yet, 1 tile queue, 2 fir_f32mf_kernel (the name follows the size guard)
    bool tq_run_forced = getenv("MSDR_TQ_RUN_SHIFT") != nullptr;       // (the override also lifts the "every wave gets >= 16 runs" rule: small test shapes reach the run code)
    float input_range = 0.0f;            // 0 = block floating point per tile (default); > 0: a fixed scale for samples below this magnitude
};

template <typename Inst, typename In, typename El>
static int fir_create(msdr_ctx *ctx, uint16_t numTaps, const In *pCoeffs, uint32_t channels, Inst **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (int rc = bind(ctx)) return rc;
    if (!pCoeffs || numTaps == 0 || channels == 0) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad FIR arguments");
    Inst *S = new (std::nothrow) Inst();
    if (!S) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    S->ctx = ctx; S->channels = channels; S->ntaps = numTaps;
    S->ntaps_pad = (numTaps + 3u) & ~3u;
    S->hist_len = S->ntaps_pad - 1;
    S->cur = 0; S->d_taps = nullptr; S->d_hist[0] = S->d_hist[1] = nullptr;
    std::vector<El> t(S->ntaps_pad, (El)0);                    // zero taps in FRONT: they meet older samples
    for (uint32_t k = 0; k < numTaps; k++) t[S->ntaps_pad - numTaps + k] = (El)pCoeffs[k];
    int rc = upload(ctx, t, &S->d_taps);
    if (!rc) rc = dzalloc(ctx, (size_t)channels * S->hist_len, &S->d_hist[0]);
    if (!rc) rc = dzalloc(ctx, (size_t)channels * S->hist_len, &S->d_hist[1]);
    if (rc) { hipFree(S->d_taps); hipFree(S->d_hist[0]); hipFree(S->d_hist[1]); delete S; return rc; }
    *out = S;
    return 0;
}

template <typename F, typename Inst, typename In>
static int fir_process(Inst *S, const In *d_src, In *d_dst, uint32_t blockSize)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(S->ctx)) return rc;
    if (blockSize == 0) return 0;
    if (!d_src || !d_dst) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    if ((const void *)d_src == (const void *)d_dst)
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "FIR process is not in-place (the reference uses separate buffers, Minimal-SDR.ino:574-578)");
    FirParams p;
    p.x = d_src; p.y = d_dst; p.hist_in = S->d_hist[S->cur];
    p.n = blockSize; p.channels = (int)S->channels;
    // time segments: enough workgroups to fill the chip, no segment shorter than 4 tiles
    long long tiles = ((long long)blockSize + kFirTile - 1) / kFirTile;
    long long want = std::max<long long>(1, (2048 + S->channels - 1) / S->channels);
    long long nseg = std::max<long long>(1, std::min<long long>(want, tiles / 4));
    long long seg_tiles = (tiles + nseg - 1) / nseg;
    p.seg_len = seg_tiles * kFirTile;
    p.nseg = (int)((tiles + seg_tiles - 1) / seg_tiles);
    p.ntaps_pad = (int)S->ntaps_pad; p.hist_len = (int)S->hist_len; p.taps = S->d_taps;
    { KernelTimer kt(S->ctx); hipLaunchKernelGGL((fir_kernel<F>), dim3(S->channels * p.nseg), dim3(kThreads), fir_lds_bytes(p.ntaps_pad), S->ctx->stream, p); }
    if (int rc = launch_check("fir_kernel")) return rc;
    hipLaunchKernelGGL((history_kernel<In>), dim3(grid_1d((long long)S->channels * S->hist_len)), dim3(256), 0, S->ctx->stream,
                       d_src, (const In *)S->d_hist[S->cur], S->d_hist[S->cur ^ 1], (long long)blockSize, (int)S->hist_len,
                       (int)S->channels);
    if (int rc = launch_check("history_kernel")) return rc;
    S->cur ^= 1;
    return 0;
}

template <typename Inst>
static int fir_reset(Inst *S)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(S->ctx)) return rc;
    size_t bytes = (size_t)S->channels * S->hist_len * sizeof(*S->d_hist[0]);
    HIP_TRY(hipMemsetAsync(S->d_hist[0], 0, bytes, S->ctx->stream));
    HIP_TRY(hipMemsetAsync(S->d_hist[1], 0, bytes, S->ctx->stream));
    return 0;
}
template <typename Inst>
static int fir_destroy(Inst *S)
{
    if (!S) return 0;
    if (int rc = bind(S->ctx)) return rc;
    (void)hipStreamSynchronize(S->ctx->stream);
    hipFree(S->d_taps); hipFree(S->d_hist[0]); hipFree(S->d_hist[1]);
    delete S;
    return 0;
}

extern "C" int msdr_fir_q15_create(msdr_ctx *ctx, uint16_t numTaps, const q15_t *pCoeffs, uint32_t channels, msdr_fir_q15 **out)
{
    if (out) *out = nullptr;
    if (numTaps & 1u)   // arm_fir_init_q15.c:93-96
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "arm_fir_init_q15: numTaps must be even (got %u)", (unsigned)numTaps);
    if (int rc = fir_create<msdr_fir_q15, int16_t, int32_t>(ctx, numTaps, pCoeffs, channels, out)) return rc;
    msdr_fir_q15 *S = *out;
    if (qm_halo((int)numTaps) <= 512) {       // the same filter on the integer matrix cores (msdr_chain_q15mf.hiph)
        const int16_t *ci[1] = {pCoeffs};
        const int par[2] = {0, 1};
        QmTables T;
        qm_build_tables((int)numTaps, 1, ci, ci, par, true, T);
        if (!T.blob.empty() && T.set_ok[0]) {
            std::vector<int> order(channels);
            for (uint32_t i = 0; i < channels; i++) order[i] = (int)i;
            int rc = upload(ctx, T.blob, &S->d_qm_tab);
            if (!rc) rc = upload(ctx, order, &S->d_qm_order);
            if (rc) { msdr_fir_q15_destroy(S); *out = nullptr; return rc; }
            S->qm_stride = T.stride; S->qm_halo = T.halo; S->qm_bsteps = T.bsteps;
            // the history in the matrix-core window's length (the block kernel stages whole windows of qm_halo + n samples and writes the next
            // history itself; the long-call kernel takes any length >= numTaps - 1)
            if ((uint32_t)T.halo > S->hist_len) {
                hipFree(S->d_hist[0]); hipFree(S->d_hist[1]); S->d_hist[0] = S->d_hist[1] = nullptr;
                S->hist_len = (uint32_t)T.halo;
                rc = dzalloc(ctx, (size_t)channels * S->hist_len, &S->d_hist[0]);
                if (!rc) rc = dzalloc(ctx, (size_t)channels * S->hist_len, &S->d_hist[1]);
                if (rc) { msdr_fir_q15_destroy(S); *out = nullptr; return rc; }
            }
        }
    }
    return 0;
}
// the block-cadence tile table of a FIR stage: every channel on the one tap set, in channel order (the rules of chain_block_tiles)
static int fir_q15_block_tiles(msdr_fir_q15 *S, int n)
{
    if (S->bt_n == (uint32_t)n && S->d_btiles) return 0;
    const int cpt = mb_cpt(n);
    int fill = cpt;
    auto tiles_at = [&](int f) { return ((long long)S->channels + f - 1) / f; };
    while (fill > 1 && tiles_at(fill) * 2 <= (long long)S->ctx->num_cus * 4) fill >>= 1;
    const long long tiles = tiles_at(fill);
    long long best_cost = -1;
    uint32_t nw = 1, tpw = 1;
    for (int w = 1; w <= 8; w++) {
        long long t = (tiles + (long long)S->ctx->num_cus * w - 1) / ((long long)S->ctx->num_cus * w);
        t = std::max<long long>(1, std::min<long long>(t, kMbMaxTableInts / cpt));
        if (qb_lds_bytes(S->qm_halo, n, S->qm_bsteps, w, (int)t) > 160 * 1024) break;
        const long long cost = t * (w <= 4 ? 2 : 3);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; nw = (uint32_t)w; tpw = (uint32_t)t; }
    }
    if (qb_lds_bytes(S->qm_halo, n, S->qm_bsteps, (int)nw, (int)tpw) > 160 * 1024) return -1;      // not even one wave with one tile: the caller takes the long-call kernel
    const size_t per_wg = (size_t)nw * tpw;
    std::vector<int> tab;
    uint32_t wgs = 0;
    for (size_t t0 = 0; t0 < (size_t)tiles; t0 += per_wg, wgs++) {
        tab.push_back(0); tab.push_back(0); tab.push_back(0); tab.push_back(0);       // kMbRecHdrInts: {tap set 0, ...}
        const size_t base = tab.size();
        tab.resize(base + per_wg * cpt, -1);
        const size_t cnt = std::min(per_wg, (size_t)tiles - t0);
        for (size_t t = 0; t < cnt; t++) {
            const size_t slot = (t % nw) * tpw + t / nw;
            for (int k = 0; k < fill; k++) {
                const size_t idx = (t0 + t) * fill + k;
                if (idx < S->channels) tab[base + slot * cpt + k] = (int)idx;
            }
        }
    }
    if (tab.size() > S->btiles_cap) {
        HIP_TRY(hipStreamSynchronize(S->ctx->stream));
        hipFree(S->d_btiles); S->d_btiles = nullptr; S->btiles_cap = 0;
        if (int rc = dzalloc(S->ctx, tab.size(), &S->d_btiles)) return rc;
        S->btiles_cap = tab.size();
    }
    HIP_TRY(hipMemcpyAsync(S->d_btiles, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, S->ctx->stream));
    HIP_TRY(hipStreamSynchronize(S->ctx->stream));
    S->bt_n = (uint32_t)n; S->bt_wgs = wgs; S->bt_nw = nw; S->bt_tpw = tpw;
    return 0;
}
extern "C" int msdr_fir_q15_process(msdr_fir_q15 *S, const q15_t *d_src, q15_t *d_dst, uint32_t blockSize)
{
    if (!S || !S->d_qm_tab) return fir_process<FirQ15>(S, d_src, d_dst, blockSize);
    if (int rc = bind(S->ctx)) return rc;
    if (blockSize == 0) return 0;
    if (!d_src || !d_dst) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    if ((const void *)d_src == (const void *)d_dst)
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "FIR process is not in-place (the reference uses separate buffers, Minimal-SDR.ino:574-578)");
    ChainParams q;
    memset(&q, 0, sizeof q);
    q.x = d_src; q.out = d_dst; q.hist_in = S->d_hist[S->cur]; q.n = (long long)blockSize; q.channels = (int)S->channels;
    q.hist_len = (int)S->hist_len; q.ntaps_pad = (int)S->ntaps_pad; q.mixer = MSDR_MIXER_FS4; q.phase0 = 0;
    q.mf_tab = S->d_qm_tab; q.mf_stride = S->qm_stride; q.mf_halo = S->qm_halo; q.mf_bsteps = S->qm_bsteps; q.mf_units = S->d_qm_order;
    // the reference's cadence -- arm_fir_fast_q15(&FIR_I, I_buffer, I_FIR_out, AUDIO_BLOCK_SAMPLES), Minimal-SDR.ino:574-575 --: channel-batched
    // tiles, ONE kernel that also writes the next history (msdr_chain_q15mb.hiph, flavour 3); MSDR_NO_BLOCK=1 keeps the long-call kernel
    if (mb_n_ok((long long)blockSize) && (int)S->hist_len == S->qm_halo && !getenv("MSDR_NO_BLOCK") &&
        ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) & 15) == 0 &&
        (unsigned long long)S->channels * std::max<unsigned long long>(S->hist_len, blockSize) * 2ull < (1ull << 32)) {
        const int brc = fir_q15_block_tiles(S, (int)blockSize);
        if (brc > 0) return brc;
        if (brc == 0) {
            q.hist_out = S->d_hist[S->cur ^ 1];
            q.mf_units = S->d_btiles; q.mf_nw = (int)S->bt_nw; q.nseg = (int)S->bt_tpw;
            const size_t lds = qb_lds_bytes(S->qm_halo, (int)blockSize, S->qm_bsteps, (int)S->bt_nw, (int)S->bt_tpw);
            { KernelTimer kt(S->ctx); (void)launch_chain_q15mb(S->ctx->stream, 3, false, S->bt_wgs, S->bt_nw * 64, lds, q); }
            if (int rc = launch_check("chain_q15mb_kernel<3>")) return rc;
            S->cur ^= 1;
            return 0;
        }
    }
    const long long qtiles = ((long long)blockSize + kQmTile - 1) / kQmTile;
    long long qseg = (8192 + S->channels - 1) / S->channels;                       // two rounds of 16 waves per CU, >= two tiles per segment
    qseg = std::max<long long>(1, std::min<long long>(qseg, std::max<long long>(1, qtiles / 2)));
    const long long qseg_len = ((qtiles + qseg - 1) / qseg) * kQmTile;
    qseg = ((long long)blockSize + qseg_len - 1) / qseg_len;
    int nw = 8;
    while (nw > 1 && (long long)S->channels * qseg < 256LL * nw) nw >>= 1;
    while (nw > 1 && qm_lds_bytes(S->qm_halo, S->qm_bsteps, nw) > 80 * 1024) nw >>= 1;
    q.mf_nw = nw; q.nseg = (int)qseg; q.seg_len = qseg_len; q.fold_period = 0; q.fold_rot = (int)S->channels; q.mf_waves = 0;
    const unsigned grid = (unsigned)(((long long)S->channels * qseg + nw - 1) / nw);
    { KernelTimer kt(S->ctx); (void)launch_chain_q15mf(S->ctx->stream, 3, false, grid, (unsigned)nw * 64, qm_lds_bytes(S->qm_halo, S->qm_bsteps, nw), q); }
    if (int rc = launch_check("chain_q15mf_kernel<3>")) return rc;
    hipLaunchKernelGGL((history_kernel<int16_t>), dim3(grid_1d((long long)S->channels * S->hist_len)), dim3(256), 0, S->ctx->stream,
                       d_src, (const int16_t *)S->d_hist[S->cur], S->d_hist[S->cur ^ 1], (long long)blockSize, (int)S->hist_len, (int)S->channels);
    if (int rc = launch_check("history_kernel")) return rc;
    S->cur ^= 1;
    return 0;
}
extern "C" int msdr_fir_q15_reset(msdr_fir_q15 *S) { return fir_reset(S); }
extern "C" int msdr_fir_q15_destroy(msdr_fir_q15 *S)
{
    if (S) { hipFree(S->d_qm_tab); hipFree(S->d_qm_order); hipFree(S->d_btiles); }
    return fir_destroy(S);
}
// New coefficients under a running filter (the reference: the array behind S->pCoeffs rewritten in place, UI.cpp:337-345 +
// Minimal-SDR.ino:221-223; arm_fir_init_q15.c:100-109 stored the pointer only).  Every device table is rebuilt the way create builds
// it -- by creating a second instance -- and the running instance's state (its history pair) moves over; the old tables go once the
// work queued on them has finished.
template <typename Inst, typename Create>
static int fir_swap_coeffs(Inst *S, Create &&create)
{
    Inst *n = nullptr;
    if (int rc = create(&n)) return rc;
    if (n->hist_len != S->hist_len || n->channels != S->channels) { (void)fir_destroy(n); return fail(MSDR_STATUS_SIZE_MISMATCH, "internal: geometry changed"); }
    std::swap(n->d_hist[0], S->d_hist[0]); std::swap(n->d_hist[1], S->d_hist[1]); n->cur = S->cur;
    std::swap(*S, *n);
    return 0;                                   // the caller destroys n (the old tables + the fresh, unused history)
}
extern "C" int msdr_fir_q15_set_coeffs(msdr_fir_q15 *S, const q15_t *pCoeffs)
{
    if (!S || !pCoeffs) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    msdr_fir_q15 *old = nullptr;
    auto create = [&](msdr_fir_q15 **n) { const int rc = msdr_fir_q15_create(S->ctx, (uint16_t)S->ntaps, pCoeffs, S->channels, n); old = *n; return rc; };
    if (int rc = fir_swap_coeffs(S, create)) { if (old) msdr_fir_q15_destroy(old); return rc; }
    return msdr_fir_q15_destroy(old);
}

extern "C" int msdr_fir_f32_destroy(msdr_fir_f32 *S);
static int fir_f32_upload_header(msdr_fir_f32 *S)
{
    F32MfHeader h;
    h.nsteps = S->fm_bsteps; h.ex = S->fm_ex; h.use_fixed = S->input_range > 0.0f ? 1 : 0; h.fixed_k = 0;
    if (h.use_fixed) {
        int k = 0;
        (void)std::frexp((double)S->input_range, &k);                 // input_range <= 2^k
        if (std::ldexp(1.0, k - 1) == (double)S->input_range) k--;    // an exact power of two
        h.fixed_k = std::max(-100, std::min(100, 15 - k));
    }
    HIP_TRY(hipMemcpyAsync(S->d_fm_tab, &h, sizeof h, hipMemcpyHostToDevice, S->ctx->stream));
    if (S->d_tr_tab) {
        F32TrHeader t;
        t.ns = S->tr_ns; t.ex = S->fm_ex; t.fixed_k = h.fixed_k; t.use_fixed = h.use_fixed; t.skip1 = S->tr_skip1;
        HIP_TRY(hipMemcpyAsync(S->d_tr_tab, &t, sizeof t, hipMemcpyHostToDevice, S->ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(S->ctx->stream));
    return 0;
}
extern "C" int msdr_fir_f32_create(msdr_ctx *ctx, uint16_t numTaps, const float32_t *pCoeffs, uint32_t channels, msdr_fir_f32 **out)
{
    if (out) *out = nullptr;
    if (int rc = fir_create<msdr_fir_f32, float, float>(ctx, numTaps, pCoeffs, channels, out)) return rc;
    msdr_fir_f32 *S = *out;
    const int N = (int)numTaps, H = fm_halo(N), ns = (H + 32) / 16;
    double maxabs = 0.0;
    bool finite = true;
    for (int k = 0; k < N; k++) { maxabs = std::max(maxabs, std::fabs((double)pCoeffs[k])); finite = finite && std::isfinite(pCoeffs[k]); }
    if (N >= 16 && H <= 512 && finite && maxabs > 0.0 && fm_lds_bytes(H, ns, 1) <= 160 * 1024) {
        int ex = 0;
        (void)std::frexp(maxabs, &ex);
        ex = 14 - ex;                                             // maxabs 2^ex in [2^13, 2^14)
        std::vector<char> blob(kFmHdrBytes + (size_t)ns * 2048, 0);
        _Float16 *fr = reinterpret_cast<_Float16 *>(blob.data() + kFmHdrBytes);
        for (int st = 0; st < ns; st++)
            for (int l = 0; l < 64; l++)
                for (int jj = 0; jj < 8; jj++) {
                    // B[i][b] = tap at delay H + b - i (arm_fir keeps its coefficients time-reversed: delay d is pCoeffs[N - 1 - d])
                    const int i = 16 * st + 8 * (l >> 5) + jj, b = l & 31, d = H + b - i;
                    const double val = (d >= 0 && d < N) ? std::ldexp((double)pCoeffs[N - 1 - d], ex) : 0.0;
                    const _Float16 vh = (_Float16)val;
                    fr[(size_t)st * 1024 + l * 8 + jj] = vh;
                    fr[(size_t)st * 1024 + 512 + l * 8 + jj] = (_Float16)(val - (double)vh);
                }
        int rc = upload(ctx, blob, &S->d_fm_tab);
        S->fm_halo_ = H; S->fm_bsteps = ns; S->fm_ex = ex;
        const int trs = tr_steps(N);
        if (!rc && trs <= kTrMaxSteps) {
            // A fragments of v_mfma_f32_16x16x32_f16 (lane l: row a = l & 15, k = 8 (l >> 4) + j): T_F(s)[a][k'] = the tap at delay
            // H + 16 F + a - k', k' = 32 s + 8 (l >> 4) + j; family F serves the sub-tiles a0 = 16 F and 32 + 16 F (one step later)
            std::vector<char> tb(tr_table_bytes(trs), 0);
            _Float16 *tf = reinterpret_cast<_Float16 *>(tb.data() + kTrHdrBytes);
            for (int F = 0; F < 2; F++)
                for (int st = 0; st < trs; st++)
                    for (int l = 0; l < 64; l++)
                        for (int jj = 0; jj < 8; jj++) {
                            const int kk = 32 * st + 8 * (l >> 4) + jj, d = H + 16 * F + (l & 15) - kk;
                            const double val = (d >= 0 && d < N) ? std::ldexp((double)pCoeffs[N - 1 - d], ex) : 0.0;
                            const _Float16 vh = (_Float16)val;
                            const size_t o = ((size_t)(F * trs + st) * 2) * 512 + l * 8 + jj;
                            tf[o] = vh;
                            tf[o + 512] = (_Float16)(val - (double)vh);
                        }
            rc = upload(ctx, tb, &S->d_tr_tab);
            S->tr_ns = trs; S->tr_skip1 = (N <= H - 15) ? 1 : 0;
            if (!rc) rc = dzalloc(ctx, (size_t)kTqMaxFronts * kTqCtrStride + 16, &S->d_tq_ctr);      // the fronts' counters | the "waves out" word
        }
        if (!rc) rc = fir_f32_upload_header(S);
        if (rc) { msdr_fir_f32_destroy(S); *out = nullptr; return rc; }
    }
    return 0;
}
extern "C" int msdr_fir_f32_set_input_range(msdr_fir_f32 *S, float max_abs)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(S->ctx)) return rc;
    if (!(max_abs >= 0.0f) || !std::isfinite(max_abs)) return fail(MSDR_STATUS_ARGUMENT_ERROR, "input range must be finite and >= 0 (0 = automatic)");
    S->input_range = max_abs;
    return S->d_fm_tab ? fir_f32_upload_header(S) : 0;
}
extern "C" int msdr_fir_f32_process(msdr_fir_f32 *S, const float32_t *d_src, float32_t *d_dst, uint32_t blockSize)
{
    if (!S || !S->d_fm_tab) return fir_process<FirF32>(S, d_src, d_dst, blockSize);
    if (int rc = bind(S->ctx)) return rc;
    if (blockSize == 0) return 0;
    if (!d_src || !d_dst) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    if ((const void *)d_src == (const void *)d_dst)
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "FIR process is not in-place (the reference uses separate buffers, Minimal-SDR.ino:574-578)");
    if (S->d_tr_tab && S->d_tq_ctr && (((long long)blockSize + kTrTile - 1) / kTrTile) * (long long)S->channels < 0x7FFFFFFFll) {
        // taps in registers, tiles dealt from a queue in address order (msdr_fir_f32tq.hiph): a persistent grid of two workgroups per CU
        TqParams q;
        q.x = d_src; q.y = d_dst; q.hist = (const float *)S->d_hist[S->cur]; q.tab = (const char *)S->d_tr_tab;
        q.n = (long long)blockSize; q.channels = (int)S->channels; q.hist_len = (int)S->hist_len;
        q.tpr = (unsigned)(((long long)blockSize + kTrTile - 1) / kTrTile);
        q.tpr_shift = -1;
        for (int sh = 0; sh < 31; sh++) if (q.tpr == (1u << sh)) q.tpr_shift = sh;
        q.total = q.tpr * (unsigned)S->channels;
        const unsigned want = (unsigned)std::min<long long>(2LL * S->ctx->num_cus, ((long long)q.total + 3) / 4);
        q.fronts = (unsigned)std::max(1, std::min<int>(S->tq_fronts, (int)want));
        q.per_front = (q.total + q.fronts - 1) / q.fronts;
        q.ctr = S->d_tq_ctr; q.ctr_next = nullptr; q.done = S->d_tq_ctr + (size_t)kTqMaxFronts * kTqCtrStride;      // (the kernel leaves both as it found them: zero)
        q.all_aligned = ((blockSize & 3u) == 0 && (reinterpret_cast<uintptr_t>(d_src) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_dst) & 15) == 0) ? 1 : 0;
        const unsigned grid = std::max(want, q.fronts);
        // runs of consecutive tiles per draw (the halo of the second .. last comes from registers): as long as every wave still gets >= 64 runs
        // (the queue's tail is one run per wave at most)
        q.run_shift = 0;
        { const int cap = S->tq_run_shift_cap;
          while (q.run_shift < cap && (q.tpr >> (q.run_shift + 1)) >= 1 && (S->tq_run_forced || (long long)q.total >= (long long)grid * 4 * 64 * (2LL << q.run_shift))) q.run_shift++; }
        const size_t lds = 4 * tr_wave_bytes(S->tr_ns);
        { KernelTimer kt(S->ctx);
          if (launch_fir_f32tq(S->ctx->stream, S->tr_ns, S->tr_skip1, grid, lds, q) == hipErrorInvalidValue) return fail(MSDR_STATUS_ARGUMENT_ERROR, "fir_f32tq: step count not built"); }
        if (int rc = launch_check("fir_f32tq_kernel")) return rc;
        S->last_launch = 1;
        hipLaunchKernelGGL((history_kernel<float>), dim3(grid_1d((long long)S->channels * S->hist_len)), dim3(256), 0, S->ctx->stream,
                           d_src, (const float *)S->d_hist[S->cur], S->d_hist[S->cur ^ 1], (long long)blockSize, (int)S->hist_len, (int)S->channels);
        if (int rc = launch_check("history_kernel")) return rc;
        S->cur ^= 1;
        return 0;
    }
    const int H = S->fm_halo_, ns = S->fm_bsteps;
    const long long tiles = ((long long)blockSize + kFmTile - 1) / kFmTile;
    long long nseg = (8192 + S->channels - 1) / S->channels;                       // two rounds of 16 waves per CU, >= two tiles per segment
    nseg = std::max<long long>(1, std::min<long long>(nseg, std::max<long long>(1, tiles / 2)));
    const long long seg_len = ((tiles + nseg - 1) / nseg) * kFmTile;
    nseg = ((long long)blockSize + seg_len - 1) / seg_len;
    int nw = 16;
    while (nw > 1 && ((long long)S->channels * nseg < 256LL * nw || fm_lds_bytes(H, ns, nw) > 160 * 1024)) nw >>= 1;
    const unsigned grid = (unsigned)(((long long)S->channels * nseg + nw - 1) / nw);
    { KernelTimer kt(S->ctx);
      (void)launch_fir_f32mf(S->ctx->stream, grid, (unsigned)nw * 64, fm_lds_bytes(H, ns, nw), d_src, d_dst,
                             (const float *)S->d_hist[S->cur], (const char *)S->d_fm_tab, (long long)blockSize, (int)S->channels, (int)nseg, seg_len,
                             (int)S->hist_len, H, ns, nw); }
    if (int rc = launch_check("fir_f32mf_kernel")) return rc;
    S->last_launch = 2;
    hipLaunchKernelGGL((history_kernel<float>), dim3(grid_1d((long long)S->channels * S->hist_len)), dim3(256), 0, S->ctx->stream,
                       d_src, (const float *)S->d_hist[S->cur], S->d_hist[S->cur ^ 1], (long long)blockSize, (int)S->hist_len, (int)S->channels);
    if (int rc = launch_check("history_kernel")) return rc;
    S->cur ^= 1;
    return 0;
}
extern "C" int msdr_fir_f32_reset(msdr_fir_f32 *S) { return fir_reset(S); }
extern "C" int msdr_fir_f32_set_coeffs(msdr_fir_f32 *S, const float32_t *pCoeffs)
{
    if (!S || !pCoeffs) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    msdr_fir_f32 *old = nullptr;
    const float range = S->input_range;
    auto create = [&](msdr_fir_f32 **n) {
        int rc = msdr_fir_f32_create(S->ctx, (uint16_t)S->ntaps, pCoeffs, S->channels, n);
        old = *n;
        if (!rc && range > 0.0f) rc = msdr_fir_f32_set_input_range(*n, range);
        return rc;
    };
    if (int rc = fir_swap_coeffs(S, create)) { if (old) msdr_fir_f32_destroy(old); return rc; }
    return msdr_fir_f32_destroy(old);
}
// which kernel msdr_fir_f32_process launches for this instance (benchmarks / tests)
extern "C" const char *msdr_fir_f32_kernel_name(msdr_fir_f32 *S)
{
    static thread_local char name[64];
    if (!S) return "";
    // (a call whose tile count does not fit the queue's 32-bit index runs the stream kernel: after a call the name is the launched one)
    if (S->d_tr_tab && S->d_tq_ctr && S->last_launch != 2) snprintf(name, sizeof name, "fir_f32tq_kernel<%d, %s>", S->tr_ns, S->tr_skip1 ? "true" : "false");
    else snprintf(name, sizeof name, "%s", S->d_fm_tab ? "fir_f32mf_kernel" : "fir_kernel<FirF32>");
    return name;
}
extern "C" int msdr_fir_f32_destroy(msdr_fir_f32 *S)
{
    if (S) { hipFree(S->d_fm_tab); hipFree(S->d_tr_tab); hipFree(S->d_tq_ctr); }
    return fir_destroy(S);
}

// ------------------------------------------------------------------------------------------------
// arm_biquad_cascade_df1_f32 mirror
// ------------------------------------------------------------------------------------------------
struct msdr_biquad_df1_f32 {
    msdr_ctx *ctx;
    uint32_t channels, stages;
    BiquadCascadeTables<kBqR> *d_tabs;
    float *d_state;   // [channels][kBqStateFloats]
    float *d_state_alt;   // ping-pong partner: a segmented launch reads one and writes the other
    double pole_radius;
    bool sequential;      // ill-conditioned for the parallel evaluation (cascade_condition): biquad_df1_seq_kernel, CMSIS order
    bool seq_segments;    // sequential kernel: split long blocks of few channels into time segments (MSDR_BIQUAD_SEQ_NO_SEGMENTS, read at init)
    float *d_coeffs;      // sequential: the 5 x stages coefficients
    float *d_seq_scratch; // sequential, few channels x long block: the segments' warm-up samples (biquad_seqseg_gather_kernel)
    size_t seq_scratch_floats;
    std::vector<float> h_coeffs;      // 5 x stages, as given (msdr_biquad_df1_f32_set_coeffs converts the state between the two cascades)
};

extern "C" int msdr_biquad_df1_f32_cascade_info(uint8_t numStages, const float32_t *pCoeffs, double *kappa, double *fp32_noise, int *cmsis_order)
{
    if (numStages > kMaxStages) return fail(MSDR_STATUS_ARGUMENT_ERROR, "numStages %u > %d", (unsigned)numStages, kMaxStages);
    if (numStages && !pCoeffs) return fail(MSDR_STATUS_ARGUMENT_ERROR, "pCoeffs is null");
    if (kappa) *kappa = cascade_condition(pCoeffs, (int)numStages);
    if (fp32_noise) *fp32_noise = numStages ? cascade_fp32_noise(pCoeffs, (int)numStages) : 0.0;
    if (cmsis_order) *cmsis_order = (numStages && cascade_needs_cmsis_order(pCoeffs, (int)numStages)) ? 1 : 0;
    return 0;
}
// set around msdr_biquad_df1_f32_create by a chain that must have its cascade in CMSIS order whatever the conditioning figures say
// (three or more sections behind chain_kernel<ArithF32>: chain_create_impl)
static thread_local bool g_biquad_force_sequential = false;
extern "C" int msdr_biquad_df1_f32_create(msdr_ctx *ctx, uint8_t numStages, const float32_t *pCoeffs, uint32_t channels,
                                          msdr_biquad_df1_f32 **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (int rc = bind(ctx)) return rc;
    if (numStages > kMaxStages) return fail(MSDR_STATUS_ARGUMENT_ERROR, "numStages %u > %d", (unsigned)numStages, kMaxStages);
    if ((numStages && !pCoeffs) || channels == 0) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad biquad arguments");
    msdr_biquad_df1_f32 *S = new (std::nothrow) msdr_biquad_df1_f32();
    if (!S) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    S->ctx = ctx; S->channels = channels; S->stages = numStages; S->d_tabs = nullptr; S->d_state = nullptr; S->d_state_alt = nullptr;
    S->pole_radius = numStages ? max_pole_radius(pCoeffs, (int)numStages) : 0.0;
    S->d_coeffs = nullptr; S->d_seq_scratch = nullptr; S->seq_scratch_floats = 0;
    S->sequential = numStages > 0 && (g_biquad_force_sequential || cascade_needs_cmsis_order(pCoeffs, (int)numStages));
    S->seq_segments = true;
    if (numStages) S->h_coeffs.assign(pCoeffs, pCoeffs + 5 * numStages);
    if (S->sequential) {
        std::vector<float> cf(pCoeffs, pCoeffs + 5 * numStages);
        if (int rc = upload(ctx, cf, &S->d_coeffs)) { delete S; return rc; }
    }
    std::vector<BiquadCascadeTables<kBqR>> tabs(1);
    make_cascade_tables<kBqR>(pCoeffs, numStages, &tabs[0]);
    int rc = upload(ctx, tabs, &S->d_tabs);
    if (!rc) rc = dzalloc(ctx, (size_t)channels * kBqStateFloats, &S->d_state);
    if (!rc) rc = dzalloc(ctx, (size_t)channels * kBqStateFloats, &S->d_state_alt);
    if (rc) { hipFree(S->d_tabs); hipFree(S->d_state); hipFree(S->d_state_alt); hipFree(S->d_coeffs); delete S; return rc; }
    *out = S;
    return 0;
}
extern "C" int msdr_biquad_df1_f32_process(msdr_biquad_df1_f32 *S, const float32_t *d_src, float32_t *d_dst, uint32_t blockSize)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(S->ctx)) return rc;
    if (blockSize == 0) return 0;
    if (!d_src || !d_dst) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    if (S->stages == 0) {      // empty cascade: pass-through
        if (d_src != d_dst) HIP_TRY(hipMemcpyAsync(d_dst, d_src, (size_t)S->channels * blockSize * sizeof(float), hipMemcpyDeviceToDevice, S->ctx->stream));
        return 0;
    }
    if (S->sequential) {
        // few channels and a long block: one lane per (channel, time segment), each segment warmed up over the samples in front of
        // it (copied aside first, so in place stays allowed); the warm-up length follows from the slowest pole (1e-10 of the state)
        long long nseg = 1, seg_len = blockSize, warm = 0;
        if (S->channels < 8192 && S->pole_radius > 0.0 && S->pole_radius < 0.99999 && S->seq_segments) {
            warm = ((long long)std::ceil(std::log(1e-10) / std::log(S->pole_radius)) + 64 * S->stages + 3) & ~3LL;
            const long long min_len = std::max<long long>(8 * warm, 1024);
            const long long want = (65536 + S->channels - 1) / S->channels;
            nseg = std::max<long long>(1, std::min(want, (long long)blockSize / min_len));
            seg_len = (((long long)blockSize + nseg - 1) / nseg + 3) & ~3LL;
            nseg = ((long long)blockSize + seg_len - 1) / seg_len;
        }
        if (nseg > 1) {
            const size_t need = (size_t)S->channels * (size_t)nseg * (size_t)warm;
            if (need > S->seq_scratch_floats) {
                HIP_TRY(hipStreamSynchronize(S->ctx->stream));
                hipFree(S->d_seq_scratch); S->d_seq_scratch = nullptr; S->seq_scratch_floats = 0;
                HIP_TRY(hipMalloc((void **)&S->d_seq_scratch, need * sizeof(float)));
                S->seq_scratch_floats = need;
            }
            hipLaunchKernelGGL(biquad_seqseg_gather_kernel, dim3((unsigned)std::min<size_t>((need + 255) / 256, 65536)), dim3(256), 0, S->ctx->stream,
                               d_src, S->d_seq_scratch, (long long)blockSize, (int)S->channels, (int)nseg, seg_len, (int)warm);
            if (int rc = launch_check("biquad_seqseg_gather_kernel")) return rc;
            hipLaunchKernelGGL(biquad_df1_seqseg_kernel, dim3((unsigned)(((long long)S->channels * nseg + 63) / 64)), dim3(64), 0, S->ctx->stream,
                               d_src, d_dst, (long long)blockSize, (int)S->channels, (int)S->stages, (const float *)S->d_coeffs,
                               (const float *)S->d_state, S->d_state_alt, (int)nseg, seg_len, (int)warm, (const float *)S->d_seq_scratch);
            std::swap(S->d_state, S->d_state_alt);
            return launch_check("biquad_df1_seqseg_kernel");
        }
        hipLaunchKernelGGL(biquad_df1_seq_kernel, dim3((S->channels + 63) / 64), dim3(64), 0, S->ctx->stream, d_src, d_dst, (long long)blockSize,
                           (int)S->channels, (int)S->stages, (const float *)S->d_coeffs, S->d_state);
        return launch_check("biquad_df1_seq_kernel");
    }
    // time segments for long blocks of few channels (never in place: a segment's warm-up reads its predecessor's input)
    const long long tiles = ((long long)blockSize + kBqTile - 1) / kBqTile;
    long long nseg = 1, warm_tiles = 0;
    if ((const void *)d_src != (const void *)d_dst && S->pole_radius > 0.0 && S->pole_radius < 0.99999) {
        const long long w = (long long)std::ceil(std::log(1e-10) / std::log(S->pole_radius)) + 64 * S->stages;
        warm_tiles = (w + kBqTile - 1) / kBqTile;
        if (warm_tiles <= 64) {
            const long long want = (2048 + S->channels - 1) / S->channels;
            nseg = std::max<long long>(1, std::min(want, tiles / std::max<long long>(4, 8 * warm_tiles)));
        }
    }
    const long long seg_tiles = (tiles + nseg - 1) / nseg;
    nseg = (tiles + seg_tiles - 1) / seg_tiles;
    hipLaunchKernelGGL(biquad_df1_kernel, dim3((unsigned)(S->channels * nseg)), dim3(kThreads), 0, S->ctx->stream, d_src, d_dst,
                       (long long)blockSize, (const BiquadCascadeTables<kBqR> *)S->d_tabs, (const float *)S->d_state, S->d_state_alt,
                       (int)nseg, seg_tiles * kBqTile, (int)(nseg > 1 ? warm_tiles * kBqTile : 0));
    std::swap(S->d_state, S->d_state_alt);
    return launch_check("biquad_df1_kernel");
}
extern "C" int msdr_biquad_df1_f32_reset(msdr_biquad_df1_f32 *S)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(S->ctx)) return rc;
    HIP_TRY(hipMemsetAsync(S->d_state, 0, (size_t)S->channels * kBqStateFloats * sizeof(float), S->ctx->stream));
    HIP_TRY(hipMemsetAsync(S->d_state_alt, 0, (size_t)S->channels * kBqStateFloats * sizeof(float), S->ctx->stream));
    return 0;
}
extern "C" int msdr_biquad_df1_f32_destroy(msdr_biquad_df1_f32 *S)
{
    if (!S) return 0;
    if (int rc = bind(S->ctx)) return rc;
    (void)hipStreamSynchronize(S->ctx->stream);
    hipFree(S->d_tabs); hipFree(S->d_state); hipFree(S->d_state_alt); hipFree(S->d_coeffs); hipFree(S->d_seq_scratch);
    delete S;
    return 0;
}

// ---- the cascade's state in CMSIS terms (msdr_cascade_state.h) ----
extern "C" int msdr_biquad_df1_f32_state_to_cmsis(uint8_t numStages, const float32_t *pCoeffs, const float32_t lib_state[16], float32_t *pState)
{
    if (numStages > kMaxStages || (numStages && !pCoeffs) || !lib_state || !pState) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad argument");
    cstate::Bridge br;
    br.build(pCoeffs, pCoeffs, (int)numStages);
    if (!br.ok_to) return fail(MSDR_STATUS_ARGUMENT_ERROR, "this cascade's block-parallel state has no unique CMSIS state (a numerator shares a root with an earlier denominator)");
    double D[8], Y[2 * kMaxStages + 2];
    for (int k = 0; k < 8; k++) D[k] = lib_state[k];
    br.lib_to_cmsis(D, lib_state, Y);
    cstate::Bridge::y_to_pstate(Y, (int)numStages, pState);
    return 0;
}
extern "C" int msdr_biquad_df1_f32_state_from_cmsis(uint8_t numStages, const float32_t *pCoeffs, const float32_t *pState, float32_t lib_state[16])
{
    if (numStages > kMaxStages || (numStages && !pCoeffs) || !lib_state || !pState) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad argument");
    cstate::Bridge br;
    br.build(pCoeffs, pCoeffs, (int)numStages);
    if (!br.ok_from) return fail(MSDR_STATUS_ARGUMENT_ERROR, "no block-parallel state reproduces this CMSIS state");
    double D[8], Y[2 * kMaxStages + 2];
    cstate::Bridge::pstate_to_y(pState, (int)numStages, Y);
    for (int k = 0; k < 8; k++) D[k] = (k < 2 * numStages) ? lib_state[k] : 0.0;
    if (numStages) { D[0] = Y[0]; D[1] = Y[1]; }
    for (int k = 0; k < 8; k++) lib_state[k] = (float)D[k];
    for (int k = 8; k < 16; k++) lib_state[k] = 0.0f;
    br.cmsis_to_lib(Y, D, lib_state + 8);
    return 0;
}
// every channel's state of a stage instance as CMSIS values Y (2 S + 2 per channel) and input history D (8 per channel)
static int biquad_df1_read_cmsis(msdr_biquad_df1_f32 *S, const cstate::Bridge &br, std::vector<double> &Y, std::vector<double> &D)
{
    const int St = (int)S->stages, ny = 2 * St + 2;
    std::vector<float> st((size_t)S->channels * kBqStateFloats);
    HIP_TRY(hipStreamSynchronize(S->ctx->stream));
    HIP_TRY(hipMemcpy(st.data(), S->d_state, st.size() * sizeof(float), hipMemcpyDeviceToHost));
    Y.assign((size_t)S->channels * ny, 0.0); D.assign((size_t)S->channels * 8, 0.0);
    for (uint32_t ch = 0; ch < S->channels; ch++) {
        const float *r = st.data() + (size_t)ch * kBqStateFloats;
        double *y = Y.data() + (size_t)ch * ny, *d = D.data() + (size_t)ch * 8;
        if (S->sequential) {
            cstate::Bridge::pstate_to_y(r, St, y);
            d[0] = y[0]; d[1] = y[1];                              // (older inputs are not part of a CMSIS state: zeros, a valid choice)
        } else {
            for (int k = 0; k < 8; k++) d[k] = r[k];
            br.lib_to_cmsis(d, r, y);
        }
    }
    return 0;
}
static int biquad_df1_write_cmsis(msdr_biquad_df1_f32 *S, const cstate::Bridge &br, const std::vector<double> &Y, const std::vector<double> &D)
{
    const int St = (int)S->stages, ny = 2 * St + 2;
    std::vector<float> st((size_t)S->channels * kBqStateFloats, 0.0f);
    for (uint32_t ch = 0; ch < S->channels; ch++) {
        float *r = st.data() + (size_t)ch * kBqStateFloats;
        const double *y = Y.data() + (size_t)ch * ny, *d = D.data() + (size_t)ch * 8;
        if (S->sequential) cstate::Bridge::y_to_pstate(y, St, r);
        else {
            for (int k = 0; k < 2 * St; k++) r[k] = (float)d[k];
            br.cmsis_to_lib(y, d, r + 8);
        }
    }
    HIP_TRY(hipMemcpy(S->d_state, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
extern "C" int msdr_biquad_df1_f32_get_cmsis_state(msdr_biquad_df1_f32 *S, uint32_t channel, float32_t *pState)
{
    if (!S || !pState || channel >= S->channels) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad argument");
    if (int rc = bind(S->ctx)) return rc;
    if (S->stages == 0) return 0;
    float r[kBqStateFloats];
    HIP_TRY(hipStreamSynchronize(S->ctx->stream));
    HIP_TRY(hipMemcpy(r, S->d_state + (size_t)channel * kBqStateFloats, sizeof r, hipMemcpyDeviceToHost));
    if (S->sequential) { memcpy(pState, r, (size_t)4 * S->stages * sizeof(float)); return 0; }
    return msdr_biquad_df1_f32_state_to_cmsis((uint8_t)S->stages, S->h_coeffs.data(), r, pState);
}
extern "C" int msdr_biquad_df1_f32_set_coeffs(msdr_biquad_df1_f32 *S, const float32_t *pCoeffs)
{
    if (!S || (S->stages && !pCoeffs)) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(S->ctx)) return rc;
    if (S->stages == 0) return 0;
    cstate::Bridge br;
    br.build(S->h_coeffs.data(), pCoeffs, (int)S->stages);
    if (!S->sequential && !br.ok_to)
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "the running cascade's block-parallel state has no unique CMSIS state (a numerator shares a root with an earlier denominator)");
    std::vector<double> Y, D;
    if (int rc = biquad_df1_read_cmsis(S, br, Y, D)) return rc;
    msdr_biquad_df1_f32 *n = nullptr;
    if (int rc = msdr_biquad_df1_f32_create(S->ctx, (uint8_t)S->stages, pCoeffs, S->channels, &n)) return rc;
    if (!n->sequential && !br.ok_from) { msdr_biquad_df1_f32_destroy(n); return fail(MSDR_STATUS_ARGUMENT_ERROR, "no block-parallel state reproduces the CMSIS state under the new coefficients"); }
    n->seq_segments = S->seq_segments;
    if (int rc = biquad_df1_write_cmsis(n, br, Y, D)) { msdr_biquad_df1_f32_destroy(n); return rc; }
    std::swap(*S, *n);
    return msdr_biquad_df1_f32_destroy(n);
}

// ------------------------------------------------------------------------------------------------
// Teensy AudioFilterBiquad mirror
// ------------------------------------------------------------------------------------------------
struct msdr_biquad_q15 {
    msdr_ctx *ctx;
    uint32_t channels;
    int *d_defs;      // [channels][32]
    int max_stage;    // highest stage index given to setCoefficients so far (-1: none): stages above 0 chain through the records' flag bits
    int pipe_ch;      // channels per workgroup of biquad_teensy_pipe4_kernel (64 / 32 / 16), 0: never that kernel (read at create time)
};
static int tq4_pipe_ch_at_create(uint32_t channels)
{
    if (const char *e = getenv("MSDR_BIQUAD_PIPE_CH")) { const int v = atoi(e); if (v == 64 || v == 32 || v == 16) return v; }
    return tq4_channels_per_group(channels);
}

// filter_biquad.cpp:84-100 applied to every channel's record
__global__ void tbq_set_coef_kernel(int *defs, int channels, int stage, int c0, int c1, int c2, int c3, int c4)
{
    for (int ch = blockIdx.x * blockDim.x + threadIdx.x; ch < channels; ch += gridDim.x * blockDim.x) {
        int *dest = defs + (long long)ch * 32 + (stage << 3);
        if (stage > 0) dest[-1] |= 0x80000000;
        dest[0] = c0; dest[1] = c1; dest[2] = c2;
        dest[3] = (int)(0u - (unsigned)c3);
        dest[4] = (int)(0u - (unsigned)c4);
        dest[7] &= 0x80000000;
    }
}

extern "C" int msdr_biquad_q15_create(msdr_ctx *ctx, uint32_t channels, msdr_biquad_q15 **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (int rc = bind(ctx)) return rc;
    if (channels == 0) return fail(MSDR_STATUS_ARGUMENT_ERROR, "channels == 0");
    msdr_biquad_q15 *S = new (std::nothrow) msdr_biquad_q15();
    if (!S) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    S->ctx = ctx; S->channels = channels; S->d_defs = nullptr; S->max_stage = -1; S->pipe_ch = tq4_pipe_ch_at_create(channels);
    if (int rc = dzalloc(ctx, (size_t)channels * 32, &S->d_defs)) { delete S; return rc; }   // h:36-39: passes nothing
    *out = S;
    return 0;
}
extern "C" int msdr_biquad_q15_set_coefficients(msdr_biquad_q15 *S, uint32_t stage, const int32_t coef[5])
{
    if (!S || !coef) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (stage >= 4) return 0;                                   // filter_biquad.cpp:86: silently ignored
    if (int rc = bind(S->ctx)) return rc;
    S->max_stage = std::max(S->max_stage, (int)stage);
    hipLaunchKernelGGL(tbq_set_coef_kernel, dim3(grid_1d(S->channels)), dim3(256), 0, S->ctx->stream, S->d_defs, (int)S->channels,
                       (int)stage, coef[0], coef[1], coef[2], coef[3], coef[4]);
    return launch_check("tbq_set_coef_kernel");
}
extern "C" int msdr_biquad_q15_update(msdr_biquad_q15 *S, q15_t *d_data, uint32_t blockSize)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(S->ctx)) return rc;
    if (blockSize == 0) return 0;
    if (!d_data) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    if (blockSize & 1u) return fail(MSDR_STATUS_LENGTH_ERROR, "AudioFilterBiquad processes sample pairs: blockSize must be even");
    if (S->max_stage == 0 && (blockSize & 127u) == 0 && (reinterpret_cast<uintptr_t>(d_data) & 15) == 0 && S->pipe_ch) {
        // one stage, slab-shaped batch: the recursion alone on one wave, the input products element-wise on the others (msdr_kernels.hiph)
        const int per_group = S->pipe_ch;
#define MSDR_TQ4_LAUNCH(CH_)                                                                                                                               \
        if (per_group == CH_ && S->channels % CH_ == 0) {                                                                                                  \
            hipLaunchKernelGGL((biquad_teensy_pipe4_kernel<1, CH_>), dim3(S->channels / CH_), dim3(tq4_threads(CH_)), tq4_lds_bytes(CH_), S->ctx->stream,  \
                               (short *)d_data, S->d_defs, (int *)nullptr, (int)S->channels, (long long)blockSize);                                        \
            return launch_check("biquad_teensy_pipe4_kernel<1>");                                                                                          \
        }
        MSDR_TQ4_LAUNCH(64) MSDR_TQ4_LAUNCH(32) MSDR_TQ4_LAUNCH(16)
#undef MSDR_TQ4_LAUNCH
    }
    hipLaunchKernelGGL((biquad_teensy_kernel<1>), dim3((S->channels + 63) / 64), dim3(64), 0, S->ctx->stream, d_data, S->d_defs,
                       (int *)nullptr, (int)S->channels, (long long)blockSize);
    return launch_check("biquad_teensy_kernel");
}
extern "C" int msdr_biquad_q15_get_definition(msdr_biquad_q15 *S, uint32_t channel, int32_t definition[32])
{
    if (!S || !definition || channel >= S->channels) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad argument");
    return msdr_memcpy_d2h(S->ctx, definition, S->d_defs + (size_t)channel * 32, 32 * sizeof(int32_t));
}
extern "C" int msdr_biquad_q15_destroy(msdr_biquad_q15 *S)
{
    if (!S) return 0;
    if (int rc = bind(S->ctx)) return rc;
    (void)hipStreamSynchronize(S->ctx->stream);
    hipFree(S->d_defs);
    delete S;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stateless stages
// ------------------------------------------------------------------------------------------------
extern "C" int msdr_mix_fs4_q15(msdr_ctx *ctx, const q15_t *d_x, q15_t *d_i, q15_t *d_q, uint32_t channels, uint32_t blockSize)
{
    if (int rc = bind(ctx)) return rc;
    if (!d_x || !d_i || !d_q) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    long long total = (long long)channels * blockSize;
    if (total == 0) return 0;
    hipLaunchKernelGGL(mix_fs4_q15_kernel, dim3(grid_1d(total)), dim3(256), 0, ctx->stream, d_x, d_i, d_q, total, (int)blockSize);
    return launch_check("mix_fs4_q15_kernel");
}

template <typename T, typename K>
static int freqconv_common(msdr_ctx *ctx, T *d_i, T *d_q, const T *osc_i, const T *osc_q, uint32_t osc_len, int dir, int pass,
                           uint32_t channels, uint32_t blockSize, K kernel, const char *name)
{
    if (int rc = bind(ctx)) return rc;
    if (!d_i || !d_q) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");   // freq_conv.cpp:40-47: nothing transmitted
    if (!pass) return 0;                                                        // :49-56: forwarded untouched
    if (!osc_i || !osc_q || osc_len == 0) return fail(MSDR_STATUS_ARGUMENT_ERROR, "oscillator tables missing");
    long long total = (long long)channels * blockSize;
    if (total == 0) return 0;
    // the two host tables go through the context's scratch buffer (grown on demand, reused between calls)
    const size_t need = 2 * (size_t)osc_len * sizeof(T);
    if (ctx->scratch_bytes < need) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr; ctx->scratch_bytes = 0;
        HIP_TRY(hipMalloc(&ctx->scratch, std::max<size_t>(need, 4096)));
        ctx->scratch_bytes = std::max<size_t>(need, 4096);
    }
    T *di = (T *)ctx->scratch, *dq = di + osc_len;
    HIP_TRY(hipMemcpyAsync(di, osc_i, osc_len * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dq, osc_q, osc_len * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(kernel, dim3(grid_1d(total)), dim3(256), 0, ctx->stream, d_i, d_q, (const T *)di, (const T *)dq,
                       (int)osc_len, dir ? 1 : 0, total, (int)blockSize);
    return launch_check(name);
}
extern "C" int msdr_freqconv_q15(msdr_ctx *ctx, q15_t *d_i, q15_t *d_q, const q15_t *osc_i, const q15_t *osc_q, uint32_t osc_len,
                                 int dir, int pass, uint32_t channels, uint32_t blockSize)
{
    return freqconv_common<short>(ctx, d_i, d_q, osc_i, osc_q, osc_len, dir, pass, channels, blockSize, freqconv_q15_kernel,
                                  "freqconv_q15_kernel");
}
extern "C" int msdr_freqconv_f32(msdr_ctx *ctx, float32_t *d_i, float32_t *d_q, const float32_t *osc_i, const float32_t *osc_q,
                                 uint32_t osc_len, int dir, int pass, uint32_t channels, uint32_t blockSize)
{
    return freqconv_common<float>(ctx, d_i, d_q, osc_i, osc_q, osc_len, dir, pass, channels, blockSize, freqconv_f32_kernel,
                                  "freqconv_f32_kernel");
}
extern "C" int msdr_demod_q15(msdr_ctx *ctx, int mode, const int32_t *d_mode, int sqrt_kind, const q15_t *d_i, const q15_t *d_q,
                              q15_t *d_out, uint32_t channels, uint32_t blockSize)
{
    if (int rc = bind(ctx)) return rc;
    if (!d_i || !d_q || !d_out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    long long total = (long long)channels * blockSize;
    if (total == 0) return 0;
    hipLaunchKernelGGL(demod_q15_kernel, dim3(grid_1d(total)), dim3(256), 0, ctx->stream, mode, d_mode, sqrt_kind, d_i, d_q, d_out,
                       total, (int)blockSize);
    return launch_check("demod_q15_kernel");
}
extern "C" int msdr_demod_f32(msdr_ctx *ctx, int mode, const int32_t *d_mode, const float32_t *d_i, const float32_t *d_q,
                              float32_t *d_out, uint32_t channels, uint32_t blockSize)
{
    if (int rc = bind(ctx)) return rc;
    if (!d_i || !d_q || !d_out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    long long total = (long long)channels * blockSize;
    if (total == 0) return 0;
    hipLaunchKernelGGL(demod_f32_kernel, dim3(grid_1d(total)), dim3(256), 0, ctx->stream, mode, d_mode, d_i, d_q, d_out, total,
                       (int)blockSize);
    return launch_check("demod_f32_kernel");
}

// ------------------------------------------------------------------------------------------------
// front end: DC block + AudioAmplifier + AGC
// ------------------------------------------------------------------------------------------------
struct msdr_frontend {
    msdr_ctx *ctx;
    uint32_t channels;
    int *d_state;          // [channels][kFeStateInts]
    int pipe_ch;           // channels per workgroup of frontend_pipe4_kernel (64 / 32 / 16), 0: the two-wave pipeline, -1: no pipeline (read at create time)
};

static int fe_edit(msdr_frontend *fe, const std::function<void(uint32_t, int *)> &f)
{
    HIP_TRY(hipStreamSynchronize(fe->ctx->stream));
    std::vector<int> h((size_t)fe->channels * kFeStateInts);
    HIP_TRY(hipMemcpy(h.data(), fe->d_state, h.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (uint32_t ch = 0; ch < fe->channels; ch++) f(ch, h.data() + (size_t)ch * kFeStateInts);
    HIP_TRY(hipMemcpy(fe->d_state, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int32_t msdr_amp_multiplier(float n) { return fe_multiplier(n); }

extern "C" int msdr_frontend_create(msdr_ctx *ctx, uint32_t channels, msdr_frontend **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (int rc = bind(ctx)) return rc;
    if (channels == 0) return fail(MSDR_STATUS_ARGUMENT_ERROR, "channels == 0");
    msdr_frontend *fe = new (std::nothrow) msdr_frontend();
    if (!fe) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    fe->ctx = ctx; fe->channels = channels; fe->d_state = nullptr;
    // One workgroup walks its channels' stream alone and its step time is the recursion's latency whether 64, 32 or 16 lanes of the recursion
    // wave carry a channel -- only the element-wise work beside it grows with the channel count.  So: as few channels per workgroup as still
    // give every CU one (MI355X: 256 CUs); from 16 384 channels on the CUs are full and 64 per workgroup issue the fewest instructions.
    fe->pipe_ch = channels >= 16384 ? 64 : channels >= 8192 ? 32 : 16;
    if (const char *e = getenv("MSDR_FRONTEND_PIPE_CH")) { const int v = atoi(e); if (v == 64 || v == 32 || v == 16) fe->pipe_ch = v; }
    std::vector<int> h((size_t)channels * kFeStateInts, 0);
    const float agc_start = 0.25f;                       // Minimal-SDR.ino:94
    int bits; memcpy(&bits, &agc_start, sizeof bits);
    for (uint32_t ch = 0; ch < channels; ch++) {
        int *s = h.data() + (size_t)ch * kFeStateInts;
        s[2] = fe_multiplier(agc_start); s[3] = kFeAgcBuf; s[4] = bits; s[5] = 1;
    }
    if (int rc = upload(ctx, h, &fe->d_state)) { delete fe; return rc; }
    *out = fe;
    return 0;
}
extern "C" int msdr_frontend_prime(msdr_frontend *fe, const uint16_t *first_conversion, uint32_t count)
{
    if (!fe || !first_conversion) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(fe->ctx)) return rc;
    if (count != 1 && count != fe->channels) return fail(MSDR_STATUS_ARGUMENT_ERROR, "count must be 1 or channels");
    return fe_edit(fe, [&](uint32_t ch, int *s) { s[1] = (int)((uint32_t)first_conversion[count == 1 ? 0 : ch] << 14); s[0] = 0; });
}
extern "C" int msdr_frontend_set_agc(msdr_frontend *fe, int on)
{
    if (!fe) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(fe->ctx)) return rc;
    return fe_edit(fe, [&](uint32_t, int *s) { s[5] = on ? 1 : 0; });
}
extern "C" int msdr_frontend_gain(msdr_frontend *fe, float n)
{
    if (!fe) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(fe->ctx)) return rc;
    int bits; memcpy(&bits, &n, sizeof bits);
    return fe_edit(fe, [&](uint32_t, int *s) { s[4] = bits; s[2] = fe_multiplier(n); });
}
extern "C" int msdr_frontend_update(msdr_frontend *fe, const void *d_adc, q15_t *d_out, uint32_t blockSize, uint32_t stages)
{
    if (!fe) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(fe->ctx)) return rc;
    if (blockSize == 0) return 0;
    if (!d_adc || !d_out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    if (blockSize % 128u) return fail(MSDR_STATUS_LENGTH_ERROR, "the front end runs in AUDIO_BLOCK_SAMPLES = 128 blocks: blockSize %u is not a multiple", blockSize);
    if (stages & ~MSDR_FE_ALL) return fail(MSDR_STATUS_ARGUMENT_ERROR, "unknown stage bits");
    if (stages == MSDR_FE_ALL && fe->pipe_ch >= 0 && (reinterpret_cast<uintptr_t>(d_adc) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0) {
        // the recursion alone on one wave, everything element-wise on the others (msdr_frontend.hiph)
#define MSDR_FE4_LAUNCH(CH_)                                                                                                                              \
        if (fe->pipe_ch == CH_ && fe->channels % CH_ == 0) {                                                                                              \
            hipLaunchKernelGGL(frontend_pipe4_kernel<CH_>, dim3(fe->channels / CH_), dim3(fe4_threads(CH_)), fe4_lds_bytes(CH_), fe->ctx->stream,         \
                               (const unsigned short *)d_adc, (short *)d_out, fe->d_state, (int)fe->channels, (long long)blockSize);                      \
            return launch_check("frontend_pipe4_kernel");                                                                                                 \
        }
        MSDR_FE4_LAUNCH(64) MSDR_FE4_LAUNCH(32) MSDR_FE4_LAUNCH(16)
#undef MSDR_FE4_LAUNCH
        if ((fe->channels & 63u) == 0) {   // DC block on one wave, gain + AGC on the next: a two-wave slab pipeline
            hipLaunchKernelGGL(frontend_pipe_kernel, dim3(fe->channels / 64), dim3(128), 0, fe->ctx->stream, (const unsigned short *)d_adc,
                               (short *)d_out, fe->d_state, (int)fe->channels, (long long)blockSize);
            return launch_check("frontend_pipe_kernel");
        }
    }
    hipLaunchKernelGGL(frontend_kernel, dim3((fe->channels + 63) / 64), dim3(64), 0, fe->ctx->stream, (const unsigned short *)d_adc,
                       (short *)d_out, fe->d_state, (int)fe->channels, (long long)blockSize, (int)stages);
    return launch_check("frontend_kernel");
}
extern "C" int msdr_frontend_get_state(msdr_frontend *fe, uint32_t channel, int32_t state[MSDR_FE_STATE_WORDS])
{
    if (!fe || !state) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(fe->ctx)) return rc;
    if (channel >= fe->channels) return fail(MSDR_STATUS_ARGUMENT_ERROR, "channel out of range");
    HIP_TRY(hipStreamSynchronize(fe->ctx->stream));
    HIP_TRY(hipMemcpy(state, fe->d_state + (size_t)channel * kFeStateInts, kFeStateInts * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int msdr_frontend_destroy(msdr_frontend *fe)
{
    if (!fe) return 0;
    if (int rc = bind(fe->ctx)) return rc;
    (void)hipStreamSynchronize(fe->ctx->stream);
    hipFree(fe->d_state);
    delete fe;
    return 0;
}
extern "C" int msdr_amp_q15(msdr_ctx *ctx, int32_t multiplier, q15_t *d_data, uint32_t channels, uint32_t blockSize, int *transmitted)
{
    if (int rc = bind(ctx)) return rc;
    if (transmitted) *transmitted = (multiplier != 0);
    if (multiplier == 0 || multiplier == 65536 || channels == 0 || blockSize == 0) return 0;     // mixer.cpp:139-149
    if (!d_data) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    const long long total = (long long)channels * blockSize;
    hipLaunchKernelGGL(amp_q15_kernel, dim3(grid_1d(total)), dim3(256), 0, ctx->stream, (short *)d_data, total, (int)multiplier);
    return launch_check("amp_q15_kernel");
}

// ------------------------------------------------------------------------------------------------
// synchronous AM PLL
// ------------------------------------------------------------------------------------------------
struct msdr_syncam {
    msdr_ctx *ctx;
    uint32_t channels;
    float *d_state;        // [channels][4]
};

extern "C" void msdr_syncam_constants(float c[4])
{
    // the initialisers of Minimal-SDR.ino:637-642 under C's usual arithmetic conversions (PI = Arduino.h's double literal)
    const double PI_ = 3.1415926535897932384626433832795;
    const float omegaN = 400.0, zeta = 0.45;
    const int SAMPLE_RATE_ = 24000;
    c[0] = (float)(2.0 * PI_ * -4000.0 / SAMPLE_RATE_);
    c[1] = (float)(2.0 * PI_ * 4000.0 / SAMPLE_RATE_);
    const float g1 = (float)(1.0 - std::exp(-2.0 * (double)omegaN * (double)zeta / SAMPLE_RATE_));
    const float e_arg = -omegaN * zeta / (float)SAMPLE_RATE_;
    const float c_arg = omegaN / (float)SAMPLE_RATE_ * sqrtf((float)(1.0 - (double)(zeta * zeta)));
    c[2] = g1;
    c[3] = (float)(-(double)g1 + 2.0 * (1 - std::exp((double)e_arg) * (double)cosf(c_arg)));
}
extern "C" int msdr_syncam_create(msdr_ctx *ctx, uint32_t channels, msdr_syncam **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (int rc = bind(ctx)) return rc;
    if (channels == 0) return fail(MSDR_STATUS_ARGUMENT_ERROR, "channels == 0");
    msdr_syncam *S = new (std::nothrow) msdr_syncam();
    if (!S) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    S->ctx = ctx; S->channels = channels; S->d_state = nullptr;
    if (int rc = dzalloc(ctx, (size_t)channels * 4, &S->d_state)) { delete S; return rc; }
    *out = S;
    return 0;
}
extern "C" int msdr_syncam_q15(msdr_syncam *S, const int32_t *d_mode, const q15_t *d_I, const q15_t *d_Q, q15_t *d_out, uint32_t blockSize)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(S->ctx)) return rc;
    if (blockSize == 0) return 0;
    if (!d_I || !d_Q || !d_out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    SyncamConst k;
    float c[4];
    msdr_syncam_constants(c);
    k.omega_min = c[0]; k.omega_max = c[1]; k.g1 = c[2]; k.g2 = c[3];
    hipLaunchKernelGGL(syncam_pll_kernel, dim3((S->channels + 63) / 64), dim3(64), 0, S->ctx->stream, (const short *)d_I, (const short *)d_Q,
                       (short *)d_out, S->d_state, (const int *)d_mode, (int)S->channels, (long long)blockSize, k);
    return launch_check("syncam_pll_kernel");
}
extern "C" int msdr_syncam_reset(msdr_syncam *S)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(S->ctx)) return rc;
    HIP_TRY(hipMemsetAsync(S->d_state, 0, (size_t)S->channels * 4 * sizeof(float), S->ctx->stream));
    return 0;
}
extern "C" int msdr_syncam_get_state(msdr_syncam *S, uint32_t channel, float state[3])
{
    if (!S || !state) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(S->ctx)) return rc;
    if (channel >= S->channels) return fail(MSDR_STATUS_ARGUMENT_ERROR, "channel out of range");
    HIP_TRY(hipStreamSynchronize(S->ctx->stream));
    HIP_TRY(hipMemcpy(state, S->d_state + (size_t)channel * 4, 3 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int msdr_syncam_destroy(msdr_syncam *S)
{
    if (!S) return 0;
    if (int rc = bind(S->ctx)) return rc;
    (void)hipStreamSynchronize(S->ctx->stream);
    hipFree(S->d_state);
    delete S;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LMS automatic notch / noise reduction
// ------------------------------------------------------------------------------------------------
struct msdr_anr {
    msdr_ctx *ctx;
    uint32_t channels;
    float *d_state;        // [channels][kAnrStateFloats]
};
static int anr_init_state(msdr_anr *A)
{
    std::vector<float> h((size_t)A->channels * kAnrStateFloats, 0.0f);
    for (uint32_t ch = 0; ch < A->channels; ch++) { h[(size_t)ch * kAnrStateFloats] = 120.0f; h[(size_t)ch * kAnrStateFloats + 1] = 0.001f; }   // .ino:715,:718
    HIP_TRY(hipMemcpyAsync(A->d_state, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, A->ctx->stream));
    HIP_TRY(hipStreamSynchronize(A->ctx->stream));
    return 0;
}
extern "C" int msdr_anr_create(msdr_ctx *ctx, uint32_t channels, msdr_anr **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (int rc = bind(ctx)) return rc;
    if (channels == 0) return fail(MSDR_STATUS_ARGUMENT_ERROR, "channels == 0");
    static_assert(kAnrStateFloats == MSDR_ANR_STATE_FLOATS, "state record size");
    msdr_anr *A = new (std::nothrow) msdr_anr();
    if (!A) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    A->ctx = ctx; A->channels = channels; A->d_state = nullptr;
    if (int rc = dzalloc(ctx, (size_t)channels * kAnrStateFloats, &A->d_state)) { delete A; return rc; }
    if (int rc = anr_init_state(A)) { hipFree(A->d_state); delete A; return rc; }
    *out = A;
    return 0;
}
extern "C" int msdr_anr_q15(msdr_anr *A, const int32_t *d_anr_on, int32_t anr_on_all, q15_t *d_data, uint32_t blockSize)
{
    if (!A) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(A->ctx)) return rc;
    if (blockSize == 0) return 0;
    if (!d_data) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    if (!d_anr_on && anr_on_all <= 0) return 0;                       // .ino:702
    hipLaunchKernelGGL(anr_kernel, dim3((A->channels + 63) / 64), dim3(64), 0, A->ctx->stream, (short *)d_data, A->d_state,
                       (const int *)d_anr_on, (int)anr_on_all, (int)A->channels, (long long)blockSize);
    return launch_check("anr_kernel");
}
extern "C" int msdr_anr_reset(msdr_anr *A)
{
    if (!A) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null instance");
    if (int rc = bind(A->ctx)) return rc;
    return anr_init_state(A);
}
extern "C" int msdr_anr_get_state(msdr_anr *A, uint32_t channel, float state[MSDR_ANR_STATE_FLOATS])
{
    if (!A || !state) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(A->ctx)) return rc;
    if (channel >= A->channels) return fail(MSDR_STATUS_ARGUMENT_ERROR, "channel out of range");
    HIP_TRY(hipStreamSynchronize(A->ctx->stream));
    HIP_TRY(hipMemcpy(state, A->d_state + (size_t)channel * kAnrStateFloats, kAnrStateFloats * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int msdr_anr_destroy(msdr_anr *A)
{
    if (!A) return 0;
    if (int rc = bind(A->ctx)) return rc;
    (void)hipStreamSynchronize(A->ctx->stream);
    hipFree(A->d_state);
    delete A;
    return 0;
}

extern "C" int msdr_dac_format_q15(msdr_ctx *ctx, const q15_t *d_src, q15_t *d_dest, uint32_t channels, uint32_t blockSize)
{
    if (int rc = bind(ctx)) return rc;
    const long long total = (long long)channels * blockSize;
    if (total == 0) return 0;
    if (!d_dest) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    hipLaunchKernelGGL(dac_format_kernel, dim3(grid_1d((total + 7) / 8)), dim3(256), 0, ctx->stream, (const short *)d_src, (short *)d_dest, total);
    return launch_check("dac_format_kernel");
}

// ------------------------------------------------------------------------------------------------
// row f4: spectrum FFT (UI.cpp:520-592)
// ------------------------------------------------------------------------------------------------
extern "C" void msdr_rfft128_tables(int16_t tables[352]) { msdr::design::fft128_tables(tables); }
extern "C" int msdr_rfft_q15_init_check(uint32_t fftLenReal, uint32_t ifftFlagR, uint32_t bitReverseFlag)
{
    bool valid = false;
    for (uint32_t n = 32; n <= 8192; n <<= 1) valid |= (n == fftLenReal);      // arm_rfft_init_q15.c:2179-2221
    if (!valid) return fail(MSDR_STATUS_ARGUMENT_ERROR, "fftLenReal %u is not a supported RFFT length", fftLenReal);
    if (fftLenReal != 128 || ifftFlagR != 0 || bitReverseFlag != 1)
        return fail(MSDR_STATUS_LENGTH_ERROR, "only the 128-point forward transform with bit reversal (initSpectrum, UI.cpp:523) is built");
    return 0;
}
extern "C" int msdr_rfft128_q15(msdr_ctx *ctx, const q15_t *d_src, uint64_t src_stride, q15_t *d_fft_out, uint8_t *d_columns, uint32_t nfft)
{
    if (int rc = bind(ctx)) return rc;
    if (nfft == 0) return 0;
    if (!d_src) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    if ((src_stride & 7) || (reinterpret_cast<uintptr_t>(d_src) & 15) || (reinterpret_cast<uintptr_t>(d_fft_out) & 15) ||
        (reinterpret_cast<uintptr_t>(d_columns) & 7))
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "rfft128: src/fft_out must be 16-byte aligned, columns 8-byte aligned, src_stride a multiple of 8");
    if (nfft > 1 && src_stride < 128) return fail(MSDR_STATUS_ARGUMENT_ERROR, "rfft128: src_stride < 128 with more than one transform");
    if (!ctx->d_fft_tables) {
        std::vector<int16_t> h(kFftTableShorts);
        msdr::design::fft128_tables(h.data());
        if (int rc = upload(ctx, h, &ctx->d_fft_tables)) return rc;
    }
    if (!d_fft_out && !d_columns) return 0;
    const int grid = (int)std::min<long long>(((long long)nfft + kFftPerBlock - 1) / kFftPerBlock, 256LL * 32);
    hipLaunchKernelGGL(spectrum_rfft128_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const short *)d_src, (long long)src_stride,
                       (short *)d_fft_out, (unsigned char *)d_columns, (const short *)ctx->d_fft_tables, (int)nfft);
    return launch_check("spectrum_rfft128_kernel");
}

struct msdr_spectrum {
    msdr_ctx *ctx;
    uint32_t channels;
    int spectrum_on;        // Spectrum_on
    int counter;            // spectrumCounter (UI.cpp:122: starts at 0, so the first call draws)
};
extern "C" int msdr_spectrum_create(msdr_ctx *ctx, uint32_t channels, msdr_spectrum **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (int rc = bind(ctx)) return rc;
    if (int rc = msdr_rfft_q15_init_check(128, 0, 1)) return rc;                 // initSpectrum(), UI.cpp:523
    msdr_spectrum *S = new (std::nothrow) msdr_spectrum();
    if (!S) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    S->ctx = ctx; S->channels = channels; S->spectrum_on = 1; S->counter = 0;
    *out = S;
    return 0;
}
extern "C" int msdr_spectrum_set_on(msdr_spectrum *S, int spectrum_on)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    S->spectrum_on = spectrum_on ? 1 : 0;
    return 0;
}
extern "C" int msdr_spectrum_show(msdr_spectrum *S, const q15_t *d_data, uint64_t channel_stride, q15_t *d_fft_out, uint8_t *d_columns, int *drawn)
{
    if (!S) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (drawn) *drawn = 0;
    if (!S->spectrum_on) return 0;                                               // UI.cpp:533
    if (--S->counter > 0) return 0;                                              // :534
    S->counter = 25;                                                             // :535
    if (int rc = msdr_rfft128_q15(S->ctx, d_data, channel_stride, d_fft_out, d_columns, S->channels)) return rc;
    if (drawn) *drawn = 1;
    return 0;
}
extern "C" int msdr_spectrum_destroy(msdr_spectrum *S)
{
    delete S;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fused chain
// ------------------------------------------------------------------------------------------------
// A deep host copy of the configuration a chain was created from: the live updates (msdr_chain_set_taps / set_osc /
// set_biquad_coeffs) edit it and rebuild the chain's device tables through msdr_chain_create itself -- one table builder, not two.
struct ChainCfgStore {
    msdr_chain_config cfg;                            // scalars; the pointers inside are not valid, view() fills them in
    std::vector<char> ci[MSDR_MAX_TAPSETS], cq[MSDR_MAX_TAPSETS], osc_i, osc_q;
    std::vector<float> bq;
    std::vector<int32_t> node[2];
    void capture(const msdr_chain_config *c)
    {
        cfg = *c;
        const size_t el = (c->arith == MSDR_ARITH_F32) ? sizeof(float) : sizeof(int16_t);
        for (uint32_t s = 0; s < c->num_tapsets && s < MSDR_MAX_TAPSETS; s++) {
            ci[s].assign((const char *)c->coeffs_i[s], (const char *)c->coeffs_i[s] + el * c->num_taps);
            cq[s].assign((const char *)c->coeffs_q[s], (const char *)c->coeffs_q[s] + el * c->num_taps);
        }
        osc_i.clear(); osc_q.clear();
        if (c->mixer == MSDR_MIXER_NCO && c->osc_i && c->osc_q) {
            osc_i.assign((const char *)c->osc_i, (const char *)c->osc_i + el * c->osc_len);
            osc_q.assign((const char *)c->osc_q, (const char *)c->osc_q + el * c->osc_len);
        }
        bq.clear();
        if (c->arith == MSDR_ARITH_F32 && c->num_biquad_stages && c->biquad_coeffs) bq.assign(c->biquad_coeffs, c->biquad_coeffs + 5 * c->num_biquad_stages);
        for (int k = 0; k < 2; k++) {
            node[k].clear();
            if (c->arith == MSDR_ARITH_Q15 && (uint32_t)k < c->num_biquad_nodes && c->node_coefs[k]) node[k].assign(c->node_coefs[k], c->node_coefs[k] + 5 * c->node_stages[k]);
        }
    }
    void view(msdr_chain_config *out, const std::vector<int> &mode, const std::vector<int> &tapset) const
    {
        *out = cfg;
        for (uint32_t s = 0; s < MSDR_MAX_TAPSETS; s++) {
            out->coeffs_i[s] = (s < cfg.num_tapsets) ? (const void *)ci[s].data() : nullptr;
            out->coeffs_q[s] = (s < cfg.num_tapsets) ? (const void *)cq[s].data() : nullptr;
        }
        out->osc_i = osc_i.empty() ? nullptr : (const void *)osc_i.data();
        out->osc_q = osc_q.empty() ? nullptr : (const void *)osc_q.data();
        out->biquad_coeffs = bq.empty() ? nullptr : bq.data();
        for (int k = 0; k < 2; k++) out->node_coefs[k] = node[k].empty() ? nullptr : node[k].data();
        out->mode = mode.data(); out->tapset = tapset.data();
    }
};

struct msdr_chain {
    msdr_ctx *ctx;
    ChainCfgStore store;
    double own_d_bound, own_sig_bound;   // what this chain's own tap sets can leave in the cascade's state (floors for a rebuilt chain's table scales)
    // ... and the floors THIS chain was built over, for as long as no sample has been processed since (floor_gen == gen): the state it carries is
    // still its predecessor's, so a second live update in a row must size its tables for that predecessor too (tests/debug/fuzz_live.py seed
    // 5202 case 44155: taps 20 x smaller, then a cascade rewrite, no call in between -- the second rebuild scaled for the small taps only and the
    // state of the large ones went through fp16 out of range: 1.1e-3 on the SSB channels' next block)
    double floor_d = 0.0, floor_sig = 0.0;
    uint64_t floor_gen = ~0ull;
    // msdr_chain_set_osc: the tables that were in force when the samples still in the FIR history arrived, oldest first.  While any is
    // pending the chain runs chain_kernel<Arith>, which mixes every history sample with the table of its own time.
    struct OscPending { void *d_tab; long long elapsed; };
    std::vector<OscPending> osc_pending;
    bool force_generic = false;          // the as-written kernel for the time being: no mode counts as numerator-folded
    OscHistory *d_osc_hist = nullptr;    // the pending tables as the kernel reads them
    float *d_f32_scratch = nullptr;      // MSDR_CHAIN_OUT_I16 where the int16 conversion cannot happen in the main kernel: fp32 audio block batch
    size_t f32_scratch_floats = 0;
    int arith, mixer, sqrt_kind;
    uint32_t channels, ntaps, ntaps_pad, hist_len, tapsets;
    uint32_t osc_len;
    float in_scale;
    uint32_t nstages;                 // F32
    uint32_t nnodes;                  // Q15
    uint32_t time_segments, warmup_cfg;
    double pole_radius;
    void *d_taps, *d_osc;
    int *d_mode, *d_tapset;
    int16_t *d_hist[2];
    int cur;
    long long phase;                  // absolute sample index modulo the NCO period
    BiquadCascadeTables<kChainR> *d_bq;
    float *d_bq_state;
    // folded F32 path (msdr_chain_fold.hiph)
    int fold_P;                       // 0 = not foldable; else NCO period 1, 2 or 4
    int mf_P;                         // oscillator period the matrix-core tables are built for (1 .. 32, a divisor of the 32-sample output row); 0 = none
    bool fold_fs4_exact;              // oscillator is exactly {1,0,-1,0}/{0,1,0,-1}: AM can be folded too
    float *d_ftaps;                   // [tapsets*3][P rotations][steps][PE rows][2 pair-columns][2]
    int *d_fset;                      // [channels]
    BiquadCascadeTables<kFoldR> *d_bq_fold;
    std::vector<int> h_mode, h_tapset;
    // matrix-core path (msdr_chain_mfma.hiph): any short-period oscillator, any mode
    bool mf_ok;
    int mf_halo, mf_bsteps, mf_stride;
    char *d_mf_tab;
    BiquadCascadeTables<kMfL> *d_bq_mf;
    BiquadCascadeTables<32> *d_bq_mf32;
    // wave-stream variant (msdr_chain_mfw.hiph): waves per workgroup, resident waves per CU, unit table and its cache key
    uint32_t flags;
    int mfw_nw, mfw_waves_per_cu;
    char *d_at_tab; int at_ns, at_stride, at_nw;          // envelope channels with the taps in registers (msdr_chain_amtr.hiph), or null
    bool mf_fr;                                           // full-rate layout (msdr_chain_mfw.hiph): any 128-periodic oscillator table, mixer products staged as two streams
    float *d_bq_state_alt;
    float *d_mw_iir;                  // folded-IIR constants (MwIirConsts) or null
    bool mfw_ssb_fold, mfw_am_fold;   // cascade as matrix products for SSB tables / envelope tables
    uint32_t units_wgs_ssb;           // unit table: workgroups of SSB-table units come first, envelope-table units after them
    long long part_nseg[2], part_seg_len[2];      // the two launches choose their own time segmentation (each must fill the GPU on its own)
    long long units_tiles;
    int *d_units;
    size_t units_cap;
    uint64_t mode_gen, units_mode_gen;
    long long units_nseg;
    uint32_t units_wgs;
    // block cadence (msdr_chain_mfb.hiph): tile table (channels per tile, grouped by table set; SSB-table part first), cached per (modes, n)
    int *d_btiles = nullptr;
    size_t btiles_cap = 0;
    uint64_t btiles_mode_gen = 0;
    long long btiles_n = -1;
    struct BlockPart { uint32_t wgs = 0, nw = 0, tpw = 0; size_t offset = 0; } bpart[2];
    // what the per-call kernel choice needs to know about the channels' modes / tap sets / LMS switches, recomputed when they change (a
    // call at block cadence must not walk 65 536 channels on the host)
    struct Summary { uint64_t mode_gen = 0, anr_gen = 0; bool any_ssb = false, any_env = false, any_syncam = false, qm_sets_ok = true, any_anr = false; } sum;
    bool dry_run = false;                // msdr_chain_graph_create: msdr_chain_process prepares a block-cadence call (tables, caches) and returns before its launches
    bool block_off = false;              // MSDR_NO_BLOCK=1 at create time: keep the wave-stream kernels at every call length (A/B runs, tests)
    bool no_fuse = false;                // MSDR_Q15_NO_FUSE=1 at create time: the biquad nodes as a kernel behind chain_q15mb_kernel, not its second phase
    int blk_force = -1;                  // MSDR_BIQUAD_BLK at create time (0: never biquad_teensy_blk_kernel, 1: always where it applies; -1: the host's rule)
    std::vector<std::vector<float>> h_coef_i, h_coef_q;   // host copies for msdr_chain_set_mode
    std::vector<double> h_osc, h_cnum;                    // oscillator pairs {cos, sin}; combined numerator
    struct DHist { double v[8]; };
    std::vector<DHist> dh_cache;                          // true numerator history per channel, valid while dh_gen == gen
    std::vector<uint64_t> dh_gen;
    uint64_t gen;                                         // process calls so far + 1
    // Q15 matrix-core kernel (msdr_chain_q15mf.hiph): tables per (tap set, phase), channels ordered by tap set
    char *d_qm_tab;
    int qm_stride, qm_halo, qm_bsteps;
    bool qm_fr;                       // full-rate layout: any 128-periodic freq_conv table (both filters over every sample)
    std::vector<char> qm_set_ok;      // per tap set: table built (every tap < 32640)
    int *d_qm_order;                  // channels grouped by tap set
    std::vector<uint32_t> qm_group_start, qm_group_count;
    uint64_t qm_order_gen;
    msdr_biquad_df1_f32 *seq_bq;       // F32: the cascade is ill-conditioned for the parallel evaluation -> applied behind the main kernel in CMSIS order
    msdr_biquad_q15 *nodes[2];
    msdr_syncam *pll;                 // Q15 + MSDR_CHAIN_SYNCAM_PLL: the PLL demodulator of SYNCAM channels and its Q scratch
    int16_t *d_pll_q;
    size_t pll_q_cap;
    msdr_anr *anr;                    // Q15: LMS notch / noise reduction between demodulator and biquad nodes (msdr_chain_set_anr)
    int *d_anr_on;
    int anr_all;
    // F32: rows f2 / f3 inside the fp32 chain ("post channels": SYNCAM channels under MSDR_CHAIN_SYNCAM_PLL, channels with the LMS filter
    // on).  Their FIR outputs come from an auxiliary chain without biquads over the gathered IF rows; PLL, LMS filter and the cascade
    // (arm_biquad_cascade_df1_f32 stage) run behind it and the result replaces the main kernel's rows.  See chain_post_run().
    bool f32_pll;
    std::vector<int> h_anr;           // per channel: 0 off, 1 notch, 2 noise reduction
    std::vector<float> h_bq;          // 5 * h_bq_stages biquad coefficients as given at creation
    uint32_t h_bq_stages;
    uint64_t anr_gen, post_mode_gen, post_anr_gen;
    msdr_chain *aux;
    msdr_biquad_df1_f32 *post_bq;
    int npost, nvirt;
    int *d_post_ch, *d_post_src, *d_post_pll, *d_post_anr, *d_virt_row;
    std::vector<int> h_post_ch;
    float *d_post_pll_state, *d_post_anr_state;      // [channels][4], [channels][kAnrStateFloats]: indexed by CHANNEL, they survive a retune
    int16_t *d_aux_x; float *d_aux_y, *d_post_scratch; size_t post_cap_virt, post_cap_post;      // capacities in elements
    msdr_chain_info info;
    // optional per-launch timing of the main kernel
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double timed_ms;
    uint64_t timed_launches;
};

static void chain_post_free(msdr_chain *c)
{
    if (c->aux) msdr_chain_destroy(c->aux);
    if (c->post_bq) msdr_biquad_df1_f32_destroy(c->post_bq);
    c->aux = nullptr; c->post_bq = nullptr;
    hipFree(c->d_post_ch); hipFree(c->d_post_src); hipFree(c->d_post_pll); hipFree(c->d_post_anr); hipFree(c->d_virt_row);
    c->d_post_ch = c->d_post_src = c->d_post_pll = c->d_post_anr = c->d_virt_row = nullptr;
    c->npost = 0; c->nvirt = 0;
}

static void chain_free(msdr_chain *c)
{
    if (!c) return;
    chain_post_free(c);
    hipFree(c->d_post_pll_state); hipFree(c->d_post_anr_state); hipFree(c->d_aux_x); hipFree(c->d_aux_y); hipFree(c->d_post_scratch);
    hipFree(c->d_taps); hipFree(c->d_osc); hipFree(c->d_mode); hipFree(c->d_tapset);
    hipFree(c->d_hist[0]); hipFree(c->d_hist[1]); hipFree(c->d_bq); hipFree(c->d_bq_state);
    hipFree(c->d_ftaps); hipFree(c->d_fset); hipFree(c->d_bq_fold);
    hipFree(c->d_mf_tab); hipFree(c->d_bq_mf); hipFree(c->d_bq_mf32);
    hipFree(c->d_bq_state_alt); hipFree(c->d_units); hipFree(c->d_mw_iir); hipFree(c->d_qm_tab); hipFree(c->d_qm_order); hipFree(c->d_at_tab);
    for (int k = 0; k < 2; k++) if (c->nodes[k]) msdr_biquad_q15_destroy(c->nodes[k]);
    if (c->seq_bq) msdr_biquad_df1_f32_destroy(c->seq_bq);
    if (c->pll) msdr_syncam_destroy(c->pll);
    hipFree(c->d_pll_q);
    if (c->anr) msdr_anr_destroy(c->anr);
    hipFree(c->d_anr_on);
    for (auto &e : c->events) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &o : c->osc_pending) hipFree(o.d_tab);
    hipFree(c->d_f32_scratch); hipFree(c->d_osc_hist); hipFree(c->d_btiles);
    delete c;
}

// floors for the state bounds of a chain that is being rebuilt under a running stream (chain_rebuild): the state the OLD tables left
// behind must fit the new tables' fp16 intake too
static thread_local double g_chain_floor_d = 0.0, g_chain_floor_sig = 0.0;
// set while chain_create_impl re-creates a chain whose cascade must run in CMSIS order for a reason only the finished chain shows
static thread_local bool g_chain_force_seq_cascade = false;

static int chain_create_impl(msdr_ctx *ctx, const msdr_chain_config *cfg, msdr_chain **out);
extern "C" int msdr_chain_create(msdr_ctx *ctx, const msdr_chain_config *cfg, msdr_chain **out)
{
    if (int rc = chain_create_impl(ctx, cfg, out)) return rc;
    (*out)->store.capture(cfg);
    return 0;
}
static int chain_create_impl(msdr_ctx *ctx, const msdr_chain_config *cfg, msdr_chain **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (int rc = bind(ctx)) return rc;
    if (!cfg || cfg->struct_size != sizeof(msdr_chain_config))
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "msdr_chain_config.struct_size mismatch (got %u, want %zu)",
                    cfg ? cfg->struct_size : 0u, sizeof(msdr_chain_config));
    const bool f32 = (cfg->arith == MSDR_ARITH_F32);
    if (cfg->arith != MSDR_ARITH_F32 && cfg->arith != MSDR_ARITH_Q15) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad arith");
    if (cfg->channels == 0) return fail(MSDR_STATUS_ARGUMENT_ERROR, "channels == 0");
    if (cfg->mixer != MSDR_MIXER_FS4 && cfg->mixer != MSDR_MIXER_NCO) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad mixer");
    if (cfg->num_taps == 0 || cfg->num_taps > 4096) return fail(MSDR_STATUS_ARGUMENT_ERROR, "num_taps must be 1..4096");
    if (!f32 && (cfg->num_taps & 1u))
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "arm_fir_init_q15: numTaps must be even (got %u)", cfg->num_taps);
    if (cfg->num_tapsets == 0 || cfg->num_tapsets > MSDR_MAX_TAPSETS) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad num_tapsets");
    for (uint32_t s = 0; s < cfg->num_tapsets; s++)
        if (!cfg->coeffs_i[s] || !cfg->coeffs_q[s]) return fail(MSDR_STATUS_ARGUMENT_ERROR, "tap set %u missing", s);
    if (cfg->mixer == MSDR_MIXER_NCO && (!cfg->osc_i || !cfg->osc_q || cfg->osc_len == 0))
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "NCO mixer needs osc tables");
    if (f32 && cfg->num_biquad_stages > kMaxStages) return fail(MSDR_STATUS_ARGUMENT_ERROR, "num_biquad_stages > 4");
    if (f32 && cfg->num_biquad_stages && !cfg->biquad_coeffs) return fail(MSDR_STATUS_ARGUMENT_ERROR, "biquad_coeffs missing");
    if (!f32 && cfg->num_biquad_nodes > 2) return fail(MSDR_STATUS_ARGUMENT_ERROR, "num_biquad_nodes > 2");
    auto mode_ok = [](int m) { return m >= MSDR_MODE_SYNCAM && m <= MSDR_MODE_CW; };
    if (!mode_ok(cfg->default_mode)) return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad default_mode");
    {
        const uint32_t known = MSDR_CHAIN_NO_TAP_FOLDING | MSDR_CHAIN_NO_FFT | MSDR_CHAIN_NO_MFMA | MSDR_CHAIN_MFMA_WG | MSDR_CHAIN_SYNCAM_PLL | MSDR_CHAIN_FOLD_ANY_PERIOD | MSDR_CHAIN_OUT_I16;
        if (cfg->flags & ~known) return fail(MSDR_STATUS_ARGUMENT_ERROR, "unknown bits 0x%x in msdr_chain_config.flags", cfg->flags & ~known);
    }

    if (f32 && cfg->num_biquad_stages &&
        (g_chain_force_seq_cascade || cascade_needs_cmsis_order(cfg->biquad_coeffs, (int)cfg->num_biquad_stages))) {
        const bool forced = g_chain_force_seq_cascade;            // (the stage object below must then be the CMSIS-order one too: it decides for itself otherwise)
        g_chain_force_seq_cascade = false;
        // the parallel "numerators first" evaluation would lose accuracy on this cascade (cascade_condition): build the chain without
        // it and run arm_biquad_cascade_df1_f32 as written behind the main kernel (one lane per channel)
        msdr_chain_config plain = *cfg;
        plain.num_biquad_stages = 0; plain.biquad_coeffs = nullptr;
        if (int rc = chain_create_impl(ctx, &plain, out)) return rc;
        g_biquad_force_sequential = forced;
        const int rcb = msdr_biquad_df1_f32_create(ctx, (uint8_t)cfg->num_biquad_stages, cfg->biquad_coeffs, cfg->channels, &(*out)->seq_bq);
        g_biquad_force_sequential = false;
        if (rcb) { msdr_chain_destroy(*out); *out = nullptr; return rcb; }
        (*out)->h_bq.assign(cfg->biquad_coeffs, cfg->biquad_coeffs + 5 * cfg->num_biquad_stages); (*out)->h_bq_stages = cfg->num_biquad_stages;
        return 0;
    }

    msdr_chain *c = new (std::nothrow) msdr_chain();
    if (!c) return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed");
    c->seq_bq = nullptr; c->own_d_bound = 0.0; c->own_sig_bound = 0.0;
    c->ctx = ctx; c->arith = cfg->arith; c->mixer = cfg->mixer; c->sqrt_kind = cfg->sqrt_kind;
    c->channels = cfg->channels; c->ntaps = cfg->num_taps; c->ntaps_pad = (cfg->num_taps + 3u) & ~3u;
    c->hist_len = c->ntaps_pad - 1; c->tapsets = cfg->num_tapsets;
    if (f32 && !(cfg->flags & MSDR_CHAIN_NO_MFMA) && mf_halo((int)(cfg->num_taps + 2 * cfg->num_biquad_stages)) <= 2048)      // the matrix-core kernel's window halo
        c->hist_len = std::max<uint32_t>(c->hist_len, (uint32_t)mf_halo((int)(cfg->num_taps + 2 * cfg->num_biquad_stages)));
    if (!f32 && !(cfg->flags & MSDR_CHAIN_NO_MFMA) && qm_halo((int)cfg->num_taps) <= 512)                   // the integer matrix-core kernels' window halo:
        c->hist_len = std::max<uint32_t>(c->hist_len, (uint32_t)qm_halo((int)cfg->num_taps));                  // whole 16-byte groups per history row (block cadence)
    c->osc_len = (cfg->mixer == MSDR_MIXER_NCO) ? cfg->osc_len : 4;
    c->in_scale = (cfg->in_scale == 0.0f) ? 1.0f / 32768.0f : cfg->in_scale;
    c->nstages = f32 ? cfg->num_biquad_stages : 0;
    c->nnodes = f32 ? 0 : cfg->num_biquad_nodes;
    c->time_segments = cfg->time_segments; c->warmup_cfg = cfg->biquad_warmup;
    c->pole_radius = c->nstages ? max_pole_radius(cfg->biquad_coeffs, (int)c->nstages) : 0.0;
    c->cur = 0; c->phase = 0; c->timing = false; c->timed_ms = 0; c->timed_launches = 0;
    c->gen = 1; c->dh_cache.resize(c->channels); c->dh_gen.assign(c->channels, 0);
    c->block_off = getenv("MSDR_NO_BLOCK") != nullptr;
    c->no_fuse = getenv("MSDR_Q15_NO_FUSE") != nullptr;
    c->blk_force = getenv("MSDR_BIQUAD_BLK") ? atoi(getenv("MSDR_BIQUAD_BLK")) : -1;
    c->flags = cfg->flags; c->mfw_nw = 0; c->mfw_waves_per_cu = 0; c->d_bq_state_alt = nullptr; c->d_mw_iir = nullptr; c->d_units = nullptr; c->units_cap = 0;
    c->mode_gen = 1; c->units_mode_gen = 0; c->units_nseg = 0; c->units_wgs = 0; c->units_wgs_ssb = 0; c->mfw_ssb_fold = false; c->mfw_am_fold = false; c->units_tiles = -1;
    c->part_nseg[0] = c->part_nseg[1] = 1; c->part_seg_len[0] = c->part_seg_len[1] = 0;
    c->d_at_tab = nullptr; c->at_ns = 0; c->at_stride = 0; c->at_nw = 0;
    c->d_qm_tab = nullptr; c->d_qm_order = nullptr; c->qm_stride = 0; c->qm_halo = 0; c->qm_bsteps = 0; c->qm_order_gen = 0; c->qm_fr = false;
    c->pll = nullptr; c->d_pll_q = nullptr; c->pll_q_cap = 0;
    c->anr = nullptr; c->d_anr_on = nullptr; c->anr_all = 0;
    c->f32_pll = f32 && (cfg->flags & MSDR_CHAIN_SYNCAM_PLL); c->h_bq_stages = 0; c->anr_gen = 1; c->post_mode_gen = 0; c->post_anr_gen = 0;
    c->aux = nullptr; c->post_bq = nullptr; c->npost = 0; c->nvirt = 0;
    c->d_post_ch = c->d_post_src = c->d_post_pll = c->d_post_anr = c->d_virt_row = nullptr;
    c->d_post_pll_state = c->d_post_anr_state = nullptr; c->d_aux_x = nullptr; c->d_aux_y = c->d_post_scratch = nullptr; c->post_cap_virt = 0; c->post_cap_post = 0;
    if (f32 && cfg->num_biquad_stages) { c->h_bq.assign(cfg->biquad_coeffs, cfg->biquad_coeffs + 5 * cfg->num_biquad_stages); c->h_bq_stages = cfg->num_biquad_stages; }
    if (f32) {
        c->h_coef_i.resize(c->tapsets); c->h_coef_q.resize(c->tapsets);
        for (uint32_t s = 0; s < c->tapsets; s++) {
            c->h_coef_i[s].assign((const float *)cfg->coeffs_i[s], (const float *)cfg->coeffs_i[s] + c->ntaps);
            c->h_coef_q[s].assign((const float *)cfg->coeffs_q[s], (const float *)cfg->coeffs_q[s] + c->ntaps);
        }
        c->h_osc.resize((size_t)c->osc_len * 2);
        for (uint32_t k = 0; k < c->osc_len; k++) {
            if (cfg->mixer == MSDR_MIXER_NCO) { c->h_osc[2 * k] = ((const float *)cfg->osc_q)[k]; c->h_osc[2 * k + 1] = ((const float *)cfg->osc_i)[k]; }
            else { const double c4[4] = {1, 0, -1, 0}, s4[4] = {0, 1, 0, -1}; c->h_osc[2 * k] = c4[k]; c->h_osc[2 * k + 1] = s4[k]; }
        }
        c->h_cnum.assign(1, 1.0);
        for (uint32_t st = 0; st < c->nstages; st++) {
            std::vector<double> nx(c->h_cnum.size() + 2, 0.0);
            for (size_t i = 0; i < c->h_cnum.size(); i++)
                for (int k = 0; k < 3; k++) nx[i + k] += c->h_cnum[i] * (double)cfg->biquad_coeffs[5 * st + k];
            c->h_cnum.swap(nx);
        }
    }
    memset(&c->info, 0, sizeof c->info);

    int rc = 0;
    // taps: pairs {hI[k], hQ[k]}, zero-padded in FRONT to a multiple of 4
    const uint32_t np = c->ntaps_pad, off = np - c->ntaps;
    if (f32) {
        std::vector<float> t((size_t)c->tapsets * np * 2, 0.0f);
        for (uint32_t s = 0; s < c->tapsets; s++)
            for (uint32_t k = 0; k < c->ntaps; k++) {
                t[((size_t)s * np + off + k) * 2 + 0] = ((const float *)cfg->coeffs_i[s])[k];
                t[((size_t)s * np + off + k) * 2 + 1] = ((const float *)cfg->coeffs_q[s])[k];
            }
        rc = upload(ctx, t, (float **)&c->d_taps);
    } else {
        std::vector<int32_t> t((size_t)c->tapsets * np * 2, 0);
        for (uint32_t s = 0; s < c->tapsets; s++)
            for (uint32_t k = 0; k < c->ntaps; k++) {
                t[((size_t)s * np + off + k) * 2 + 0] = ((const int16_t *)cfg->coeffs_i[s])[k];
                t[((size_t)s * np + off + k) * 2 + 1] = ((const int16_t *)cfg->coeffs_q[s])[k];
            }
        rc = upload(ctx, t, (int32_t **)&c->d_taps);
    }
    // osc pairs {osc_q ("cos"), osc_i ("sin")}
    if (!rc) {
        if (f32) {
            std::vector<float> o((size_t)c->osc_len * 2, 0.0f);
            if (cfg->mixer == MSDR_MIXER_NCO)
                for (uint32_t k = 0; k < c->osc_len; k++) { o[2 * k] = ((const float *)cfg->osc_q)[k]; o[2 * k + 1] = ((const float *)cfg->osc_i)[k]; }
            else { const float c4[4] = {1, 0, -1, 0}, s4[4] = {0, 1, 0, -1}; for (int k = 0; k < 4; k++) { o[2 * k] = c4[k]; o[2 * k + 1] = s4[k]; } }
            rc = upload(ctx, o, (float **)&c->d_osc);
        } else {
            std::vector<int32_t> o((size_t)c->osc_len * 2, 0);
            if (cfg->mixer == MSDR_MIXER_NCO)
                for (uint32_t k = 0; k < c->osc_len; k++) { o[2 * k] = ((const int16_t *)cfg->osc_q)[k]; o[2 * k + 1] = ((const int16_t *)cfg->osc_i)[k]; }
            rc = upload(ctx, o, (int32_t **)&c->d_osc);
        }
    }
    if (!rc) {
        std::vector<int> m(c->channels), ts(c->channels);
        for (uint32_t ch = 0; ch < c->channels && !rc; ch++) {
            m[ch] = cfg->mode ? cfg->mode[ch] : cfg->default_mode;
            ts[ch] = cfg->tapset ? cfg->tapset[ch] : 0;
            if (!mode_ok(m[ch]) || ts[ch] < 0 || (uint32_t)ts[ch] >= c->tapsets)
                rc = fail(MSDR_STATUS_ARGUMENT_ERROR, "channel %u: bad mode/tapset", ch);
        }
        if (!rc) rc = upload(ctx, m, &c->d_mode);
        if (!rc) rc = upload(ctx, ts, &c->d_tapset);
        c->h_mode = m; c->h_tapset = ts;
    }
    if (!rc) rc = dzalloc(ctx, (size_t)c->channels * c->hist_len, &c->d_hist[0]);
    if (!rc) rc = dzalloc(ctx, (size_t)c->channels * c->hist_len, &c->d_hist[1]);
    // ---- Q15 on the integer matrix cores (msdr_chain_q15mf.hiph): byte-split Toeplitz fragments per (tap set, phase mod 4) ----
    // mixer: the Fs/4 loop, or a freq_conv table of period 4 that is zero in osc_q at one parity of phases and zero in osc_i at the
    // other (the reference node driven at fs/4): then, too, every sample feeds only the I or only the Q filter
    int qm_par[2] = {0, 1};                                   // phase parity that feeds the I / the Q accumulator
    bool qm_mixer_ok = (cfg->mixer == MSDR_MIXER_FS4);
    if (!f32 && cfg->mixer == MSDR_MIXER_NCO && cfg->osc_len % 4 == 0) {
        const int16_t *oq = (const int16_t *)cfg->osc_q, *oi = (const int16_t *)cfg->osc_i;
        bool ok = true;
        for (uint32_t k = 4; k < cfg->osc_len && ok; k++) ok = (oq[k] == oq[k - 4]) && (oi[k] == oi[k - 4]);
        int pq = -1, pi = -1;                                  // parity of the phases where osc_q / osc_i is not zero
        for (int k = 0; k < 4 && ok; k++) {
            if (oq[k] && oi[k]) ok = false;
            if (oq[k]) { if (pq >= 0 && pq != (k & 1)) ok = false; pq = k & 1; }
            if (oi[k]) { if (pi >= 0 && pi != (k & 1)) ok = false; pi = k & 1; }
        }
        if (ok && pq >= 0 && pi >= 0 && pq == pi) ok = false;
        if (ok) { if (pq < 0) pq = (pi >= 0) ? 1 - pi : 0; if (pi < 0) pi = 1 - pq; qm_par[0] = pq; qm_par[1] = pi; qm_mixer_ok = true; }
    }
    // any other table of the node still repeats with the 128-sample block: the full-rate layout (the products are made at staging time)
    if (!f32 && cfg->mixer == MSDR_MIXER_NCO && !qm_mixer_ok && cfg->osc_len > 0 && (128 % cfg->osc_len) == 0) { qm_mixer_ok = true; c->qm_fr = true; }
    if (!rc && !f32 && qm_mixer_ok && !(cfg->flags & MSDR_CHAIN_NO_MFMA) && qm_halo((int)c->ntaps) <= 512) {
        std::vector<const int16_t *> ci(c->tapsets), cq(c->tapsets);
        for (uint32_t s = 0; s < c->tapsets; s++) { ci[s] = (const int16_t *)cfg->coeffs_i[s]; cq[s] = (const int16_t *)cfg->coeffs_q[s]; }
        QmTables T;
        qm_build_tables((int)c->ntaps, c->tapsets, ci.data(), cq.data(), qm_par, false, T, c->qm_fr);
        c->qm_set_ok = T.set_ok;
        if (!T.blob.empty()) {
            rc = upload(ctx, T.blob, &c->d_qm_tab);
            c->qm_stride = T.stride; c->qm_halo = T.halo; c->qm_bsteps = T.bsteps;
        }
    }
    if (!rc && f32) {
        std::vector<BiquadCascadeTables<kChainR>> tabs(1);
        make_cascade_tables<kChainR>(cfg->biquad_coeffs, (int)c->nstages, &tabs[0]);
        rc = upload(ctx, tabs, &c->d_bq);
        if (!rc) rc = dzalloc(ctx, (size_t)c->channels * kBqStateFloats, &c->d_bq_state);
    }
    // ---- tap folding (F32): oscillator period, folded tables, per-channel folded-set index ----------
    c->fold_P = 0; c->fold_fs4_exact = false; c->mf_P = 0; c->mf_fr = false;
    if (!rc && f32 && !(cfg->flags & MSDR_CHAIN_NO_TAP_FOLDING)) {
        std::vector<double> oc, os;                       // one period of cos / sin
        if (cfg->mixer == MSDR_MIXER_FS4) { oc = {1, 0, -1, 0}; os = {0, 1, 0, -1}; }
        else {
            const float *fq = (const float *)cfg->osc_q, *fi = (const float *)cfg->osc_i;
            // periods 1, 2, 4: the VALU kernel's per-phase tap tables too; 8, 16, 32 (divisors of the 32-sample output row, so that every
            // row of a tile sees the same oscillator phases): the matrix-core tables only -- one (H + 32) x 32 block per starting phase
            for (uint32_t P : {1u, 2u, 4u, 8u, 16u, 32u}) {
                if (cfg->osc_len % P) continue;
                // a table that only "repeats" after its own length is folded on request only: the folded evaluation differs from the
                // as-written one by ~2e-7 of the STRONGER sideband, which shows as > 1e-5 of the output where the wanted sideband is
                // 40+ dB down (a pure tone on the suppressed side); the as-written kernel tracks the CMSIS order there
                if (P > 4 && P == cfg->osc_len && !(cfg->flags & MSDR_CHAIN_FOLD_ANY_PERIOD)) continue;
                bool periodic = true;
                for (uint32_t k = P; k < cfg->osc_len && periodic; k++) periodic = (fq[k] == fq[k - P]) && (fi[k] == fi[k - P]);
                if (periodic) { oc.assign(fq, fq + P); os.assign(fi, fi + P); break; }
            }
        }
        c->mf_P = (int)oc.size();
        // no short period: every AudioEffectFreqConv table still repeats with the block (128 entries, freq_conv.cpp:67-103) -- the
        // matrix-core kernel then stages the two mixer products as full-rate streams and runs both FIRs over every sample
        // (up to 247 taps: beyond that the window does not fit next to the fragments; chain_kernel<ArithF32> takes those)
        c->mf_fr = oc.empty() && cfg->mixer == MSDR_MIXER_NCO && cfg->osc_len > 0 && (128 % cfg->osc_len) == 0 &&
                   c->ntaps < (getenv("MSDR_FR_MAX_TAPS") ? (uint32_t)atoi(getenv("MSDR_FR_MAX_TAPS")) : 100000u);      // (round 5: whether the fragments fit LDS beside a wave's windows is decided where the tables are built; the variable re-creates round 4's limit of 248 for A/B runs)
        if (c->mf_fr) c->mf_P = 1;                             // one table per (tap set, flavour): the oscillator is not in it
        if (oc.size() > 4) oc.clear();                         // no VALU fold tables beyond period 4
        if (!oc.empty()) {
            const int P = (int)oc.size();
            c->fold_P = P;
            c->fold_fs4_exact = (P == 4 && oc[0] == 1 && oc[1] == 0 && oc[2] == -1 && oc[3] == 0 &&
                                 os[0] == 0 && os[1] == 1 && os[2] == 0 && os[3] == -1);
            // T_phi[k] = in_scale * (hI[k] c[psi] -/+ hQ[k] s[psi]),  psi = (phi - (Np-1) + k) mod P  (k = padded tap index).
            // Stored once per starting phase `rot` as [set*3+v][rot][step][row j < PE][pair-column i < 2][2]:
            // row j serves outputs r = j (mod PE) whose phase is phi = (rot + j) mod P; pair-column m = 2*step + i holds
            // (T[2m-1], T[2m]) for even rows and (T[2m], T[2m+1]) for odd rows (T = 0 outside [0, Np)).
            const int PE = P < 2 ? 2 : P;
            const int steps = fold_steps((int)np);
            std::vector<float> ft((size_t)c->tapsets * 3 * P * steps * 4 * PE, 0.0f);
            std::vector<double> T(np + 2);
            for (uint32_t s = 0; s < c->tapsets; s++)
                for (int v = 0; v < 3; v++)
                    for (int rot = 0; rot < P; rot++)
                        for (int j = 0; j < PE; j++) {
                            const int phi = (rot + j) % P;
                            std::fill(T.begin(), T.end(), 0.0);
                            for (uint32_t k = 0; k < c->ntaps; k++) {
                                const uint32_t kp = off + k;
                                const int psi = (int)((((long long)phi - (long long)(np - 1) + kp) % P + P) % P);
                                const double hi = ((const float *)cfg->coeffs_i[s])[k], hq = ((const float *)cfg->coeffs_q[s])[k];
                                T[kp] = (hi * oc[psi] + (v == 0 ? -1.0 : 1.0) * hq * os[psi]) * (double)c->in_scale;
                            }
                            auto tap = [&](long long k) { return (k >= 0 && k < (long long)np) ? (float)T[k] : 0.0f; };
                            float *base = ft.data() + (((size_t)s * 3 + v) * P + rot) * ((size_t)steps * 4 * PE);
                            for (int st = 0; st < steps; st++)
                                for (int i = 0; i < 2; i++) {
                                    const long long m = 2LL * st + i;
                                    const long long k0 = (j & 1) ? 2 * m : 2 * m - 1;
                                    base[((size_t)st * PE + j) * 4 + 2 * i + 0] = tap(k0);
                                    base[((size_t)st * PE + j) * 4 + 2 * i + 1] = tap(k0 + 1);
                                }
                        }
            rc = upload(ctx, ft, &c->d_ftaps);
            if (!rc) {
                std::vector<BiquadCascadeTables<kFoldR>> tabs(1);
                make_cascade_tables<kFoldR>(cfg->biquad_coeffs, (int)c->nstages, &tabs[0]);
                rc = upload(ctx, tabs, &c->d_bq_fold);
            }
        }
    }
    // ---- matrix-core tables (F32, short-period oscillator): B fragments per (tap set x {LSB, USB, envelope}, rotation) ----
    c->mf_ok = false;
    if (!rc && f32 && c->mf_P > 0 && !(cfg->flags & MSDR_CHAIN_NO_MFMA) && mf_halo((int)(c->ntaps + 2 * c->nstages)) <= 2048) {
        // SSB tables carry the cascade's numerator C(z) = prod (b0 + b1 z^-1 + b2 z^-2): the FIR grows by 2 taps per section
        const bool fr = c->mf_fr;
        // (full rate: the window is handled as if it were interleaved I0 Q0 I1 Q1 ..: "even" positions = the I stream, "odd" = the Q
        //  stream, twice as many of them, so that the run search and the fragment emission below serve both layouts)
        const int P = c->mf_P, N = (int)c->ntaps, NF = N + 2 * (int)c->nstages, H = mf_halo(NF), KI = H + 32, KIv = fr ? 2 * KI : KI, J = KIv / 32;
        std::vector<double> cnum(1, 1.0);
        for (uint32_t st = 0; st < c->nstages; st++) {
            std::vector<double> nx(cnum.size() + 2, 0.0);
            for (size_t i = 0; i < cnum.size(); i++)
                for (int k = 0; k < 3; k++) nx[i + k] += cnum[i] * (double)cfg->biquad_coeffs[5 * st + k];
            cnum.swap(nx);
        }
        std::vector<double> oc(P), os(P);
        if (cfg->mixer == MSDR_MIXER_FS4) { const double c4[4] = {1, 0, -1, 0}, s4[4] = {0, 1, 0, -1}; for (int k = 0; k < 4; k++) { oc[k] = c4[k]; os[k] = s4[k]; } }
        else if (fr) { oc[0] = 1.0; os[0] = 1.0; }
        else for (int k = 0; k < P; k++) { oc[k] = ((const float *)cfg->osc_q)[k]; os[k] = ((const float *)cfg->osc_i)[k]; }
        struct Tab { MfmaTableHeader h; std::vector<_Float16> frags; };
        std::vector<Tab> tabs((size_t)c->tapsets * 3 * P);
        // full rate, long filters: the k-step fragments of a Toeplitz block are 47 taps spread over 1024 entries -- 512 taps x two filters are
        // 136 KB of LDS and leave room for two waves' windows.  COMPACT layout: the taps themselves (reversed, scaled, hi | lo pieces) in eight
        // copies shifted by one entry each, so that the 8 consecutive entries a lane needs for ANY k-step are one aligned 16-byte read
        // (msdr_chain_mfw.hiph, mw_compact_stride): 2 x 19 KB at 512 taps, 6 - 7 waves per CU instead of 2.  What it cannot hold is a block
        // that is not Toeplitz: the SSB tables then keep the cascade out of their columns (it runs as the lane scan, like 3 - 4 sections).
        // From 128 taps on (profiles/r05/nco_long_taps.txt); MSDR_FR_COMPACT=0 / 1 overrides (tests, A/B runs).
        bool compact = fr && N >= 128;
        if (const char *e = getenv("MSDR_FR_COMPACT")) compact = fr && atoi(e) != 0;
        const int CS = compact ? mw_compact_stride(H) : 0;
        int bsteps = 0;
        const bool ok = true;
        std::vector<double> M[2];
        M[0].resize((size_t)KIv * 32); M[1].resize((size_t)KIv * 32);
        std::vector<double> di(NF), dq(NF), fi(NF), fq(NF);
        // ---- folded IIR (wave-stream kernel, SSB tables, 1 or 2 sections): the all-pole cascade's zero-state response inside a
        // 32-sample row is a lower-triangular Toeplitz matrix L (impulse response g); B' = B L^T puts it into the FIR's own matrix
        // product.  What is left per row is the response R sigma to the state sigma at the row's start (cascade basis:
        // (w_s[-1], w_s[-2]) per section) and the row-to-row recurrence sigma' = z + M sigma; see MwIirConsts.
        const int S_ = (int)c->nstages, NS = 2 * S_;
        bool iirfold = (S_ == 1 || S_ == 2);
        bool amfold = false;
        double gap[32] = {0}, sec_a1[2] = {0, 0}, sec_a2[2] = {0, 0}, gl1[2] = {0, 0}, Rmax = 0.0;
        std::vector<float> iirc;
        auto run_cascade = [&](double *sig, const double *vin, double *yout, int len) {      // all-pole sections in series, state in/out
            for (int t = 0; t < len; t++) {
                double u = vin ? vin[t] : 0.0;
                for (int q = 0; q < S_; q++) {
                    const double w = u + sec_a1[q] * sig[2 * q] + sec_a2[q] * sig[2 * q + 1];
                    sig[2 * q + 1] = sig[2 * q]; sig[2 * q] = w; u = w;
                }
                if (yout) yout[t] = u;
            }
        };
        if (iirfold) {
            for (int q = 0; q < S_; q++) { sec_a1[q] = cfg->biquad_coeffs[5 * q + 3]; sec_a2[q] = cfg->biquad_coeffs[5 * q + 4]; }
            {   // impulse response (32 samples for L) and L1 norms of the partial cascades (state bounds)
                std::vector<double> imp(8192, 0.0), y(8192);
                imp[0] = 1.0;
                double sg[4] = {0, 0, 0, 0};
                run_cascade(sg, imp.data(), y.data(), 8192);
                for (int n = 0; n < 32; n++) gap[n] = y[n];
                if (std::fabs(y[8191]) + std::fabs(y[8190]) > 1e-12) iirfold = false;          // does not decay: no bound, keep the VALU scan
                for (int q = 0; q < S_ && iirfold; q++) {
                    double s2[4] = {0, 0, 0, 0}, acc = 0.0;
                    const int keep = S_;
                    (void)keep;
                    // partial cascade 0..q: rerun with only q+1 sections
                    for (int t = 0; t < 8192; t++) {
                        double u = imp[t];
                        for (int r = 0; r <= q; r++) { const double w = u + sec_a1[r] * s2[2 * r] + sec_a2[r] * s2[2 * r + 1]; s2[2 * r + 1] = s2[2 * r]; s2[2 * r] = w; u = w; }
                        acc += std::fabs(u);
                    }
                    gl1[q] = acc;
                }
            }
        }
        if (iirfold) {
            iirc.assign(kMwIirFloats, 0.0f);
            double Mt[4][4] = {{0}}, Rr[32][4] = {{0}};
            for (int j = 0; j < NS; j++) {                          // unit state e_j, zero input, one row
                double sg[4] = {0, 0, 0, 0}, y[32];
                sg[j] = 1.0;
                run_cascade(sg, nullptr, y, 32);
                for (int m = 0; m < 32; m++) { Rr[m][j] = y[m]; Rmax = std::max(Rmax, std::fabs(y[m])); }
                for (int i = 0; i < NS; i++) Mt[i][j] = sg[i];
            }
            auto matmul = [&](const double A[4][4], const double B[4][4], double C[4][4]) {
                double t[4][4];
                for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { t[i][j] = 0; for (int k = 0; k < 4; k++) t[i][j] += A[i][k] * B[k][j]; }
                memcpy(C, t, sizeof t);
            };
            double Pw[4][4];
            memcpy(Pw, Mt, sizeof Pw);
            for (int k = 0; k < 4; k++) {                           // M^(2^k), stored by columns
                for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) iirc[kMwIirPow + k * 16 + j * 4 + i] = (float)Pw[i][j];
                matmul(Pw, Pw, Pw);
            }
            memcpy(Pw, Mt, sizeof Pw);
            for (int l = 0; l < 32; l++) {                          // M^(l+1): lane-major [lane][pair] float2, pairs = the column halves the scan uses
                const int pidx2[6] = {0, 2, 4, 6, 10, 14}, pidx1[2] = {0, 4};      // first float of each pair in the column-major 4x4
                const int npairs = (S_ == 2) ? 6 : 2;
                for (int q = 0; q < npairs; q++) {
                    const int f = (S_ == 2) ? pidx2[q] : pidx1[q];
                    for (int e = 0; e < 2; e++) iirc[kMwIirLane + (l * npairs + q) * 2 + e] = (float)Pw[(f + e) & 3][(f + e) >> 2];
                }
                matmul(Pw, Mt, Pw);
            }
            for (int m = 0; m < 32; m++) for (int k = 0; k < NS; k++) iirc[kMwIirGfix + m * 4 + k] = (m - k >= 0) ? (float)gap[m - k] : 0.0f;
            if (Rmax * std::ldexp(1.0, kMwIirSigExp) > 60000.0) iirfold = false;
            // response fragments for the correction MFMA: A operand, lane (m, kg), element j: kg = 1 and j < NS: R[m][j] 2^e, else 0
            _Float16 *rh = reinterpret_cast<_Float16 *>(iirc.data() + kMwIirRfrag), *rl = rh + 512;
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const double val = ((l >> 5) == 1 && j < NS) ? Rr[l & 31][j] * std::ldexp(1.0, kMwIirSigExp) : 0.0;
                    const _Float16 vh = (_Float16)val;
                    rh[l * 8 + j] = vh; rl[l * 8 + j] = (_Float16)(val - (double)vh);
                }
            iirc[kMwIirCoef + 0] = (S_ == 2) ? (float)sec_a1[1] : 0.0f;       // the last section's feedback: w_(S-1)[n] = y[n] - a1 y[n-1] - a2 y[n-2]
            iirc[kMwIirCoef + 1] = (S_ == 2) ? (float)sec_a2[1] : 0.0f;
            // ---- envelope modes: the cascade cannot go into the FIR (the envelope is in between), so its zero-state response runs as
            // its own matrix product  Y0 = L D + Rd delta:  L = full cascade (numerator * all-pole) impulse response, lower-triangular
            // Toeplitz; delta = the previous row's last 2S inputs (numerator history), Rd their effect on this row
            double hfull[32], Lmax = 0.0;
            for (int m = 0; m < 32; m++) {
                double a = 0.0;
                for (size_t t = 0; t < cnum.size() && (int)t <= m; t++) a += cnum[t] * gap[m - (int)t];
                hfull[m] = a; Lmax = std::max(Lmax, std::fabs(a));
            }
            double Rd[32][4] = {{0}};
            for (int m = 0; m < 32; m++)
                for (int k = 1; k <= 2 * S_; k++) {                 // input d[-k]
                    double a = 0.0;
                    for (int t = 0; t + k < (int)cnum.size() && t <= m; t++) a += gap[m - t] * cnum[t + k];
                    Rd[m][4 - k] = a; Lmax = std::max(Lmax, std::fabs(a));          // element j = 4 - k: the row's inputs 28..31 in order
                }
            amfold = iirfold && Lmax * std::ldexp(1.0, kMwIirEnvExp) < 60000.0;
            _Float16 *lf = reinterpret_cast<_Float16 *>(iirc.data() + kMwIirLfrag), *df = reinterpret_cast<_Float16 *>(iirc.data() + kMwIirDfrag);
            for (int st2 = 0; st2 < 2; st2++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 8; j++) {
                        const int m = l & 31, kk = (j & 3) + 8 * (2 * st2 + (j >> 2)) + 4 * (l >> 5);
                        const double val = (m >= kk) ? hfull[m - kk] * std::ldexp(1.0, kMwIirEnvExp) : 0.0;
                        const _Float16 vh = (_Float16)val;
                        lf[st2 * 1024 + l * 8 + j] = vh; lf[st2 * 1024 + 512 + l * 8 + j] = (_Float16)(val - (double)vh);
                    }
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const double val = ((l >> 5) == 1 && j < 4) ? Rd[l & 31][j] * std::ldexp(1.0, kMwIirEnvExp) : 0.0;
                    const _Float16 vh = (_Float16)val;
                    df[l * 8 + j] = vh; df[512 + l * 8 + j] = (_Float16)(val - (double)vh);
                }
        }
        // A retune (msdr_chain_set_mode) hands the cascade's state and numerator history of the OLD mode / tap set to the kernel of
        // the NEW one, which takes them in through fp16 at its own table's scale: the scale of every table must therefore hold
        // what ANY table of this chain can leave behind, not only its own outputs (a tap set with 20x the gain of its neighbour
        // used to saturate the first rows after the switch).  Bounds in units of in_scale, like smax / emax below.
        double chain_d_bound = 0.0, chain_sig_bound = 0.0, oamp = 0.0;
        for (int k = 0; k < P; k++) oamp = std::max(oamp, std::max(std::fabs(oc[k]), std::fabs(os[k])));
        if (fr) for (uint32_t k = 0; k < cfg->osc_len; k++) oamp = std::max(oamp, (double)std::max(std::fabs(((const float *)cfg->osc_q)[k]), std::fabs(((const float *)cfg->osc_i)[k])));
        if (iirfold) {
            double cl1 = 0.0, glmax = 0.0;
            for (double cc : cnum) cl1 += std::fabs(cc);
            for (int q = 0; q < S_; q++) glmax = std::max(glmax, gl1[q]);
            for (uint32_t s = 0; s < c->tapsets; s++) {
                const float *hi = (const float *)cfg->coeffs_i[s], *hq = (const float *)cfg->coeffs_q[s];
                std::vector<double> gi(NF, 0.0), gq(NF, 0.0);
                double a = 0.0, b = 0.0, f1 = 0.0;
                for (int dl = 0; dl < N; dl++) {
                    a += std::fabs((double)hi[dl]); b += std::fabs((double)hq[dl]);
                    for (size_t i = 0; i < cnum.size(); i++) { gi[dl + i] += cnum[i] * hi[N - 1 - dl]; gq[dl + i] += cnum[i] * hq[N - 1 - dl]; }
                }
                for (int dl = 0; dl < NF; dl++) f1 += std::fabs(gi[dl]) + std::fabs(gq[dl]);
                const double xs = 32768.0 * oamp * (a + b);                       // |SSB demodulator output| (>= the envelope)
                const double env = 32768.0 * oamp * std::sqrt(a * a + b * b);     // |envelope|
                chain_d_bound = std::max(chain_d_bound, xs);
                chain_sig_bound = std::max(chain_sig_bound, std::max(32768.0 * oamp * f1, env * cl1) * glmax);
            }
            c->own_d_bound = chain_d_bound; c->own_sig_bound = chain_sig_bound;
            chain_d_bound = std::max(chain_d_bound, g_chain_floor_d); chain_sig_bound = std::max(chain_sig_bound, g_chain_floor_sig);
        }
        for (uint32_t s = 0; s < c->tapsets && ok; s++) {
            const float *hi = (const float *)cfg->coeffs_i[s], *hq = (const float *)cfg->coeffs_q[s];
            // taps by delay (arm_fir keeps its coefficients time-reversed: index N - 1 - delay); plain and numerator-folded
            std::fill(di.begin(), di.end(), 0.0); std::fill(dq.begin(), dq.end(), 0.0);
            std::fill(fi.begin(), fi.end(), 0.0); std::fill(fq.begin(), fq.end(), 0.0);
            for (int dl = 0; dl < N; dl++) {
                di[dl] = hi[N - 1 - dl]; dq[dl] = hq[N - 1 - dl];
                for (size_t i = 0; i < cnum.size(); i++) { fi[dl + i] += cnum[i] * di[dl]; fq[dl + i] += cnum[i] * dq[dl]; }
            }
            for (int v = 0; v < 3 && ok; v++)
                for (int rot = 0; rot < P && ok; rot++) {
                    // B[i][b]: window sample i (oscillator phase (rot + i) mod P) meets output column b at delay H + b - i
                    const bool numfold = (v != 2) && c->nstages > 0;
                    const std::vector<double> &ti = numfold ? fi : di, &tq = numfold ? fq : dq;
                    const int nt = numfold ? NF : N;
                    double maxabs = 0.0;
                    if (fr) {
                        // the samples arrive mixed: stream I meets hI, stream Q meets -/+ hQ (SSB: one accumulator) or hQ (envelope: its own)
                        std::fill(M[0].begin(), M[0].end(), 0.0); std::fill(M[1].begin(), M[1].end(), 0.0);
                        for (int i = 0; i < KI; i++)
                            for (int b = 0; b < 32; b++) {
                                const int delay = H + b - i;
                                if (delay < 0 || delay >= nt) continue;
                                if (v == 2) { M[0][(size_t)(2 * i) * 32 + b] = di[delay]; M[1][(size_t)(2 * i + 1) * 32 + b] = dq[delay]; }
                                else { M[0][(size_t)(2 * i) * 32 + b] = ti[delay]; M[0][(size_t)(2 * i + 1) * 32 + b] = (v == 0 ? -1.0 : 1.0) * tq[delay]; }
                                maxabs = std::max(maxabs, std::max(std::fabs(v == 2 ? di[delay] : ti[delay]), std::fabs(v == 2 ? dq[delay] : tq[delay])));
                            }
                    } else
                    for (int i = 0; i < KI; i++)
                        for (int b = 0; b < 32; b++) {
                            const int delay = H + b - i;
                            double m0 = 0.0, m1 = 0.0;
                            if (delay >= 0 && delay < nt) {
                                const int psi = (((rot + i - H) % P) + P) % P;       // window sample i is stream sample t0 - H + i, t0 = 0 (mod P)
                                if (v == 2) { m0 = di[delay] * oc[psi]; m1 = dq[delay] * os[psi]; }
                                else m0 = ti[delay] * oc[psi] + (v == 0 ? -1.0 : 1.0) * tq[delay] * os[psi];
                            }
                            M[0][(size_t)i * 32 + b] = m0; M[1][(size_t)i * 32 + b] = m1;
                            maxabs = std::max(maxabs, std::max(std::fabs(m0), std::fabs(m1)));
                        }
                    const bool fold_iir = numfold && iirfold && !compact;
                    if (fold_iir) {                                      // B' = B L^T: columns take in the zero-state all-pole response
                        maxabs = 0.0;
                        for (int i = 0; i < KIv; i++) {
                            double row[32];
                            for (int b2 = 0; b2 < 32; b2++) {
                                double a = 0.0;
                                for (int b = 0; b <= b2; b++) a += M[0][(size_t)i * 32 + b] * gap[b2 - b];
                                row[b2] = a;
                            }
                            for (int b2 = 0; b2 < 32; b2++) { M[0][(size_t)i * 32 + b2] = row[b2]; maxabs = std::max(maxabs, std::fabs(row[b2])); }
                        }
                    }
                    int ex = 0;
                    if (maxabs > 0) { std::frexp(maxabs, &ex); ex = (fold_iir ? 10 : 14) - ex; }           // maxabs * 2^ex in [2^13, 2^14) ([2^9, 2^10) with the folded IIR)
                    if (fold_iir) {
                        // the state sigma (in accumulator units = true / post) goes through fp16 as sigma 2^-kMwIirSigExp: keep it <= 2^15
                        double tl1 = 0.0;
                        for (int dl = 0; dl < nt; dl++) tl1 += std::fabs(ti[dl]) + std::fabs(tq[dl]);
                        double smax = 0.0;
                        for (int q = 0; q < S_; q++) smax = std::max(smax, 32768.0 * oamp * tl1 * gl1[q]);   // |w_q| in units of in_scale
                        smax = std::max(smax, chain_sig_bound);
                        while (ex > -8 && smax * std::ldexp(1.0, ex - kMwIirSigExp) > 30000.0) ex--;
                    }
                    const bool fold_am = (v == 2) && amfold;
                    if (fold_am) {
                        // the envelope (accumulator units) goes through fp16 as e 2^-kMwIirEnvExp, the cascade's states as sigma 2^-kMwIirSigExp
                        double colmax = 0.0;
                        for (int b = 0; b < 32; b++) {
                            double c0 = 0.0, c1 = 0.0;
                            for (int i = 0; i < KIv; i++) { c0 += std::fabs(M[0][(size_t)i * 32 + b]); c1 += std::fabs(M[1][(size_t)i * 32 + b]); }
                            colmax = std::max(colmax, std::sqrt(c0 * c0 + c1 * c1));
                        }
                        if (fr) colmax *= oamp;                                              // the streams carry the oscillator's amplitude
                        const double emax = std::max(32768.0 * colmax, chain_d_bound);      // |envelope| / 2^ex, and whatever another mode's numerator history holds
                        double cl1 = 0.0, smax = 0.0;
                        for (double cc : cnum) cl1 += std::fabs(cc);
                        for (int q = 0; q < S_; q++) smax = std::max(smax, 32768.0 * colmax * cl1 * gl1[q]);
                        smax = std::max(smax, chain_sig_bound);
                        while (ex > -8 && (emax * std::ldexp(1.0, ex - kMwIirEnvExp) > 30000.0 || smax * std::ldexp(1.0, ex - kMwIirSigExp) > 30000.0)) ex--;
                    }
                    const double scale = std::ldexp(1.0, ex);
                    Tab &T = tabs[((size_t)s * 3 + v) * P + rot];
                    memset(&T.h, 0, sizeof T.h);
                    T.h.am = (v == 2); T.h.numfold = numfold; T.h.post = (float)((double)c->in_scale / scale);
                    T.h.iirfold = fold_iir || fold_am; T.h.ipost = (float)(scale / (double)c->in_scale);
                    int ns = 0;
                    // full rate, envelope table, one low-pass for both streams: M[1] over the Q positions IS M[0] over the I positions -- one set of
                    // fragments serves both accumulators (half the LDS: 512 taps fit where two sets do not)
                    const bool share = fr && v == 2 && memcmp(hi, hq, (size_t)N * sizeof(float)) == 0;
                    for (int o = 0; o < (v == 2 ? 2 : 1); o++)
                        for (int src = 0; src < 2; src++) {
                            int jlo = J, jhi = -1;                                   // chunks with any non-zero entry: one contiguous run
                            for (int j = 0; j < J; j++) {
                                bool any = false;
                                for (int m = 0; m < 16 && !any; m++)
                                    for (int b = 0; b < 32 && !any; b++) any = M[o][(size_t)(2 * (16 * j + m) + src) * 32 + b] != 0.0;
                                if (any) { jlo = std::min(jlo, j); jhi = std::max(jhi, j); }
                            }
                            T.h.run[o][src].j0 = (jhi >= 0) ? jlo : 0;
                            T.h.run[o][src].cnt = (jhi >= 0) ? jhi - jlo + 1 : 0;
                            if (o == 1 && src == 0) T.h.acc1_base = share ? 0 : ns;      // (accumulator 1's fragments start behind accumulator 0's two runs)
                            T.h.bk[o][src] = (o == 1 ? T.h.acc1_base : 0) * 2048;      // where the kernel's step counter meets the run (bytes at step 0)
                            if (compact) {
                                // step st of the run = chunk j = j0 + st (after run a: j0b + st - cnta) sits 32 j bytes into the array
                                const int before = (src == 1) ? T.h.run[o][0].cnt : 0;
                                if (share && o == 1) { T.h.bk[o][src] = T.h.bk[0][0] - 32 * T.h.run[0][0].j0 + 32 * (T.h.run[o][src].j0 - before); continue; }
                                if (jhi < 0) continue;
                                const size_t base = T.frags.size();
                                T.h.bk[o][src] = (int)(base * 2) + 32 * (T.h.run[o][src].j0 - before);
                                T.frags.resize(base + (size_t)8 * CS, (_Float16)0.0f);         // 8 copies x (hi | lo) x CS / 2 entries
                                for (int sh = 0; sh < 8; sh++)
                                    for (int e = 0; e < CS / 2; e++) {
                                        // g[u]: the block's entry on the diagonal i - b + 31 = u (window sample i, output column b)
                                        const int u = e + sh, b = u < 31 ? 31 - u : 0, i = u - 31 + b;
                                        if (i >= KI) continue;
                                        double val = M[o][(size_t)(2 * i + src) * 32 + b] * scale;
#if defined(MSDR_MUTATE) && MSDR_MUTATE == 2
                                        if (val != 0.0) { int e_; const double f_ = std::frexp(val, &e_); val = std::ldexp(std::nearbyint(std::ldexp(f_, 16)), e_ - 16); }
#endif
                                        const _Float16 vh = (_Float16)val;
                                        T.frags[base + (size_t)sh * (CS / 2) + e] = vh;
                                        T.frags[base + (size_t)(8 + sh) * (CS / 2) + e] = (_Float16)(val - (double)vh);
#if defined(MSDR_MUTATE) && MSDR_MUTATE == 1
                                        T.frags[base + (size_t)(8 + sh) * (CS / 2) + e] = (_Float16)0.0f;
#endif
                                    }
                                continue;
                            }
                            if (share && o == 1) continue;                             // same chunks, same values as run[0][0]: stored once
                            for (int j = jlo; j <= jhi; j++, ns++) {
                                const size_t base = T.frags.size();
                                T.frags.resize(base + 1024);                          // hi piece [64 lanes][8], then lo piece
                                for (int l = 0; l < 64; l++)
                                    for (int jj = 0; jj < 8; jj++) {
                                        double val = M[o][(size_t)(2 * (16 * j + 8 * (l >> 5) + jj) + src) * 32 + (l & 31)] * scale;
#if defined(MSDR_MUTATE) && MSDR_MUTATE == 2       /* `make mutants`, never the product: the taps rounded to 16 significant bits */
                                        if (val != 0.0) { int e_; const double f_ = std::frexp(val, &e_); val = std::ldexp(std::nearbyint(std::ldexp(f_, 16)), e_ - 16); }
#endif
                                        const _Float16 vh = (_Float16)val;
                                        T.frags[base + l * 8 + jj] = vh;
                                        T.frags[base + 512 + l * 8 + jj] = (_Float16)(val - (double)vh);
#if defined(MSDR_MUTATE) && MSDR_MUTATE == 1       /* `make mutants`, never the product: the lo pieces of the tap fragments dropped (the xh x Bl product is gone) */
                                        T.frags[base + 512 + l * 8 + jj] = (_Float16)0.0f;
#endif
                                    }
                            }
                        }
                    if (compact) ns = (int)((T.frags.size() * 2 + 2047) / 2048);          // (LDS sizes are counted in 2 KB steps everywhere)
                    T.frags.resize((size_t)ns * 1024, (_Float16)0.0f);
                    T.h.cs = CS; T.h.bstep = compact ? 32 : 2048;
                    T.h.nsteps = ns;
                    bsteps = std::max(bsteps, ns);
                }
        }
        if (ok && bsteps > 0) {
            const int stride = kMfHdrBytes + bsteps * 2048;
            std::vector<char> blob((size_t)stride * tabs.size(), 0);
            for (size_t t = 0; t < tabs.size(); t++) {
                memcpy(blob.data() + t * stride, &tabs[t].h, sizeof(MfmaTableHeader));
                memcpy(blob.data() + t * stride + kMfHdrBytes, tabs[t].frags.data(), tabs[t].frags.size() * sizeof(_Float16));
            }
            rc = upload(ctx, blob, &c->d_mf_tab);
            if (!rc) {
                std::vector<BiquadCascadeTables<kMfL>> bt(1);
                make_cascade_tables<kMfL>(cfg->biquad_coeffs, (int)c->nstages, &bt[0]);
                rc = upload(ctx, bt, &c->d_bq_mf);
            }
            if (!rc) {
                std::vector<BiquadCascadeTables<32>> bt(1);
                make_cascade_tables<32>(cfg->biquad_coeffs, (int)c->nstages, &bt[0]);
                rc = upload(ctx, bt, &c->d_bq_mf32);
            }
            if (!rc) {
                c->mf_ok = true; c->mf_halo = H; c->mf_bsteps = bsteps; c->mf_stride = stride;
                // wave-stream variant: waves per workgroup that put the most waves on a CU (<= 16: the kernel's <= 128 VGPRs allow
                // 4 per SIMD) under the 160 KB of LDS -- counted in whole waves per SIMD.  A SIMD's tile rate is the same from two waves
                // on (profiles/r03/c3_trims.txt), so a workgroup is as slow as its fullest SIMD: 13 waves (4 + 3 + 3 + 3) ran 4 % behind
                // 12 per tile and CU.  Ties go to fewer waves, then to the smaller workgroup.
                int best = 0, best_eff = 0, wcap = 16;
#ifdef MSDR_STAMPS
                if (const char *e = getenv("MSDR_DBG_NW")) wcap = std::max(1, std::min(16, atoi(e)));    // stamps build: a lone wave per SIMD etc.
#endif
                for (int w = 1; w <= wcap; w++) {
                    const size_t l = mw_lds_bytes(H, bsteps, w, fr);
                    if (l > 160 * 1024) break;
                    const int wgs = std::min<int>((int)((160 * 1024) / l), wcap / w);
                    const int on_cu = wgs * w, eff = on_cu >= 4 ? 4 * (on_cu / 4) : on_cu;
                    if (eff > best_eff) { best_eff = eff; best = on_cu; c->mfw_nw = w; }
                }
                if (const char *e = getenv("MSDR_MFW_NW")) {     // A/B runs: a given number of waves per workgroup, where it fits (one workgroup per CU then)
                    const int w = atoi(e);
                    if (w >= 1 && w <= 16 && mw_lds_bytes(H, bsteps, w, fr) <= 160 * 1024) { c->mfw_nw = w; best = w * std::max<int>(1, std::min<int>((int)((160 * 1024) / mw_lds_bytes(H, bsteps, w, fr)), 16 / w)); }
                }
                c->mfw_waves_per_cu = best;
                if (best > 0) rc = dzalloc(ctx, (size_t)c->channels * kBqStateFloats, &c->d_bq_state_alt);
                if (!rc && iirfold && !iirc.empty()) rc = upload(ctx, iirc, &c->d_mw_iir);
                c->mfw_ssb_fold = iirfold && !compact && c->d_mw_iir; c->mfw_am_fold = amfold && c->d_mw_iir;
                if (best <= 0) c->mf_ok = false;               // not even one wave's window fits next to the fragments: the VALU kernels run
                // (full-rate layout: a lone wave per CU waits out every LDS round trip of a 60-step burst on its own -- two waves were 1.4 x the
                //  vector-ALU kernel at 384 taps, profiles/r05/nco_long_taps.txt; one is not taken)
                if (fr && c->mfw_nw < 2) c->mf_ok = false;
            }
        }
    }
    if (c->mf_fr && !c->mf_ok) { c->mf_fr = false; c->mf_P = 0; }
    // ---- envelope channels with the taps in registers (msdr_chain_amtr.hiph): the exact Fs/4 mixer, both FIRs of every tap set with the
    // same taps (the reference's AM case, Minimal-SDR.ino:917-924), up to 257 taps; rides on the wave-stream kernel's unit table ----
    if (!rc && f32 && c->mf_ok && c->mfw_nw > 0 && !c->mf_fr && cfg->mixer == MSDR_MIXER_FS4 && !(cfg->flags & MSDR_CHAIN_NO_MFMA) &&
        at_steps((int)c->ntaps) <= kAtMaxSteps && c->ntaps >= 2 &&
        // where it wins (256 taps with 0 / 1 biquad sections: 7 - 11 % faster than the wave-stream kernel in round 2, profiles/r02/am_matrix.txt;
        // 2 - 4 % after round 3's work on the other kernel, profiles/r03/c3_trims.txt; shorter filters or 2+ sections: equal or slower, the other
        // kernel runs the cascade on the matrix cores).  MSDR_AMTR=1 forces it (tests).
        ((at_steps((int)c->ntaps) == kAtMaxSteps && c->nstages <= 1) || getenv("MSDR_AMTR"))) {
        bool same = true;
        for (uint32_t s2 = 0; s2 < c->tapsets && same; s2++)
            same = memcmp(cfg->coeffs_i[s2], cfg->coeffs_q[s2], (size_t)c->ntaps * sizeof(float)) == 0;
        if (same) {
            const int N = (int)c->ntaps, H = at_halo(N), ns = at_steps(N);
            const size_t stride = at_table_bytes(ns);
            std::vector<char> blob(stride * c->tapsets, 0);
            for (uint32_t s2 = 0; s2 < c->tapsets; s2++) {
                const float *h = (const float *)cfg->coeffs_i[s2];
                double maxabs = 0.0;
                for (int k = 0; k < N; k++) maxabs = std::max(maxabs, std::fabs((double)h[k]));
                int ex = 0;
                if (maxabs > 0) { (void)std::frexp(maxabs, &ex); ex = 14 - ex; }             // maxabs 2^ex in [2^13, 2^14)
                AmTrHeader hd;
                memset(&hd, 0, sizeof hd);
                hd.ns = ns; hd.post = (float)((double)c->in_scale / std::ldexp(1.0, ex));
                memcpy(blob.data() + s2 * stride, &hd, sizeof hd);
                // g[d] = h_by_delay[d] j^-d: gR = h {1,0,-1,0}[d mod 4] (even delays), gI = h {0,-1,0,1}[d mod 4] (odd delays);
                // arm_fir keeps its taps time-reversed: delay d is pCoeffs[N - 1 - d]
                auto gR = [&](int d) -> double { return (d >= 0 && d < N && !(d & 1)) ? (double)h[N - 1 - d] * ((d & 2) ? -1.0 : 1.0) : 0.0; };
                auto gI = [&](int d) -> double { return (d >= 0 && d < N && (d & 1)) ? (double)h[N - 1 - d] * ((d & 2) ? 1.0 : -1.0) : 0.0; };
                _Float16 *tf = reinterpret_cast<_Float16 *>(blob.data() + s2 * stride + kAtHdrBytes);
                for (int m = 0; m < 3; m++)
                    for (int st = 0; st < ns; st++)
                        for (int l = 0; l < 64; l++)
                            for (int jj = 0; jj < 8; jj++) {
                                const int ap = l & 15, kp = 32 * st + 8 * (l >> 4) + jj;
                                const double v = (m == 0 ? gR(H + 2 * ap - 2 * kp) : m == 1 ? gI(H + 2 * ap - 2 * kp - 1) : gI(H + 2 * ap - 2 * kp + 1)) * std::ldexp(1.0, ex);
                                const _Float16 vh = (_Float16)v;
                                const size_t o = ((size_t)(m * ns + st) * 2) * 512 + l * 8 + jj;
                                tf[o] = vh; tf[o + 512] = (_Float16)(v - (double)vh);
                            }
            }
            int nw = 8;
            while (nw > 1 && at_lds_bytes(ns, nw) > 160 * 1024) nw--;
            if (at_lds_bytes(ns, nw) <= 160 * 1024) {
                rc = upload(ctx, blob, &c->d_at_tab);
                c->at_ns = ns; c->at_stride = (int)stride; c->at_nw = nw;
            }
        }
    }
    if (!rc && f32 && !c->d_fset) {       // per-channel table-set index (tap set x {LSB, USB, AM}), shared by the folded kernels
        std::vector<int> fs(c->channels);
        for (uint32_t ch = 0; ch < c->channels; ch++) {
            const int m = c->h_mode[ch];
            fs[ch] = c->h_tapset[ch] * 3 + (m == MSDR_MODE_LSB ? 0 : m == MSDR_MODE_USB ? 1 : 2);
        }
        rc = upload(ctx, fs, &c->d_fset);
    }
    for (uint32_t k = 0; k < c->nnodes && !rc; k++) {
        if (cfg->node_stages[k] < 1 || cfg->node_stages[k] > 4 || !cfg->node_coefs[k]) { rc = fail(MSDR_STATUS_ARGUMENT_ERROR, "biquad node %u misconfigured", k); break; }
        rc = msdr_biquad_q15_create(ctx, c->channels, &c->nodes[k]);
        for (uint32_t s = 0; s < cfg->node_stages[k] && !rc; s++)
            rc = msdr_biquad_q15_set_coefficients(c->nodes[k], s, cfg->node_coefs[k] + 5 * s);
    }
    if (!rc && !f32 && (cfg->flags & MSDR_CHAIN_SYNCAM_PLL)) rc = msdr_syncam_create(ctx, c->channels, &c->pll);
    if (rc) { chain_free(c); return rc; }
    if (f32 && c->nstages >= 3 && !c->mf_ok && c->fold_P == 0) {
        // Three or four sections behind chain_kernel<ArithF32> (no matrix-core tables, no folded tables: a long FIR behind a general
        // oscillator table): the only place where the block-parallel cascade runs with 12-sample lanes, and the one configuration in
        // round 3's fuzz records (profiles/r03/fuzz_kernels_606_final.txt: four sections, 1.07e-5) that no criterion excused.  The
        // kernel is the slow fallback anyway: the cascade runs section by section in CMSIS order behind it.
        chain_free(c);
        g_chain_force_seq_cascade = true;
        const int rc2 = chain_create_impl(ctx, cfg, out);
        g_chain_force_seq_cascade = false;
        return rc2;
    }
    *out = c;
    return 0;
}


// ---- rows f2 / f3 inside the fp32 chain: the post path (struct msdr_chain) ------------------------------------------------------
__global__ void post_hist_copy_kernel(const int16_t *__restrict__ src, int16_t *__restrict__ dst, const int *__restrict__ row, int hl_src, int hl_dst)
{
    // row v of dst = the newest min(hl_src, hl_dst) history samples of row[v] of src, older entries zero (both "oldest first")
    const int v = blockIdx.y;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < hl_dst; k += gridDim.x * blockDim.x) {
        const int back = hl_dst - k;                                  // 1 = newest
        dst[(long long)v * hl_dst + k] = (back <= hl_src) ? src[(long long)row[v] * hl_src + (hl_src - back)] : (int16_t)0;
    }
}

static const msdr_chain::Summary &chain_summary(msdr_chain *c)
{
    msdr_chain::Summary &u = c->sum;
    if (u.mode_gen != c->mode_gen) {
        u.any_ssb = u.any_env = u.any_syncam = false; u.qm_sets_ok = true;
        for (int m : c->h_mode) {
            if (m == MSDR_MODE_LSB || m == MSDR_MODE_USB) u.any_ssb = true; else u.any_env = true;
            if (m == MSDR_MODE_SYNCAM) u.any_syncam = true;
        }
        if (!c->qm_set_ok.empty()) for (int ts : c->h_tapset) if (!c->qm_set_ok[ts]) { u.qm_sets_ok = false; break; }
        u.mode_gen = c->mode_gen;
    }
    if (u.anr_gen != c->anr_gen) {
        u.any_anr = false;
        for (int v : c->h_anr) if (v > 0) { u.any_anr = true; break; }
        u.anr_gen = c->anr_gen;
    }
    return u;
}

static bool chain_post_wanted(const msdr_chain *c, uint32_t ch, bool *pll, int *anr)
{
    *pll = c->f32_pll && c->h_mode[ch] == MSDR_MODE_SYNCAM;
    *anr = c->h_anr.empty() ? 0 : c->h_anr[ch];
    return *pll || *anr > 0;
}

// (re)builds the auxiliary chain and the index tables for the current modes / LMS switches; FIR history and oscillator position are
// taken over from the main chain, PLL and LMS state are per CHANNEL and persist, the post cascade restarts
static int chain_post_build_steps(msdr_chain *c)
{
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    // the cascade state of channels that stay post channels goes with them (df1 stage state: 16 floats per row)
    std::vector<float> old_bq;
    const std::vector<int> old_ch = c->h_post_ch;
    if (c->post_bq && !old_ch.empty()) {
        old_bq.resize(old_ch.size() * kBqStateFloats);
        HIP_TRY(hipMemcpy(old_bq.data(), c->post_bq->d_state, old_bq.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    chain_post_free(c);
    c->h_post_ch.clear();
    std::vector<int> post_ch, post_src, post_pll, post_anr, virt_row, virt_mode, virt_ts;
    for (uint32_t ch = 0; ch < c->channels; ch++) {
        bool pll; int anr;
        if (!chain_post_wanted(c, ch, &pll, &anr)) continue;
        post_ch.push_back((int)ch); post_src.push_back((int)virt_row.size()); post_pll.push_back(pll ? 1 : 0); post_anr.push_back(anr);
        if (pll) {          // I + Q and I - Q: the two sideband flavours of the same filters
            virt_row.push_back((int)ch); virt_mode.push_back(MSDR_MODE_USB); virt_ts.push_back(c->h_tapset[ch]);
            virt_row.push_back((int)ch); virt_mode.push_back(MSDR_MODE_LSB); virt_ts.push_back(c->h_tapset[ch]);
        } else {
            virt_row.push_back((int)ch); virt_mode.push_back(c->h_mode[ch]); virt_ts.push_back(c->h_tapset[ch]);
        }
    }
    if (post_ch.empty()) return 0;
    msdr_chain_config a;
    memset(&a, 0, sizeof a);
    a.struct_size = sizeof a; a.arith = MSDR_ARITH_F32; a.channels = (uint32_t)virt_row.size(); a.mixer = c->mixer;
    a.num_taps = c->ntaps; a.num_tapsets = c->tapsets;
    for (uint32_t ts = 0; ts < c->tapsets; ts++) { a.coeffs_i[ts] = c->h_coef_i[ts].data(); a.coeffs_q[ts] = c->h_coef_q[ts].data(); }
    a.default_mode = MSDR_MODE_AM; a.mode = virt_mode.data(); a.tapset = virt_ts.data(); a.sqrt_kind = c->sqrt_kind;
    std::vector<float> oi(c->osc_len), oq(c->osc_len);
    for (uint32_t k = 0; k < c->osc_len; k++) { oq[k] = (float)c->h_osc[2 * k]; oi[k] = (float)c->h_osc[2 * k + 1]; }
    if (c->mixer == MSDR_MIXER_NCO) { a.osc_len = c->osc_len; a.osc_i = oi.data(); a.osc_q = oq.data(); }
    a.in_scale = c->in_scale; a.num_biquad_stages = 0; a.time_segments = 0;
    a.flags = c->flags & ~(uint32_t)(MSDR_CHAIN_SYNCAM_PLL | MSDR_CHAIN_OUT_I16);      // (the auxiliary chain hands fp32 to the PLL / LMS steps)
    if (int rc = msdr_chain_create(c->ctx, &a, &c->aux)) return rc;
    c->aux->phase = c->phase;
    for (const auto &o : c->osc_pending) {          // the auxiliary chain takes the same raw history over: the same tables apply to it
        msdr_chain::OscPending q{nullptr, o.elapsed};
        const size_t bytes = (size_t)c->osc_len * 2 * sizeof(float);
        HIP_TRY(hipMalloc(&q.d_tab, bytes));
        HIP_TRY(hipMemcpy(q.d_tab, o.d_tab, bytes, hipMemcpyDeviceToDevice));
        c->aux->osc_pending.push_back(q); c->aux->force_generic = true;
    }
    if (c->h_bq_stages) {
        if (int rc = msdr_biquad_df1_f32_create(c->ctx, (uint8_t)c->h_bq_stages, c->h_bq.data(), (uint32_t)post_ch.size(), &c->post_bq)) return rc;
        if (!old_bq.empty()) {
            std::vector<float> st(post_ch.size() * kBqStateFloats, 0.0f);
            for (size_t pn = 0; pn < post_ch.size(); pn++)
                for (size_t po = 0; po < old_ch.size(); po++)
                    if (old_ch[po] == post_ch[pn]) { memcpy(&st[pn * kBqStateFloats], &old_bq[po * kBqStateFloats], kBqStateFloats * sizeof(float)); break; }
            HIP_TRY(hipMemcpy(c->post_bq->d_state, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (int rc = upload(c->ctx, post_ch, &c->d_post_ch)) return rc;
    if (int rc = upload(c->ctx, post_src, &c->d_post_src)) return rc;
    if (int rc = upload(c->ctx, post_pll, &c->d_post_pll)) return rc;
    if (int rc = upload(c->ctx, post_anr, &c->d_post_anr)) return rc;
    if (int rc = upload(c->ctx, virt_row, &c->d_virt_row)) return rc;
    if (!c->d_post_pll_state) if (int rc = dzalloc(c->ctx, (size_t)c->channels * 4, &c->d_post_pll_state)) return rc;
    if (!c->d_post_anr_state) {
        std::vector<float> h((size_t)c->channels * kAnrStateFloats, 0.0f);
        for (uint32_t ch = 0; ch < c->channels; ch++) { h[(size_t)ch * kAnrStateFloats] = 120.0f; h[(size_t)ch * kAnrStateFloats + 1] = 0.001f; }   // .ino:715,:718
        if (int rc = upload(c->ctx, h, &c->d_post_anr_state)) return rc;
    }
    // the auxiliary chain continues the main chain's stream: same raw IF history (the newest samples both keep), same table position
    hipLaunchKernelGGL(post_hist_copy_kernel, dim3(4, (unsigned)virt_row.size()), dim3(256), 0, c->ctx->stream, (const int16_t *)c->d_hist[c->cur],
                       c->aux->d_hist[c->aux->cur], (const int *)c->d_virt_row, (int)c->hist_len, (int)c->aux->hist_len);
    if (int rc = launch_check("post_hist_copy_kernel")) return rc;
    // only a COMPLETE build counts (row counts and the channel list are what chain_post_run and the next rebuild go by)
    c->h_post_ch = post_ch;
    c->npost = (int)post_ch.size(); c->nvirt = (int)virt_row.size();
    return 0;
}

// The generation stamps are set only after every step succeeded.  A failed build leaves nothing half-made behind (no auxiliary
// chain, no post cascade, npost = 0) and the stamps stale, so the next msdr_chain_process tries again -- or returns the same error
// again: PLL / LMS channels never fall back silently to the main kernel's plain AM audio.
static int chain_post_build(msdr_chain *c)
{
    const int rc = chain_post_build_steps(c);
    if (rc != 0) {
        chain_post_free(c);
        c->h_post_ch.clear();
        return rc;
    }
    c->post_mode_gen = c->mode_gen; c->post_anr_gen = c->anr_gen;
    return 0;
}

// called by msdr_chain_process (F32) after the main kernel: replaces the rows of the post channels in d_audio
static int chain_post_run(msdr_chain *c, const int16_t *d_if, float *d_audio, uint64_t n)
{
    const bool any = c->f32_pll || chain_summary(c).any_anr;
    if (!any && !c->aux) return 0;
    if (c->post_mode_gen != c->mode_gen || c->post_anr_gen != c->anr_gen) if (int rc = chain_post_build(c)) return rc;
    if (c->npost == 0) return 0;
    if (n > 0xFFFFFFFFull) return fail(MSDR_STATUS_LENGTH_ERROR, "PLL / LMS channels of an fp32 chain take blocks below 2^32 samples");
    if (c->post_cap_virt < (size_t)c->nvirt * n || c->post_cap_post < (size_t)c->npost * n) {      // (a rebuild may have grown the row counts)
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        hipFree(c->d_aux_x); hipFree(c->d_aux_y); hipFree(c->d_post_scratch);
        c->d_aux_x = nullptr; c->d_aux_y = nullptr; c->d_post_scratch = nullptr; c->post_cap_virt = 0; c->post_cap_post = 0;
        const size_t cv = std::max(c->post_cap_virt, (size_t)c->nvirt * n), cp = std::max(c->post_cap_post, (size_t)c->npost * n);
        HIP_TRY(hipMalloc(&c->d_aux_x, cv * sizeof(int16_t)));
        HIP_TRY(hipMalloc(&c->d_aux_y, cv * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_post_scratch, cp * sizeof(float)));
        c->post_cap_virt = cv; c->post_cap_post = cp;
    }
    const unsigned gx = (unsigned)std::min<uint64_t>(64, (n + 255) / 256);
    hipLaunchKernelGGL(post_gather_rows_kernel, dim3(gx, c->nvirt), dim3(256), 0, c->ctx->stream, (const short *)d_if, (short *)c->d_aux_x,
                       (const int *)c->d_virt_row, (long long)n);
    if (int rc = launch_check("post_gather_rows_kernel")) return rc;
    if (int rc = msdr_chain_process(c->aux, c->d_aux_x, c->d_aux_y, n)) return rc;
    float kc[4];
    msdr_syncam_constants(kc);
    const SyncamConst k{kc[0], kc[1], kc[2], kc[3]};
    hipLaunchKernelGGL(post_pll_f32_kernel, dim3((c->npost + 63) / 64), dim3(64), 0, c->ctx->stream, (const float *)c->d_aux_y, c->d_post_scratch,
                       c->d_post_pll_state, (const int *)c->d_post_ch, (const int *)c->d_post_src, (const int *)c->d_post_pll, c->npost, (long long)n, k);
    if (int rc = launch_check("post_pll_f32_kernel")) return rc;
    const bool any_anr = chain_summary(c).any_anr;
    if (any_anr) {
        hipLaunchKernelGGL(post_anr_f32_kernel, dim3((c->npost + 63) / 64), dim3(64), 0, c->ctx->stream, c->d_post_scratch, c->d_post_anr_state,
                           (const int *)c->d_post_ch, (const int *)c->d_post_anr, c->npost, (long long)n);
        if (int rc = launch_check("post_anr_f32_kernel")) return rc;
    }
    if (c->post_bq) if (int rc = msdr_biquad_df1_f32_process(c->post_bq, c->d_post_scratch, c->d_post_scratch, (uint32_t)n)) return rc;
    hipLaunchKernelGGL(post_scatter_rows_kernel, dim3(gx, c->npost), dim3(256), 0, c->ctx->stream, (const float *)c->d_post_scratch, d_audio,
                       (const int *)c->d_post_ch, (long long)n);
    return launch_check("post_scatter_rows_kernel");
}

// fp32 audio -> int16 as arm_float_to_q15 (MSDR_CHAIN_OUT_I16 behind the kernels that do not convert in their own store phase)
__global__ void f32_to_q15_kernel(const float *__restrict__ src, short *__restrict__ dst, long long total)
{
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += stride) {
        if (i + 4 <= total && ((reinterpret_cast<uintptr_t>(src + i) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst + i) & 7) == 0))
            *reinterpret_cast<u32x2 *>(dst + i) = mw_q15x4(*reinterpret_cast<const f32x4 *>(src + i));
        else
            for (long long e = i; e < total && e < i + 4; e++) dst[e] = mw_q15(src[e]);
    }
}

// ---- block cadence (msdr_chain_mfb.hiph, msdr_chain_q15mb.hiph): the tile table ------------------------------------------------------------------
// CPT channels per tile, channels grouped by `key_of` (the table set / tap set a workgroup's waves share) inside two parts (`part_of`: SSB
// tables first, envelope tables after: two launches of two kernels); a workgroup's record = {key, 0, 0, 0} | nw waves x tpw tiles of ONE key, a
// group's tiles dealt round-robin to the waves, idle slots = -1.  Cached per (modes, n).
static int chain_block_tiles(msdr_chain *c, int n_, const std::function<int(uint32_t)> &part_of, const std::function<int(uint32_t)> &key_of,
                             const std::function<size_t(int, int)> &lds_of, int prefer_nw = 0)
{
    if (c->btiles_mode_gen == c->mode_gen && c->btiles_n == (long long)n_) return 0;
    const int cpt = mb_cpt(n_);
    std::vector<uint32_t> order(c->channels);
    for (uint32_t i = 0; i < c->channels; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return part_of(a) * 1000000 + key_of(a) < part_of(b) * 1000000 + key_of(b); });
    std::vector<int> tab;
    size_t i0 = 0;
    for (int part = 0; part < 2; part++) {
        std::vector<std::pair<size_t, size_t>> groups;          // [begin, end) in `order`
        size_t i = i0;
        while (i < order.size() && part_of(order[i]) == part) {
            size_t j = i;
            while (j < order.size() && part_of(order[j]) == part && key_of(order[j]) == key_of(order[i])) j++;
            groups.emplace_back(i, j);
            i = j;
        }
        i0 = i;
        // channels placed per tile: all CPT slots -- or fewer while full tiles would leave SIMDs without a wave (a tile is one wave's work: its
        // staging and stores shrink with the channels in it, its matrix products do not, and an idle SIMD is worth nothing): halve the fill
        // until the part has a tile for every second SIMD (4096 channels x 128: 512 full tiles on 1024 SIMDs -> 1024 tiles of 4 channels)
        int fill = cpt;
        auto tiles_at = [&](int f) { long long t = 0; for (auto &g : groups) t += (long long)((g.second - g.first) + f - 1) / f; return t; };
        if (const char *e = getenv("MSDR_MB_FILL")) fill = std::max(1, std::min(cpt, atoi(e)));
        else while (fill > 1 && tiles_at(fill) * 2 <= (long long)c->ctx->num_cus * 4) fill >>= 1;
        const long long tiles_part = tiles_at(fill);
        msdr_chain::BlockPart &bp = c->bpart[part];
        bp = msdr_chain::BlockPart();
        bp.offset = tab.size();
        if (tiles_part == 0) continue;
        // waves per workgroup / tiles per wave: the least time per workgroup (below), ties to fewer waves; one workgroup per CU at these LDS sizes
        // (MSDR_MB_NW: experiments / tests; a value that does not fit is ignored -- the loop below must always end with nw, tpw >= 1:
        //  the caller checked that one wave with one tile fits)
        int force_nw = 0;
        if (const char *e = getenv("MSDR_MB_NW")) {
            force_nw = std::max(1, std::min(8, atoi(e)));
            long long tpw = (tiles_part + (long long)c->ctx->num_cus * force_nw - 1) / ((long long)c->ctx->num_cus * force_nw);
            tpw = std::max<long long>(1, std::min<long long>(tpw, kMbMaxTableInts / cpt));
            if (lds_of(force_nw, (int)tpw) > 160 * 1024) force_nw = 0;
        }
        long long best_cost = -1;
        for (int w = force_nw ? force_nw : 1; w <= (force_nw ? force_nw : 8); w++) {
            long long tpw = (tiles_part + (long long)c->ctx->num_cus * w - 1) / ((long long)c->ctx->num_cus * w);
            tpw = std::max<long long>(1, std::min<long long>(tpw, kMbMaxTableInts / cpt));           // (a longer list: more workgroups than CUs)
            if (lds_of(w, (int)tpw) > 160 * 1024) break;
            // time of a workgroup ~ tiles per wave x what its fullest SIMD carries: a second wave on a SIMD fills the first one's
            // waits (two tiles in ~1.5 x the time of one: profiles/r05/mfb_nw_sweep.txt)
            const long long cost = tpw * (w <= 4 ? 2 : 3);
            // (prefer_nw: among equal costs, at least that many waves -- a Q15 chain whose biquad nodes can run as the kernel's second phase
            //  needs three waves per workgroup for it, also where one would do for the tiles: the single receiver)
            if (best_cost < 0 || cost < best_cost || (cost == best_cost && (int)bp.nw < prefer_nw && w <= prefer_nw)) { best_cost = cost; bp.nw = (uint32_t)w; bp.tpw = (uint32_t)tpw; }
        }
        if (bp.nw == 0 || bp.tpw == 0) { bp.nw = 1; bp.tpw = 1; }
        const size_t per_wg = (size_t)bp.nw * bp.tpw;
        for (auto &g : groups) {
            const size_t tiles_g = ((g.second - g.first) + fill - 1) / fill;
            for (size_t t0 = 0; t0 < tiles_g; t0 += per_wg) {
                tab.push_back(key_of(order[g.first])); tab.push_back(0); tab.push_back(0); tab.push_back(0);      // kMbRecHdrInts
                const size_t base = tab.size();
                tab.resize(base + per_wg * cpt, -1);
                const size_t cnt = std::min(per_wg, tiles_g - t0);
                for (size_t t = 0; t < cnt; t++) {           // tile t of this workgroup -> wave t % nw, its slot t / nw
                    const size_t slot = (t % bp.nw) * bp.tpw + t / bp.nw;
                    for (int k = 0; k < fill; k++) {
                        const size_t idx = g.first + (t0 + t) * fill + k;
                        if (idx < g.second) tab[base + slot * cpt + k] = (int)order[idx];
                    }
                }
                bp.wgs++;
            }
        }
    }
    if (tab.size() > c->btiles_cap) {
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        hipFree(c->d_btiles); c->d_btiles = nullptr; c->btiles_cap = 0;
        if (int rc = dzalloc(c->ctx, tab.size(), &c->d_btiles)) return rc;
        c->btiles_cap = tab.size();
    }
    HIP_TRY(hipMemcpyAsync(c->d_btiles, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, c->ctx->stream));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));        // `tab` is a local
    c->btiles_mode_gen = c->mode_gen; c->btiles_n = (long long)n_;
    return 0;
}

static int chain_leave_generic(msdr_chain *c);
extern "C" int msdr_chain_process(msdr_chain *c, const int16_t *d_if, void *d_audio, uint64_t n_samples)
{
    if (!c) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null chain");
    if (int rc = bind(c->ctx)) return rc;
    if (n_samples == 0) return 0;
    if (!d_if || !d_audio) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer");
    if (n_samples > (1ull << 31) - 4096) return fail(MSDR_STATUS_LENGTH_ERROR, "n_samples too large for one call");
    const bool f32 = (c->arith == MSDR_ARITH_F32);
    if (!f32 && c->nnodes && (n_samples & 1u))
        return fail(MSDR_STATUS_LENGTH_ERROR, "AudioFilterBiquad processes sample pairs: n_samples must be even");

    ChainParams p;
    memset(&p, 0, sizeof p);
    p.x = d_if; p.out = d_audio; p.hist_in = c->d_hist[c->cur]; p.hist_out = c->d_hist[c->cur ^ 1];
    p.n = (long long)n_samples; p.channels = (int)c->channels;
    p.ntaps_pad = (int)c->ntaps_pad; p.hist_len = (int)c->hist_len; p.taps = c->d_taps;
    p.chan_mode = c->d_mode; p.chan_tapset = c->d_tapset; p.mixer = c->mixer; p.osc = c->d_osc; p.osc_len = (int)c->osc_len;
    p.phase0 = (int)c->phase; p.in_scale = c->in_scale; p.sqrt_kind = c->sqrt_kind;
    p.nstages = (int)c->nstages; p.bq = c->d_bq; p.bq_state = c->d_bq_state;

    // ---- kernel choice: the folded kernel needs a short-period oscillator; AM additionally the exact Fs/4 pattern
    bool use_fold = f32 && c->fold_P > 0;
    if (use_fold && !c->fold_fs4_exact)
        if (chain_summary(c).any_env) use_fold = false;
    bool pll_active = false;
    if (c->pll) {
        pll_active = chain_summary(c).any_syncam;
        if (pll_active) {
            const size_t need = (size_t)c->channels * n_samples;
            if (need > c->pll_q_cap) {
                HIP_TRY(hipStreamSynchronize(c->ctx->stream));
                hipFree(c->d_pll_q); c->d_pll_q = nullptr; c->pll_q_cap = 0;
                HIP_TRY(hipMalloc((void **)&c->d_pll_q, need * sizeof(int16_t)));
                c->pll_q_cap = need;
            }
            p.syncam_q = c->d_pll_q;
        }
    }
    // (history samples that arrived under an earlier oscillator table are still in reach, msdr_chain_set_osc: the as-written kernel mixes
    //  each with the table of its own time; the fast kernels come back once the history has turned over)
    if (c->force_generic) use_fold = false;
    const bool use_mf = f32 && c->mf_ok && !c->force_generic;
    const bool use_mfw = use_mf && c->mfw_nw > 0;
    bool use_qm = !f32 && c->d_qm_tab != nullptr && !c->force_generic;
    p.osc_hist = nullptr;
    if (!c->osc_pending.empty()) {
        OscHistory oh;
        memset(&oh, 0, sizeof oh);
        for (const auto &o : c->osc_pending) {
            if (oh.n >= kOscHistMax) break;
            oh.tab[oh.n] = o.d_tab; oh.sw[oh.n] = -o.elapsed; oh.n++;
        }
        if (!c->d_osc_hist) HIP_TRY(hipMalloc((void **)&c->d_osc_hist, sizeof(OscHistory)));
        HIP_TRY(hipMemcpyAsync(c->d_osc_hist, &oh, sizeof oh, hipMemcpyHostToDevice, c->ctx->stream));
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));                  // (`oh` is a local; this path runs for one history length after a table change)
        p.osc_hist = c->d_osc_hist;
    }
    // ---- MSDR_CHAIN_OUT_I16: int16 audio.  The matrix-core kernels convert in their store phase when nothing runs behind them;
    // otherwise the fp32 audio goes to a scratch block batch and is converted last.
    void *fout = d_audio;                                        // where the fp32 passes of this call read and write the audio
    bool i16_via_scratch = false;
    if (f32 && (c->flags & MSDR_CHAIN_OUT_I16)) {
        const bool post_active = c->f32_pll || c->aux != nullptr || chain_summary(c).any_anr;
        if (use_mfw && !c->seq_bq && !post_active) p.dbg |= kChainOutI16;
        else {
            const size_t need = (size_t)c->channels * n_samples;
            if (need > c->f32_scratch_floats) {
                HIP_TRY(hipStreamSynchronize(c->ctx->stream));
                hipFree(c->d_f32_scratch); c->d_f32_scratch = nullptr; c->f32_scratch_floats = 0;
                HIP_TRY(hipMalloc((void **)&c->d_f32_scratch, need * sizeof(float)));
                c->f32_scratch_floats = need;
            }
            fout = c->d_f32_scratch; p.out = fout; i16_via_scratch = true;
        }
    }
    if (use_qm)
        if (!chain_summary(c).qm_sets_ok) use_qm = false;                                  // a tap >= 32640: the VALU kernel runs
    if (use_mf) use_fold = false;
    const int kTile = use_mfw ? kMwTile : use_fold ? kFoldTile : kChainTile;
    // ---- block cadence (msdr_chain_mfb.hiph): one AUDIO_BLOCK (or another short block) per call -- channel-batched tiles, the next
    // history written by the kernel itself.  Taken where the wave-stream kernel would run its folded flavours (or no cascade at all).
    bool use_mfb = false;
    if (use_mfw && !c->block_off && !c->mf_fr && c->nstages <= 2 && mb_n_ok((long long)n_samples) && (int)c->hist_len == c->mf_halo &&
        (reinterpret_cast<uintptr_t>(d_if) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 &&
        mb_lds_bytes(c->mf_halo, (int)n_samples, c->mf_bsteps, 1, 1) <= 160 * 1024 &&
        (uint64_t)c->channels * std::max<uint64_t>((uint64_t)c->hist_len * 2, n_samples * 4) < (1ull << 32)) {       // (the kernel addresses with 32-bit byte offsets)
        const bool need_ssb = chain_summary(c).any_ssb, need_env = chain_summary(c).any_env;
        use_mfb = c->nstages == 0 || ((!need_ssb || c->mfw_ssb_fold) && (!need_env || c->mfw_am_fold));
    }
    p.mf_tab = c->d_mf_tab; p.mf_stride = c->mf_stride; p.mf_halo = c->mf_halo; p.mf_bsteps = c->mf_bsteps; p.bq_mf = c->d_bq_mf; p.bq_mf32 = c->d_bq_mf32;
    // the same for the Q15 chain (msdr_chain_q15mb.hiph); SYNCAM channels under the PLL hand I and Q to a kernel behind: the streaming kernel
    bool use_qb = false;
    if (use_qm && chain_summary(c).qm_sets_ok && !c->block_off && !c->qm_fr && !pll_active && mb_n_ok((long long)n_samples) && (int)c->hist_len == c->qm_halo &&
        (reinterpret_cast<uintptr_t>(d_if) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_audio) & 15) == 0 &&
        qb_lds_bytes(c->qm_halo, (int)n_samples, c->qm_bsteps, 1, 1) <= 160 * 1024 &&
        (uint64_t)c->channels * std::max<uint64_t>((uint64_t)c->hist_len * 2, n_samples * 2) < (1ull << 32)) use_qb = true;
    const int osc_P = use_mf ? c->mf_P : c->fold_P;            // the matrix-core tables exist for periods up to 32, the VALU fold tables up to 4
    p.ftaps = c->d_ftaps; p.chan_fset = c->d_fset; p.fold_period = osc_P; p.bq_fold = c->d_bq_fold;
    p.fold_rot = osc_P ? (int)(c->phase % osc_P) : 0;

    // ---- time segmentation (DESIGN.md "IIR along time") ------------------------------------------
    const long long tiles = ((long long)n_samples + kTile - 1) / kTile;
    long long warm_tiles = 0;
    bool can_split = true;
    if (c->nstages) {
        long long w = c->warmup_cfg;
        if (w == 0) {
            if (c->pole_radius >= 0.99999) can_split = false;        // marginal/unstable: never re-converges
            else if (c->pole_radius > 0) w = (long long)std::ceil(std::log(use_mfw ? 1e-8 : 1e-10) / std::log(c->pole_radius)) + 64 * c->nstages;
        }
        warm_tiles = (w + kTile - 1) / kTile;
#if defined(MSDR_MUTATE) && MSDR_MUTATE == 3           /* `make mutants`, never the product: time segments start their cascade from zero state, no re-convergence */
        warm_tiles = 0;
#endif
        if (warm_tiles > 64 * (use_mfw ? 4 : 1)) can_split = false;
    }
    // segment count for `nch` channels that are launched together
    auto choose_nseg = [&](long long nch) -> long long {
        long long ns_ = 1;
        if (c->time_segments == 1 || !can_split || nch <= 0) return 1;
        if (use_mfw && c->time_segments == 0) {
            // one unit per wave, all units equally long: the launch takes ceil(units / resident waves) rounds of (segment + warm-up)
            // tiles.  Pick the segment count that minimises that product (an exact multiple of the resident waves wins).
            const long long slots = (long long)c->mfw_waves_per_cu * c->ctx->num_cus;
            const long long max_nseg = std::max<long long>(1, tiles / std::max<long long>(4, 8 * warm_tiles));
            const long long kmax = std::min<long long>(max_nseg, std::max<long long>(1, (8 * slots + nch - 1) / nch));
            double best = 1e300;
            for (long long k = 1; k <= kmax; k++) {
                const long long st = (tiles + k - 1) / k, ns = (tiles + st - 1) / st;
                const long long rounds = (nch * ns + slots - 1) / slots;
                const double cost = (double)rounds * (double)(st + (ns > 1 ? warm_tiles : 0));
                if (cost < best * 0.999) { best = cost; ns_ = ns; }
            }
        } else {
            long long min_seg_tiles = std::max<long long>(4, 32 * warm_tiles);   // <= ~3 % redone work
            long long max_nseg = std::max<long long>(1, tiles / min_seg_tiles);
            long long want = c->time_segments > 1 ? c->time_segments : std::max<long long>(1, (2048 + nch - 1) / nch);
            ns_ = std::max<long long>(1, std::min(want, max_nseg));
        }
        const long long st = (tiles + ns_ - 1) / ns_;
        return (tiles + st - 1) / st;
    };
    long long nseg = choose_nseg(c->channels);
    long long seg_tiles = (tiles + nseg - 1) / nseg;
    p.nseg = (int)nseg; p.seg_len = seg_tiles * kTile; p.warm = (int)(nseg > 1 ? warm_tiles * kTile : 0);

    unsigned grid = (unsigned)(c->channels * nseg);
    if (use_mfb) {
        auto fset_of = [&](uint32_t ch) { const int m = c->h_mode[ch]; return c->h_tapset[ch] * 3 + (m == MSDR_MODE_LSB ? 0 : m == MSDR_MODE_USB ? 1 : 2); };
        if (int rc = chain_block_tiles(c, (int)n_samples, [&](uint32_t ch) { return fset_of(ch) % 3 == 2 ? 1 : 0; }, fset_of,
                                       [&](int w, int tpw) { return mb_lds_bytes(c->mf_halo, (int)n_samples, c->mf_bsteps, w, tpw); })) return rc;
        nseg = 1; p.nseg = 1; p.warm = 0;
        p.bq_state_out = c->d_bq_state_alt; p.mw_iir = c->d_mw_iir;
    } else if (use_qb) {
        if (int rc = chain_block_tiles(c, (int)n_samples, [&](uint32_t ch) { const int m = c->h_mode[ch]; return (m == MSDR_MODE_LSB || m == MSDR_MODE_USB) ? 0 : 1; },
                                       [&](uint32_t ch) { return c->h_tapset[ch]; },
                                       [&](int w, int tpw) { return qb_lds_bytes(c->qm_halo, (int)n_samples, c->qm_bsteps, w, tpw); },
                                       (c->nnodes == 2 && c->nodes[0]->max_stage == 0 && c->nodes[1]->max_stage == 0 && n_samples == 128 && !c->no_fuse) ? 3 : 0)) return rc;
        nseg = 1; p.nseg = 1; p.warm = 0;
    } else
    if (use_mfw) {
        // unit table: (channel, segment) per wave; the waves of a workgroup share one tap table, so channels are grouped by
        // table set and every group is padded to whole workgroups.  SSB-table units and envelope-table units are two launches
        // (two kernels), each with the segmentation that fills the GPU with its own channels.
        const int nw = c->mfw_nw;
        if (c->units_mode_gen != c->mode_gen || c->units_tiles != tiles) {
            std::vector<uint32_t> order(c->channels);
            for (uint32_t i = 0; i < c->channels; i++) order[i] = i;
            auto fset_of = [&](uint32_t ch) { const int m = c->h_mode[ch]; return c->h_tapset[ch] * 3 + (m == MSDR_MODE_LSB ? 0 : m == MSDR_MODE_USB ? 1 : 2); };
            auto env_of = [&](uint32_t ch) { return fset_of(ch) % 3 == 2 ? 1 : 0; };
            auto key_of = [&](uint32_t ch) { return env_of(ch) * 1000000 + fset_of(ch); };      // SSB tables first, envelope tables after
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key_of(a) < key_of(b); });
            long long count[2] = {0, 0};
            for (uint32_t ch : order) count[env_of(ch)]++;
            for (int part = 0; part < 2; part++) {
                c->part_nseg[part] = choose_nseg(count[part]);
                c->part_seg_len[part] = ((tiles + c->part_nseg[part] - 1) / c->part_nseg[part]) * kTile;
            }
            std::vector<int> units;
            units.reserve(((size_t)c->channels * std::max(c->part_nseg[0], c->part_nseg[1]) + 3 * MSDR_MAX_TAPSETS * nw) * 2);
            size_t ssb_units = 0;
            for (size_t i = 0; i < order.size(); i++) {
                if (i > 0 && fset_of(order[i]) != fset_of(order[i - 1]))
                    while ((units.size() / 2) % nw) { units.push_back(-1); units.push_back(0); }
                if (env_of(order[i]) && (i == 0 || !env_of(order[i - 1]))) ssb_units = units.size() / 2;      // first envelope-table unit
                for (long long sg = 0; sg < c->part_nseg[env_of(order[i])]; sg++) { units.push_back((int)order[i]); units.push_back((int)sg); }
            }
            while ((units.size() / 2) % nw) { units.push_back(-1); units.push_back(0); }
            if (order.empty() || !env_of(order.back())) ssb_units = units.size() / 2;                               // no envelope tables at all
            if (units.size() > c->units_cap) {
                HIP_TRY(hipStreamSynchronize(c->ctx->stream));
                hipFree(c->d_units); c->d_units = nullptr; c->units_cap = 0;
                if (int rc = dzalloc(c->ctx, units.size(), &c->d_units)) return rc;
                c->units_cap = units.size();
            }
            HIP_TRY(hipMemcpyAsync(c->d_units, units.data(), units.size() * sizeof(int), hipMemcpyHostToDevice, c->ctx->stream));
            HIP_TRY(hipStreamSynchronize(c->ctx->stream));        // `units` is a local
            c->units_mode_gen = c->mode_gen; c->units_tiles = tiles; c->units_wgs = (uint32_t)(units.size() / 2 / nw);
            c->units_wgs_ssb = (uint32_t)(ssb_units / nw);
        }
        nseg = (c->units_wgs_ssb > 0) ? c->part_nseg[0] : c->part_nseg[1];            // reported by msdr_chain_get_info
        p.warm = (int)(nseg > 1 ? warm_tiles * kTile : 0);
        grid = c->units_wgs;
        p.mf_units = c->d_units; p.mf_nw = nw; p.bq_state_out = c->d_bq_state_alt; p.mw_iir = c->d_mw_iir;
#ifdef MSDR_STAMPS
        { const char *e = getenv("MSDR_DBG"); p.dbg = (p.dbg & kChainOutI16) | (e ? (atoi(e) & ~kChainOutI16) : 0); }
        static unsigned long long *stamp_buf = nullptr;
        const size_t stamp_n = (size_t)grid * nw * 8;
        if (!stamp_buf) HIP_TRY(hipMalloc(&stamp_buf, (1u << 20) * 8 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(stamp_buf, 0, stamp_n * 8, c->ctx->stream));
        p.dbg_buf = stamp_buf;
#endif
    }

    if (c->dry_run) {
        // msdr_chain_graph_create's preparation pass: everything a block-cadence call needs from the host is in place now; say whether the
        // launches that follow are fixed (capturable into a HIP graph) -- and make none
        if (!use_mfb && !use_qb) return fail(MSDR_STATUS_ARGUMENT_ERROR, "not a block-cadence call (32 .. 512 samples, a divisor of 1024, 16-byte aligned buffers, matrix-core tables, no pending oscillator change): nothing to capture");
        if (c->seq_bq) return fail(MSDR_STATUS_ARGUMENT_ERROR, "the cascade runs in CMSIS order behind the kernel (a kernel of its own with host-side sizing): not capturable");
        if (i16_via_scratch) return fail(MSDR_STATUS_ARGUMENT_ERROR, "int16 audio through the scratch batch: not capturable");
        if (f32 && (c->f32_pll || c->aux || chain_summary(c).any_anr)) return fail(MSDR_STATUS_ARGUMENT_ERROR, "PLL / LMS channels run behind the kernel through an auxiliary chain: not capturable");
        if (!f32 && c->anr && (c->d_anr_on || c->anr_all > 0)) return fail(MSDR_STATUS_ARGUMENT_ERROR, "the LMS filter runs behind the kernel: not capturable");
        if (n_samples % c->osc_len) return fail(MSDR_STATUS_ARGUMENT_ERROR, "the oscillator's position changes from call to call at this block length: not capturable");
        return 0;
    }
    const size_t lds = use_mfw ? mw_lds_bytes(c->mf_halo, c->mf_bsteps, c->mfw_nw, c->mf_fr) : use_fold ? fold_lds_bytes(p.ntaps_pad) : chain_lds_bytes(p.ntaps_pad);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing && c->events.size() < 8192) {
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, c->ctx->stream));
    }
    const char *kname = f32 ? "chain_kernel<ArithF32>" : "chain_kernel<ArithQ15>";
    bool nodes_fused = false;            // Q15 block cadence: the biquad nodes ran inside chain_q15mb_kernel
    unsigned block = kThreads;
    size_t lds_used = lds;
    if (use_mfb) {
        static const char *const names[3] = {"chain_mfb_kernel<0> (channel-batched block tiles)", "chain_mfb_kernel<1> (channel-batched block tiles)",
                                             "chain_mfb_kernel<2> (channel-batched block tiles)"};
        grid = 0;
        for (int part = 0; part < 2; part++) {
            const msdr_chain::BlockPart &bp = c->bpart[part];
            if (bp.wgs == 0) continue;
            ChainParams q = p;
            q.mf_units = c->d_btiles + bp.offset; q.mf_nw = (int)bp.nw; q.nseg = (int)bp.tpw;
            lds_used = mb_lds_bytes(c->mf_halo, (int)n_samples, c->mf_bsteps, (int)bp.nw, (int)bp.tpw);
            if (launch_chain_mfb(c->ctx->stream, (int)c->nstages, part == 1, bp.wgs, bp.nw * 64, lds_used, q) != hipSuccess)
                return fail(MSDR_STATUS_HIP_ERROR, "chain_mfb_kernel launch failed");
            grid += bp.wgs; block = bp.nw * 64;
        }
        kname = names[c->nstages];
        if (c->nstages > 0) std::swap(c->d_bq_state, c->d_bq_state_alt);      // the kernel read bq_state and wrote bq_state_out
    } else
    if (use_mfw) {
        block = (unsigned)c->mfw_nw * 64;
        // SSB-table units and envelope-table units are separate launches of separate kernels (register allocation per flavour)
        for (int part = 0; part < 2; part++) {
            const unsigned g = part == 0 ? c->units_wgs_ssb : c->units_wgs - c->units_wgs_ssb;
            if (g == 0) continue;
            ChainParams q = p;
            q.mf_units = c->d_units + (size_t)(part == 0 ? 0 : c->units_wgs_ssb) * c->mfw_nw * 2;
            q.nseg = (int)c->part_nseg[part]; q.seg_len = c->part_seg_len[part]; q.warm = (int)(c->part_nseg[part] > 1 ? warm_tiles * kTile : 0);
            const bool fold = part == 0 ? c->mfw_ssb_fold : c->mfw_am_fold;
            if (part == 1 && c->d_at_tab) {
                // envelope units on the taps-in-registers kernel: the same unit table read with this kernel's own workgroup size
                ChainParams a = q;
                const long long entries = (long long)g * c->mfw_nw;
                a.mf_tab = c->d_at_tab; a.mf_stride = c->at_stride; a.mf_nw = c->at_nw; a.fold_rot = (int)entries;
                const unsigned ag = (unsigned)((entries + c->at_nw - 1) / c->at_nw);
                const size_t alds = at_lds_bytes(c->at_ns, c->at_nw);
                (void)launch_chain_amtr(c->ctx->stream, c->at_ns, (int)c->nstages, ag, (unsigned)c->at_nw * 64, alds, a);
                if (int rc2 = launch_check("chain_amtr_kernel")) return rc2;
                continue;
            }
            (void)launch_chain_mfw(c->ctx->stream, (int)c->nstages, part == 1, fold, c->mf_fr, g, block, lds, q);
            if (int rc = launch_check("chain_mfw_kernel")) return rc;
        }
        static const char *const names[5] = {"chain_mfw_kernel<0>", "chain_mfw_kernel<1>", "chain_mfw_kernel<2>", "chain_mfw_kernel<3>", "chain_mfw_kernel<4>"};
        static const char *const names_fr[5] = {"chain_mfw_kernel<0> full-rate NCO streams", "chain_mfw_kernel<1> full-rate NCO streams", "chain_mfw_kernel<2> full-rate NCO streams",
                                                "chain_mfw_kernel<3> full-rate NCO streams", "chain_mfw_kernel<4> full-rate NCO streams"};
        kname = c->mf_fr ? names_fr[c->nstages] : names[c->nstages];
        if (c->d_at_tab && c->units_wgs > c->units_wgs_ssb) kname = c->units_wgs_ssb ? "chain_mfw_kernel + chain_amtr_kernel" : "chain_amtr_kernel";
        std::swap(c->d_bq_state, c->d_bq_state_alt);          // the kernel read bq_state and wrote bq_state_out
#ifdef MSDR_STAMPS
        if (getenv("MSDR_STAMP_PRINT")) {
            HIP_TRY(hipStreamSynchronize(c->ctx->stream));
            std::vector<unsigned long long> h((size_t)grid * c->mfw_nw * 8);
            HIP_TRY(hipMemcpy(h.data(), p.dbg_buf, h.size() * 8, hipMemcpyDeviceToHost));
            double sum[8] = {0}; size_t cnt = 0;
            for (size_t u = 0; u < h.size() / 8; u++) { if (!h[u * 8 + 0] && !h[u * 8 + 1]) continue; cnt++; for (int k = 0; k < 8; k++) sum[k] += (double)h[u * 8 + k]; }
            const double tl = (double)((n_samples + kMwTile - 1) / kMwTile) * c->channels / std::max<size_t>(cnt, 1);
            fprintf(stderr, "stamps (cycles per tile per wave, %zu units, %.1f tiles each): prefetch-issue %.0f | mfma %.0f | demod+swap %.0f | iir %.0f | store %.0f | halo %.0f | vmwait %.0f | stage %.0f\n",
                    cnt, tl, sum[0] / cnt / tl, sum[1] / cnt / tl, sum[2] / cnt / tl, sum[3] / cnt / tl, sum[4] / cnt / tl, sum[5] / cnt / tl, sum[6] / cnt / tl, sum[7] / cnt / tl);
        }
#endif
    }
    else if (use_fold) {
        (void)launch_chain_fold(c->ctx->stream, c->fold_P, grid, lds, p);
        kname = c->fold_P == 4 ? "chain_fold_kernel<4>" : c->fold_P == 2 ? "chain_fold_kernel<2>" : "chain_fold_kernel<1>";
    }
    else if (f32) (void)launch_chain_generic(c->ctx->stream, false, grid, lds, p);
    else if (use_qb) {
        grid = 0;
        // the two biquad nodes as the kernel's second phase: the reference's configuration (one stage per node, nothing between the
        // demodulator and the nodes), one 128-sample block, and every launch of this call with one tile per wave on three or more waves
        // (small batches: up to 16 384 channels of one flavour on this part); MSDR_Q15_NO_FUSE=1 at create time: the node kernel behind it as before
        nodes_fused = c->nnodes == 2 && c->nodes[0]->max_stage == 0 && c->nodes[1]->max_stage == 0 && n_samples == 128 && !pll_active &&
                      !(c->anr && (c->d_anr_on || c->anr_all > 0)) && !c->no_fuse;
        for (int part = 0; part < 2 && nodes_fused; part++) {
            const msdr_chain::BlockPart &bp = c->bpart[part];
            if (bp.wgs && (bp.tpw != 1 || bp.nw < 3 || qb_nodes_lds_bytes(c->qm_halo, 128, c->qm_bsteps, (int)bp.nw) > 160 * 1024)) nodes_fused = false;
        }
        for (int part = 0; part < 2; part++) {
            const msdr_chain::BlockPart &bp = c->bpart[part];
            if (bp.wgs == 0) continue;
            ChainParams q = p;
            q.mf_tab = c->d_qm_tab; q.mf_stride = c->qm_stride; q.mf_halo = c->qm_halo; q.mf_bsteps = c->qm_bsteps;
            q.mf_units = c->d_btiles + bp.offset; q.mf_nw = (int)bp.nw; q.nseg = (int)bp.tpw;
            lds_used = nodes_fused ? qb_nodes_lds_bytes(c->qm_halo, 128, c->qm_bsteps, (int)bp.nw) : qb_lds_bytes(c->qm_halo, (int)n_samples, c->qm_bsteps, (int)bp.nw, (int)bp.tpw);
            if (nodes_fused) { q.bq_state = reinterpret_cast<float *>(c->nodes[0]->d_defs); q.bq_state_out = reinterpret_cast<float *>(c->nodes[1]->d_defs); }
            const int flavour = part == 0 ? 0 : (c->sqrt_kind == 1 ? 2 : 1);
            if (launch_chain_q15mb(c->ctx->stream, flavour, nodes_fused, bp.wgs, bp.nw * 64, lds_used, q) != hipSuccess)
                return fail(MSDR_STATUS_HIP_ERROR, "chain_q15mb_kernel launch failed");
            grid += bp.wgs; block = bp.nw * 64;
        }
        kname = nodes_fused ? "chain_q15mb_kernel (block tiles) + both biquad nodes" : "chain_q15mb_kernel (channel-batched block tiles)";
    }
    else if (use_qm) {
        // channels grouped by (tap set, flavour): a workgroup's waves share one table; one launch per group in use
        const int env_flavour = (c->sqrt_kind == 1) ? 2 : 1;
        auto flavour_of = [&](uint32_t ch) { const int m = c->h_mode[ch]; return (m == MSDR_MODE_LSB || m == MSDR_MODE_USB) ? 0 : env_flavour; };
        if (c->qm_order_gen != c->mode_gen) {
            std::vector<int> order;
            c->qm_group_start.assign((size_t)c->tapsets * 3, 0); c->qm_group_count.assign((size_t)c->tapsets * 3, 0);
            for (uint32_t s = 0; s < c->tapsets; s++)
                for (int fl = 0; fl < 3; fl++) {
                    const size_t gi = (size_t)s * 3 + fl;
                    c->qm_group_start[gi] = (uint32_t)order.size();
                    for (uint32_t ch = 0; ch < c->channels; ch++)
                        if ((uint32_t)c->h_tapset[ch] == s && flavour_of(ch) == fl) order.push_back((int)ch);
                    c->qm_group_count[gi] = (uint32_t)order.size() - c->qm_group_start[gi];
                }
            if (!c->d_qm_order) HIP_TRY(hipMalloc((void **)&c->d_qm_order, (size_t)c->channels * sizeof(int)));
            HIP_TRY(hipMemcpyAsync(c->d_qm_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, c->ctx->stream));
            HIP_TRY(hipStreamSynchronize(c->ctx->stream));
            c->qm_order_gen = c->mode_gen;
        }
        const long long qtiles = ((long long)n_samples + kQmTile - 1) / kQmTile;
        ChainParams q = p;
        q.mf_tab = c->d_qm_tab; q.mf_stride = c->qm_stride; q.mf_halo = c->qm_halo; q.mf_bsteps = c->qm_bsteps; q.warm = 0; q.mf_units = c->d_qm_order;
        int nw = 8;
        long long qseg = 1;
        for (size_t gi = 0; gi < c->qm_group_count.size(); gi++) {
            const long long cnt = c->qm_group_count[gi];
            if (!cnt) continue;
            // every launch fills the GPU on its own: aim at two rounds of 16 waves per CU, at least two tiles per segment
            qseg = (8192 + cnt - 1) / cnt;
            qseg = std::max<long long>(1, std::min<long long>(qseg, std::max<long long>(1, qtiles / 2)));
            if (c->time_segments > 0) qseg = std::max<long long>(1, std::min<long long>(c->time_segments, qtiles));
            const long long qseg_len = ((qtiles + qseg - 1) / qseg) * kQmTile;
            qseg = ((long long)n_samples + qseg_len - 1) / qseg_len;
            nw = 8;
            while (nw > 1 && cnt * qseg < 256LL * nw) nw >>= 1;
            while (nw > 1 && qm_lds_bytes(c->qm_halo, c->qm_bsteps, nw, c->qm_fr) > 80 * 1024) nw >>= 1;     // two workgroups per CU
            const size_t qlds = qm_lds_bytes(c->qm_halo, c->qm_bsteps, nw, c->qm_fr);
            q.mf_nw = nw; q.nseg = (int)qseg; q.seg_len = qseg_len;
            q.fold_period = (int)c->qm_group_start[gi]; q.fold_rot = (int)cnt; q.mf_waves = (int)(gi / 3);
            grid = (unsigned)((cnt * qseg + nw - 1) / nw);
            (void)launch_chain_q15mf(c->ctx->stream, (int)(gi % 3), c->qm_fr, grid, (unsigned)nw * 64, qlds, q);
            if (int rc = launch_check("chain_q15mf_kernel")) return rc;
        }
        kname = c->qm_fr ? "chain_q15mf_kernel full-rate NCO streams" : "chain_q15mf_kernel"; block = (unsigned)nw * 64; nseg = qseg;
    }
    else     (void)launch_chain_generic(c->ctx->stream, true, grid, lds, p);
    if (int rc = launch_check("chain_kernel")) return rc;
    if (e0) { HIP_TRY(hipEventRecord(e1, c->ctx->stream)); c->events.emplace_back(e0, e1); }

    if (c->seq_bq)             // F32, ill-conditioned cascade: arm_biquad_cascade_df1_f32 in CMSIS order, in place on the audio
        if (int rc = msdr_biquad_df1_f32_process(c->seq_bq, (const float *)fout, (float *)fout, (uint32_t)n_samples)) return rc;

    if (pll_active)            // SYNCAM channels: I (in d_audio) and Q (scratch) -> PLL demodulator -> audio, before the biquad nodes
        if (int rc = msdr_syncam_q15(c->pll, c->d_mode, (const q15_t *)d_audio, c->d_pll_q, (q15_t *)d_audio, (uint32_t)n_samples)) return rc;

    if (c->anr && (c->d_anr_on || c->anr_all > 0))      // LMS notch / noise reduction (.ino:702-770), then the biquad nodes
        if (int rc = msdr_anr_q15(c->anr, c->d_anr_on, c->anr_all, (q15_t *)d_audio, (uint32_t)n_samples)) return rc;

    if (c->nnodes == 2 && !nodes_fused) {      // biquad1_dac -> biquad2_dac in one pass over the audio
        const bool slabs = (c->channels & 63u) == 0 && (n_samples & 127u) == 0 && (reinterpret_cast<uintptr_t>(d_audio) & 15) == 0;
        const int per_group = c->nodes[0]->pipe_ch ? c->nodes[0]->pipe_ch : 64;
        // one 128-sample block, the reference's cadence, one stage per node: the pipeline over sub-slabs inside the block (biquad_teensy_blk_kernel),
        // ANY channel count -- the reference's own single receiver included (1 channel: 27 -> 12 us per tick; its rows past the end idle) --
        // while its 16-channel workgroups find a CU each (4096 channels: 15.1 -> 13.2 us per tick; at 8192 the slab kernel is ahead again,
        // 16.6 vs 17.3, tools/r05_nodes_x.sh); MSDR_BIQUAD_BLK=0 / 1 at create time overrides
        const uint32_t blk_wgs = (c->channels + kTqbCh - 1) / kTqbCh;
        if (n_samples == 128 && (reinterpret_cast<uintptr_t>(d_audio) & 15) == 0 && c->nodes[0]->max_stage == 0 && c->nodes[1]->max_stage == 0 &&
            (c->blk_force >= 1 || (c->blk_force < 0 && blk_wgs <= (uint32_t)c->ctx->num_cus)))
            hipLaunchKernelGGL(biquad_teensy_blk_kernel, dim3(blk_wgs), dim3(kTqbThreads), tqb_lds_bytes(), c->ctx->stream, (short *)d_audio,
                               c->nodes[0]->d_defs, c->nodes[1]->d_defs, (int)c->channels);
        else
        if ((n_samples & 127u) == 0 && (reinterpret_cast<uintptr_t>(d_audio) & 15) == 0 && c->channels % (unsigned)per_group == 0 &&
            c->nodes[0]->max_stage == 0 && c->nodes[1]->max_stage == 0) {
            // one stage per node (the reference's configuration): the recursions alone on two waves, the input products element-wise on the others
            if (per_group == 64)
                hipLaunchKernelGGL((biquad_teensy_pipe4_kernel<2, 64>), dim3(c->channels / 64), dim3(tq4_threads(64)), tq4_lds_bytes(64), c->ctx->stream, (short *)d_audio,
                                   c->nodes[0]->d_defs, c->nodes[1]->d_defs, (int)c->channels, (long long)n_samples);
            else if (per_group == 32)
                hipLaunchKernelGGL((biquad_teensy_pipe4_kernel<2, 32>), dim3(c->channels / 32), dim3(tq4_threads(32)), tq4_lds_bytes(32), c->ctx->stream, (short *)d_audio,
                                   c->nodes[0]->d_defs, c->nodes[1]->d_defs, (int)c->channels, (long long)n_samples);
            else
                hipLaunchKernelGGL((biquad_teensy_pipe4_kernel<2, 16>), dim3(c->channels / 16), dim3(tq4_threads(16)), tq4_lds_bytes(16), c->ctx->stream, (short *)d_audio,
                                   c->nodes[0]->d_defs, c->nodes[1]->d_defs, (int)c->channels, (long long)n_samples);
        }
        else if (slabs)
            hipLaunchKernelGGL(biquad_teensy_pipe_kernel, dim3(c->channels / 64), dim3(128), 0, c->ctx->stream, (short *)d_audio,
                               c->nodes[0]->d_defs, c->nodes[1]->d_defs, (int)c->channels, (long long)n_samples);     // node per wave, slab pipeline
        else
            hipLaunchKernelGGL((biquad_teensy_kernel<2>), dim3((c->channels + 63) / 64), dim3(64), 0, c->ctx->stream, (short *)d_audio,
                               c->nodes[0]->d_defs, c->nodes[1]->d_defs, (int)c->channels, (long long)n_samples);
        if (int rc = launch_check("biquad_teensy_kernel<2>")) return rc;
    } else if (c->nnodes == 1) {
        if (int rc = msdr_biquad_q15_update(c->nodes[0], (q15_t *)d_audio, (uint32_t)n_samples)) return rc;
    }

    // rows f2 / f3 inside the fp32 chain: PLL / LMS channels are redone behind the main kernel (before the history moves on: a rebuilt
    // auxiliary chain takes over the history and table position this call started from)
    if (f32) if (int rc = chain_post_run(c, d_if, (float *)fout, n_samples)) return rc;
    if (i16_via_scratch) {
        const long long total = (long long)c->channels * (long long)n_samples;
        hipLaunchKernelGGL(f32_to_q15_kernel, dim3(grid_1d((total + 3) / 4)), dim3(256), 0, c->ctx->stream, (const float *)fout, (short *)d_audio, total);
        if (int rc = launch_check("f32_to_q15_kernel")) return rc;
    }

    if (!use_mfb && !use_qb) {            // (the block kernels write the next history themselves)
        hipLaunchKernelGGL((history_kernel<int16_t>), dim3(grid_1d((long long)c->channels * c->hist_len)), dim3(256), 0, c->ctx->stream,
                           d_if, (const int16_t *)c->d_hist[c->cur], c->d_hist[c->cur ^ 1], (long long)n_samples, (int)c->hist_len,
                           (int)c->channels);
        if (int rc = launch_check("history_kernel")) return rc;
    }
    c->cur ^= 1; c->gen++;
    c->phase = (c->phase + (long long)(n_samples % c->osc_len)) % c->osc_len;
    if (c->force_generic) {
        for (auto &o : c->osc_pending) o.elapsed += (long long)n_samples;
        if (c->osc_pending.empty() || c->osc_pending.back().elapsed >= (long long)c->hist_len)
            if (int rc = chain_leave_generic(c)) return rc;
    }

    snprintf(c->info.kernel, sizeof c->info.kernel, "%s%s", kname, !c->seq_bq ? "" : c->seq_bq->sequential ? " + biquad_df1_seq_kernel" : " + biquad_df1_kernel");
    c->info.grid = grid; c->info.block = block; c->info.lds_bytes = (uint32_t)lds_used;
    c->info.time_segments = (uint32_t)nseg; c->info.warmup = (uint32_t)p.warm; c->info.tile = (uint32_t)kTile;
    c->info.taps_padded = c->ntaps_pad;
    c->info.mfma_ksteps = use_mf ? (uint32_t)c->mf_bsteps : use_qm ? (uint32_t)c->qm_bsteps : 0u;
    return 0;
}

// ---- msdr_chain_graph_*: `ticks` consecutive block-cadence calls as ONE HIP graph ------------------------------------------------------------
// At the reference's cadence (one 128-sample block per call, Minimal-SDR.ino:518-530) a call is 7 - 10 us of GPU work behind 3 - 5 us of host
// work per launch; a graph replay enqueues `ticks` calls for the price of one.  The chain's history and cascade state alternate between two
// buffers from call to call, so a graph holds an EVEN number of calls: after a replay the host's record of which buffer is current is what it
// was, and direct calls, live updates and replays can follow one another in any order.
struct msdr_chain_graph {
    msdr_chain *c;
    hipGraph_t graph;
    hipGraphExec_t exec;
    uint32_t ticks;
    uint64_t n;
    // what the captured launches point at: a replay is refused once any of it has moved (a live update rebuilt the tables, a call of
    // another length ran in between an odd number of times, the chain was reset)
    const void *k_hist, *k_state, *k_tab, *k_tiles;
    int k_cur;
    uint64_t k_mode_gen;
};
static void chain_graph_key(const msdr_chain *c, msdr_chain_graph *g)
{
    g->k_hist = c->d_hist[c->cur]; g->k_state = c->d_bq_state; g->k_tab = c->arith == MSDR_ARITH_F32 ? (const void *)c->d_mf_tab : (const void *)c->d_qm_tab;
    g->k_tiles = c->d_btiles; g->k_cur = c->cur; g->k_mode_gen = c->mode_gen;
}

extern "C" int msdr_chain_graph_create(msdr_chain *c, uint32_t ticks, const int16_t *const *d_if, void *const *d_audio, uint64_t n_samples, msdr_chain_graph **out)
{
    if (!out) return fail(MSDR_STATUS_ARGUMENT_ERROR, "out is null");
    *out = nullptr;
    if (!c || !d_if || !d_audio) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(c->ctx)) return rc;
    if (ticks < 2 || (ticks & 1u) || ticks > 1024) return fail(MSDR_STATUS_ARGUMENT_ERROR, "a chain graph holds an even number of calls, 2 .. 1024 (the state buffers alternate from call to call)");
    for (uint32_t k = 0; k < ticks; k++) if (!d_if[k] || !d_audio[k]) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null buffer for call %u", k);
    // 1. preparation: every table / cache a call of this shape needs, built outside the capture (uploads and synchronisations are not
    //    capturable); the same pass says whether the call's launches are fixed
    const bool timing = c->timing;
    c->timing = false; c->dry_run = true;
    int rc = 0;
    for (uint32_t k = 0; k < ticks && !rc; k++) rc = msdr_chain_process(c, d_if[k], d_audio[k], n_samples);      // (alignment is per buffer)
    c->dry_run = false;
    if (rc) { c->timing = timing; return rc; }
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    // 2. capture: the launches of `ticks` calls.  The host's bookkeeping runs with them (an even number of flips: back where it was).
    const int cur0 = c->cur; const uint64_t gen0 = c->gen; const long long phase0 = c->phase;
    float *const st0 = c->d_bq_state, *const st1 = c->d_bq_state_alt;
    msdr_chain_graph *g = new (std::nothrow) msdr_chain_graph();
    if (!g) { c->timing = timing; return fail(MSDR_STATUS_OUT_OF_MEMORY, "host allocation failed"); }
    g->c = c; g->graph = nullptr; g->exec = nullptr; g->ticks = ticks; g->n = n_samples;
    chain_graph_key(c, g);
    if (hipStreamBeginCapture(c->ctx->stream, hipStreamCaptureModeRelaxed) != hipSuccess) { delete g; c->timing = timing; return fail(MSDR_STATUS_HIP_ERROR, "hipStreamBeginCapture failed"); }
    for (uint32_t k = 0; k < ticks && !rc; k++) rc = msdr_chain_process(c, d_if[k], d_audio[k], n_samples);
    const std::string why = rc ? g_err : std::string();
    const hipError_t ee = hipStreamEndCapture(c->ctx->stream, &g->graph);
    c->timing = timing;
    c->cur = cur0; c->gen = gen0; c->phase = phase0; c->d_bq_state = st0; c->d_bq_state_alt = st1;        // (nothing ran: the stream is where it was)
    if (rc || ee != hipSuccess || !g->graph) {
        if (g->graph) hipGraphDestroy(g->graph);
        delete g;
        (void)hipGetLastError();
        return rc ? fail(rc, "capture of a chain call failed: %s", why.c_str()) : fail(MSDR_STATUS_HIP_ERROR, "hipStreamEndCapture failed");
    }
    if (hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) {
        hipGraphDestroy(g->graph); delete g;
        return fail(MSDR_STATUS_HIP_ERROR, "hipGraphInstantiate failed");
    }
    *out = g;
    return 0;
}

extern "C" int msdr_chain_graph_launch(msdr_chain_graph *g)
{
    if (!g || !g->c) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null graph");
    msdr_chain *c = g->c;
    if (int rc = bind(c->ctx)) return rc;
    msdr_chain_graph now;
    chain_graph_key(c, &now);
    if (now.k_hist != g->k_hist || now.k_state != g->k_state || now.k_tab != g->k_tab || now.k_tiles != g->k_tiles || now.k_cur != g->k_cur || now.k_mode_gen != g->k_mode_gen)
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "the chain has changed since this graph was made (a live update, a reset, or an odd number of direct calls in between): make the graph again");
    HIP_TRY(hipGraphLaunch(g->exec, c->ctx->stream));
    c->gen += g->ticks;                                     // (an even number of calls: buffers, table position and caches stay as they are)
    return 0;
}

extern "C" int msdr_chain_graph_destroy(msdr_chain_graph *g)
{
    if (!g) return 0;
    if (g->c) (void)bind(g->c->ctx);
    if (g->exec) hipGraphExecDestroy(g->exec);
    if (g->graph) hipGraphDestroy(g->graph);
    delete g;
    return 0;
}

extern "C" int msdr_chain_reset(msdr_chain *c)
{
    if (!c) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null chain");
    if (int rc = bind(c->ctx)) return rc;
    size_t hb = (size_t)c->channels * c->hist_len * sizeof(int16_t);
    HIP_TRY(hipMemsetAsync(c->d_hist[0], 0, hb, c->ctx->stream));
    HIP_TRY(hipMemsetAsync(c->d_hist[1], 0, hb, c->ctx->stream));
    if (c->d_bq_state_alt) HIP_TRY(hipMemsetAsync(c->d_bq_state_alt, 0, (size_t)c->channels * kBqStateFloats * sizeof(float), c->ctx->stream));
    if (c->d_bq_state) HIP_TRY(hipMemsetAsync(c->d_bq_state, 0, (size_t)c->channels * kBqStateFloats * sizeof(float), c->ctx->stream));
    // (a full clear for tests and stream restarts; the reference's init_FIR() is msdr_chain_init_fir(): FIR state only.  The
    //  Teensy biquad NODES keep their history here too, as the reference never clears it: filter_biquad.cpp:95-97.)
    if (c->pll) if (int rc = msdr_syncam_reset(c->pll)) return rc;
    if (c->anr) if (int rc = msdr_anr_reset(c->anr)) return rc;
    if (c->seq_bq) if (int rc = msdr_biquad_df1_f32_reset(c->seq_bq)) return rc;
    if (c->aux) { HIP_TRY(hipStreamSynchronize(c->ctx->stream)); chain_post_free(c); c->h_post_ch.clear(); c->post_mode_gen = 0; }      // rebuilt from the cleared state
    if (c->d_post_pll_state) {
        HIP_TRY(hipMemsetAsync(c->d_post_pll_state, 0, (size_t)c->channels * 4 * sizeof(float), c->ctx->stream));
        std::vector<float> h((size_t)c->channels * kAnrStateFloats, 0.0f);
        for (uint32_t ch = 0; ch < c->channels; ch++) { h[(size_t)ch * kAnrStateFloats] = 120.0f; h[(size_t)ch * kAnrStateFloats + 1] = 0.001f; }
        HIP_TRY(hipMemcpy(c->d_post_anr_state, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (!c->osc_pending.empty() || c->force_generic) {          // no sample of an earlier oscillator table is left in a cleared history
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        for (auto &o : c->osc_pending) hipFree(o.d_tab);
        c->osc_pending.clear(); c->force_generic = false;
    }
    c->phase = 0; c->gen++;
    return 0;
}

// A mode whose matrix-core table folds the cascade's numerator C(z) into the FIR (msdr_chain_mfma.hiph): it neither maintains nor
// reads the numerator history d[n-1-k] of the carried state.
static bool chain_mode_folded(const msdr_chain *c, int m)
{
    return c->arith == MSDR_ARITH_F32 && c->mf_ok && !c->force_generic && c->nstages > 0 && (m == MSDR_MODE_LSB || m == MSDR_MODE_USB);
}
// d[-1-k], k < 8, of an SSB mode from a channel's raw IF history (what the folded FIR implies the cascade has seen)
static void chain_ssb_history(const msdr_chain *c, const int16_t *hist, int m, int ts, double *D)
{
    const int N = (int)c->ntaps, HL = (int)c->hist_len, OL = (int)c->osc_len;
    const float *hi = c->h_coef_i[ts].data(), *hq = c->h_coef_q[ts].data();
    const double sign = (m == MSDR_MODE_LSB) ? -1.0 : 1.0;
    for (int k = 0; k < 8; k++) {
        double acc = 0.0;
        for (int dl = 0; dl < N; dl++) {
            const int t = -1 - k - dl, hx = HL + t;
            if (hx < 0) break;
            const int ph = (int)((((long long)c->phase + t) % OL + OL) % OL);
            const double x = (double)hist[hx] * (double)c->in_scale;
            acc += (double)hi[N - 1 - dl] * (x * c->h_osc[2 * ph]) + sign * (double)hq[N - 1 - dl] * (x * c->h_osc[2 * ph + 1]);
        }
        D[k] = acc;
    }
}
// the numerator history as the cascade has really seen it: the cache (retuned again before any sample was processed), the folded
// mode's own FIR over the raw history, or the state record
static void chain_true_history(msdr_chain *c, uint32_t channel, const int16_t *hist, const float *st, double *D)
{
    if (c->dh_gen[channel] == c->gen) memcpy(D, c->dh_cache[channel].v, 8 * sizeof(double));
    else if (chain_mode_folded(c, c->h_mode[channel])) chain_ssb_history(c, hist, c->h_mode[channel], c->h_tapset[channel], D);
    else for (int k = 0; k < 8; k++) D[k] = st[k];
    memcpy(c->dh_cache[channel].v, D, 8 * sizeof(double)); c->dh_gen[channel] = c->gen;
}
// first-sample corrections sum_k c[j+1+k] (D[k] - Dn[k]) of a folded mode whose FIR implies the history Dn
static void chain_folded_correction(const msdr_chain *c, const double *D, const double *Dn, float *out)
{
    const int H2 = 2 * (int)c->nstages;
    for (int j = 0; j < 8; j++) out[j] = 0.0f;
    for (int j = 0; j < H2; j++) {
        double a = 0.0;
        for (int k = 0; j + 1 + k <= H2; k++) a += c->h_cnum[j + 1 + k] * (D[k] - Dn[k]);
        out[j] = (float)a;
    }
}

// init_FIR() (Minimal-SDR.ino:901-930) re-runs arm_fir_init_q15 and nothing else: the FIR state is zeroed; the biquad nodes, the
// SYNCAM PLL statics (fil_out, omega2, phzerror) and the LMS weights / delay line persist across a retune, and so does the mixer's
// position in its table.
extern "C" int msdr_chain_init_fir(msdr_chain *c)
{
    if (!c) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null chain");
    if (int rc = bind(c->ctx)) return rc;
    bool any_folded = false;
    for (uint32_t ch = 0; ch < c->channels && !any_folded; ch++) any_folded = chain_mode_folded(c, c->h_mode[ch]);
    if (any_folded && c->d_bq_state) {
        // a folded channel's numerator history lives in the raw IF history that is about to be cleared: hand it over as
        // first-sample corrections, exactly as a retune does (msdr_chain_set_mode)
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        std::vector<int16_t> hist((size_t)c->channels * c->hist_len);
        std::vector<float> st((size_t)c->channels * kBqStateFloats);
        HIP_TRY(hipMemcpy(hist.data(), c->d_hist[c->cur], hist.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(st.data(), c->d_bq_state, st.size() * sizeof(float), hipMemcpyDeviceToHost));
        const double zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t ch = 0; ch < c->channels; ch++) {
            if (!chain_mode_folded(c, c->h_mode[ch])) continue;
            double D[8];
            chain_true_history(c, ch, hist.data() + (size_t)ch * c->hist_len, st.data() + (size_t)ch * kBqStateFloats, D);
            chain_folded_correction(c, D, zero, st.data() + (size_t)ch * kBqStateFloats);
        }
        HIP_TRY(hipMemcpy(c->d_bq_state, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (c->aux) if (int rc = msdr_chain_init_fir(c->aux)) return rc;             // the PLL / LMS channels' filters with it
    const size_t hb = (size_t)c->channels * c->hist_len * sizeof(int16_t);
    HIP_TRY(hipMemsetAsync(c->d_hist[0], 0, hb, c->ctx->stream));
    HIP_TRY(hipMemsetAsync(c->d_hist[1], 0, hb, c->ctx->stream));
    return 0;
}

extern "C" int msdr_chain_set_mode(msdr_chain *c, uint32_t channel, int32_t mode, int32_t tapset)
{
    if (!c) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null chain");
    if (int rc = bind(c->ctx)) return rc;
    if (channel >= c->channels || mode < MSDR_MODE_SYNCAM || mode > MSDR_MODE_CW || tapset < 0 || (uint32_t)tapset >= c->tapsets)
        return fail(MSDR_STATUS_ARGUMENT_ERROR, "bad channel/mode/tapset");
    // Keep the stream exact across a retune when the old or the new mode's table folds the numerator into the FIR: D = the
    // history as the cascade has really seen it; an unfolded new mode gets D; a folded one gets the correction
    // sum_k c[j+1+k] (D[k] - D'[k]) to its first samples j, D' = the history its own FIR implies.
    {
        const int old_mode = c->h_mode[channel], old_ts = c->h_tapset[channel];
        if ((chain_mode_folded(c, old_mode) || chain_mode_folded(c, mode)) && (old_mode != mode || old_ts != tapset)) {
            HIP_TRY(hipStreamSynchronize(c->ctx->stream));
            std::vector<int16_t> hist(c->hist_len);
            float st[kBqStateFloats];
            HIP_TRY(hipMemcpy(hist.data(), c->d_hist[c->cur] + (size_t)channel * c->hist_len, hist.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(st, c->d_bq_state + (size_t)channel * kBqStateFloats, sizeof st, hipMemcpyDeviceToHost));
            double D[8];
            chain_true_history(c, channel, hist.data(), st, D);
            float out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (chain_mode_folded(c, mode)) {
                double Dn[8];
                chain_ssb_history(c, hist.data(), mode, tapset, Dn);
                chain_folded_correction(c, D, Dn, out);
            } else for (int k = 0; k < 8; k++) out[k] = (float)D[k];
            HIP_TRY(hipMemcpy(c->d_bq_state + (size_t)channel * kBqStateFloats, out, sizeof out, hipMemcpyHostToDevice));
        }
    }
    c->h_mode[channel] = mode; c->h_tapset[channel] = tapset; c->mode_gen++;
    if (c->d_fset) {
        const int fs = tapset * 3 + (mode == MSDR_MODE_LSB ? 0 : mode == MSDR_MODE_USB ? 1 : 2);
        HIP_TRY(hipMemcpyAsync(c->d_fset + channel, &fs, sizeof(int), hipMemcpyHostToDevice, c->ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(c->d_mode + channel, &mode, sizeof(int), hipMemcpyHostToDevice, c->ctx->stream));
    HIP_TRY(hipMemcpyAsync(c->d_tapset + channel, &tapset, sizeof(int), hipMemcpyHostToDevice, c->ctx->stream));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    return 0;
}

// ---- live updates (include/msdr.h): the chain's tables are rebuilt by msdr_chain_create from the edited configuration, every piece of
// STATE moves from the running chain into the rebuilt one, and the two structs trade places, so the caller's handle now names the new
// tables over the old state.  The stream is drained first: the old tables are freed with the husk.
// raw IF history carried into a rebuilt chain whose history length differs (a cascade that moves between the chain kernel and the
// CMSIS-order pass changes the matrix-core window's halo): the newest min(hl_src, hl_dst) samples, older entries zero; both "oldest first"
__global__ void hist_resize_kernel(const int16_t *__restrict__ src, int16_t *__restrict__ dst, int channels, int hl_src, int hl_dst)
{
    const long long total = (long long)channels * hl_dst;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i / hl_dst), k = (int)(i - (long long)ch * hl_dst), back = hl_dst - k;       // back = 1: the newest sample
        dst[i] = (back <= hl_src) ? src[(long long)ch * hl_src + (hl_src - back)] : (int16_t)0;
    }
}

// extra_floor_d / extra_floor_sig: what the caller KNOWS it will hand the rebuilt chain (numerator history / section states, in units of
// in_scale): msdr_chain_set_biquad_coeffs converts the running cascade's state into the new cascade's basis, and neither chain's own bounds
// say how large that comes out (tests/debug/fuzz_live.py seed 5342 case 38296: a resonant cascade that ran in CMSIS order behind the kernel,
// rewritten to one that runs inside it -- the old chain had no bounds at all, the new tables were scaled for the new cascade's gains, and
// the converted state left fp16's range: 23 x the output on the next call, decaying over the two after it)
static int chain_rebuild(msdr_chain *c, const ChainCfgStore &edited, void **steal_osc = nullptr, double extra_floor_d = 0.0, double extra_floor_sig = 0.0)
{
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    msdr_chain_config cfg;
    edited.view(&cfg, c->h_mode, c->h_tapset);
    msdr_chain *n = nullptr;
    g_chain_floor_d = c->own_d_bound; g_chain_floor_sig = c->own_sig_bound;
    if (c->floor_gen == c->gen) { g_chain_floor_d = std::max(g_chain_floor_d, c->floor_d); g_chain_floor_sig = std::max(g_chain_floor_sig, c->floor_sig); }
    g_chain_floor_d = std::max(g_chain_floor_d, extra_floor_d); g_chain_floor_sig = std::max(g_chain_floor_sig, extra_floor_sig);
    const double used_d = g_chain_floor_d, used_sig = g_chain_floor_sig;
    const int rc = msdr_chain_create(c->ctx, &cfg, &n);
    g_chain_floor_d = 0.0; g_chain_floor_sig = 0.0;
    if (rc) return rc;
    if (n->channels != c->channels || n->osc_len != c->osc_len || n->ntaps != c->ntaps) {
        chain_free(n);
        return fail(MSDR_STATUS_SIZE_MISMATCH, "internal: a live update changed the chain's geometry");
    }
    // FIR history and table position
    if (n->hist_len == c->hist_len) { std::swap(n->d_hist[0], c->d_hist[0]); std::swap(n->d_hist[1], c->d_hist[1]); n->cur = c->cur; }
    else {
        hipLaunchKernelGGL(hist_resize_kernel, dim3(grid_1d((long long)n->channels * n->hist_len)), dim3(256), 0, c->ctx->stream,
                           (const int16_t *)c->d_hist[c->cur], n->d_hist[n->cur], (int)n->channels, (int)c->hist_len, (int)n->hist_len);
        if (int rc2 = launch_check("hist_resize_kernel")) { chain_free(n); return rc2; }
        if (hipStreamSynchronize(c->ctx->stream) != hipSuccess) { chain_free(n); return fail(MSDR_STATUS_HIP_ERROR, "hist_resize_kernel failed"); }
    }
    n->phase = c->phase;
    // fp32 cascade state (the callers that change the cascade itself rewrite it afterwards)
    if (n->d_bq_state && c->d_bq_state) std::swap(n->d_bq_state, c->d_bq_state);
    if (n->seq_bq && c->seq_bq) std::swap(n->seq_bq, c->seq_bq);
    // Teensy biquad nodes (coefficients and history both live in their records), PLL, LMS filter
    for (int k = 0; k < 2; k++) if (n->nodes[k] && c->nodes[k]) std::swap(n->nodes[k], c->nodes[k]);
    std::swap(n->pll, c->pll); std::swap(n->d_pll_q, c->d_pll_q); std::swap(n->pll_q_cap, c->pll_q_cap);
    std::swap(n->anr, c->anr); std::swap(n->d_anr_on, c->d_anr_on); n->anr_all = c->anr_all;
    // fp32 post channels: per-channel PLL / LMS state and the post cascade persist; the auxiliary chain is rebuilt from the new
    // configuration at the next call and takes the history over from d_hist (chain_post_build_steps)
    n->h_anr = c->h_anr; n->anr_gen = c->anr_gen; n->f32_pll = c->f32_pll;
    std::swap(n->d_post_pll_state, c->d_post_pll_state); std::swap(n->d_post_anr_state, c->d_post_anr_state);
    std::swap(n->post_bq, c->post_bq); n->h_post_ch = c->h_post_ch; c->h_post_ch.clear();
    n->post_mode_gen = 0; n->post_anr_gen = 0;
    // bookkeeping of the stream
    n->gen = c->gen; n->dh_cache = c->dh_cache; n->dh_gen = c->dh_gen;
    n->floor_d = used_d; n->floor_sig = used_sig; n->floor_gen = c->gen;
    n->timing = c->timing; n->events.swap(c->events); n->timed_ms = c->timed_ms; n->timed_launches = c->timed_launches;
    n->info = c->info;
    n->osc_pending.swap(c->osc_pending); n->force_generic = c->force_generic;
    std::swap(*c, *n);
    if (steal_osc) { *steal_osc = n->d_osc; n->d_osc = nullptr; }
    chain_free(n);
    return 0;
}

// The history has turned over since the last oscillator change: the fast kernels take the stream back.  A mode whose table folds the
// cascade's numerator into the FIR takes its state record the other way round (first-sample corrections instead of the history).
static int chain_leave_generic(msdr_chain *c)
{
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    for (auto &o : c->osc_pending) hipFree(o.d_tab);
    c->osc_pending.clear();
    if (!c->force_generic) return 0;
    c->force_generic = false;
    bool any = false;
    for (uint32_t ch = 0; ch < c->channels && !any; ch++) any = chain_mode_folded(c, c->h_mode[ch]);
    if (!any || !c->d_bq_state) return 0;
    std::vector<int16_t> hist((size_t)c->channels * c->hist_len);
    std::vector<float> st((size_t)c->channels * kBqStateFloats);
    HIP_TRY(hipMemcpy(hist.data(), c->d_hist[c->cur], hist.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st.data(), c->d_bq_state, st.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (uint32_t ch = 0; ch < c->channels; ch++) {
        if (!chain_mode_folded(c, c->h_mode[ch])) continue;
        float *r = st.data() + (size_t)ch * kBqStateFloats;
        double D[8], Dn[8];
        for (int k = 0; k < 8; k++) D[k] = r[k];
        chain_ssb_history(c, hist.data() + (size_t)ch * c->hist_len, c->h_mode[ch], c->h_tapset[ch], Dn);
        chain_folded_correction(c, D, Dn, r);
        memcpy(c->dh_cache[ch].v, D, sizeof D); c->dh_gen[ch] = c->gen;
    }
    HIP_TRY(hipMemcpy(c->d_bq_state, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// A rebuild that changes what a numerator-folded SSB table implies about the cascade's input history (new taps, new oscillator):
// the true history D (old tables) is handed over as first-sample corrections against the history Dn the new tables imply, exactly as a
// retune does (msdr_chain_set_mode).  affected(ch): the channel's FIR or oscillator changes.
static int chain_rebuild_keep_folded(msdr_chain *c, const ChainCfgStore &edited, const std::function<bool(uint32_t)> &affected, bool osc_changes = false)
{
    std::vector<uint32_t> fix;
    if (c->arith == MSDR_ARITH_F32 && c->d_bq_state)
        for (uint32_t ch = 0; ch < c->channels; ch++) if (chain_mode_folded(c, c->h_mode[ch]) && affected(ch)) fix.push_back(ch);
    std::vector<int16_t> hist;
    std::vector<float> st;
    std::vector<double> Dall;
    if (!fix.empty()) {
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        hist.resize((size_t)c->channels * c->hist_len); st.resize((size_t)c->channels * kBqStateFloats); Dall.resize((size_t)c->channels * 8);
        HIP_TRY(hipMemcpy(hist.data(), c->d_hist[c->cur], hist.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(st.data(), c->d_bq_state, st.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (uint32_t ch : fix) chain_true_history(c, ch, hist.data() + (size_t)ch * c->hist_len, st.data() + (size_t)ch * kBqStateFloats, Dall.data() + (size_t)ch * 8);
    }
    void *old_osc = nullptr;
    if (int rc = chain_rebuild(c, edited, osc_changes ? &old_osc : nullptr)) return rc;
    if (osc_changes) {
        // the samples in the FIR history were mixed with the old tables when they arrived: keep those for as long as the history holds
        // such samples.  A table that was replaced before any sample arrived under it (two changes with no call in between) mixed nothing:
        // it is not a generation (tests/debug/fuzz_live.py seed 5312 case 23466: five changes inside one history length, two of them such,
        // used up round 4's four slots and the oldest REAL generation was dropped: 0.26 of the output).  At most kOscHistMax generations; one
        // more inside a single history length drops the oldest (include/msdr.h says so).
        if (!c->osc_pending.empty() && c->osc_pending.back().elapsed == 0) hipFree(old_osc);
        else {
            if (c->osc_pending.size() >= (size_t)kOscHistMax) { hipFree(c->osc_pending.front().d_tab); c->osc_pending.erase(c->osc_pending.begin()); }
            c->osc_pending.push_back(msdr_chain::OscPending{old_osc, 0});
        }
        c->force_generic = true;
    }
    if (!fix.empty() && c->d_bq_state) {
        if (hist.size() != (size_t)c->channels * c->hist_len) {
            hist.resize((size_t)c->channels * c->hist_len);
            HIP_TRY(hipMemcpy(hist.data(), c->d_hist[c->cur], hist.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
        }
        for (uint32_t ch : fix) {
            float *r = st.data() + (size_t)ch * kBqStateFloats;
            const double *D = Dall.data() + (size_t)ch * 8;
            if (chain_mode_folded(c, c->h_mode[ch])) {
                double Dn[8];
                chain_ssb_history(c, hist.data() + (size_t)ch * c->hist_len, c->h_mode[ch], c->h_tapset[ch], Dn);
                chain_folded_correction(c, D, Dn, r);
            } else for (int k = 0; k < 8; k++) r[k] = (float)D[k];       // (the rebuilt chain does not fold this mode any more)
            HIP_TRY(hipMemcpy(c->d_bq_state + (size_t)ch * kBqStateFloats, r, 8 * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    return 0;
}

extern "C" int msdr_chain_set_taps(msdr_chain *c, uint32_t tapset, const void *coeffs_i, const void *coeffs_q)
{
    if (!c || !coeffs_i || !coeffs_q) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(c->ctx)) return rc;
    if (tapset >= c->tapsets) return fail(MSDR_STATUS_ARGUMENT_ERROR, "tap set %u of %u", tapset, c->tapsets);
    ChainCfgStore ed = c->store;
    const size_t bytes = ed.ci[tapset].size();
    ed.ci[tapset].assign((const char *)coeffs_i, (const char *)coeffs_i + bytes);
    ed.cq[tapset].assign((const char *)coeffs_q, (const char *)coeffs_q + bytes);
    return chain_rebuild_keep_folded(c, ed, [&](uint32_t ch) { return (uint32_t)c->h_tapset[ch] == tapset; });
}

extern "C" int msdr_chain_set_osc(msdr_chain *c, const void *osc_i, const void *osc_q)
{
    if (!c || !osc_i || !osc_q) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(c->ctx)) return rc;
    if (c->mixer != MSDR_MIXER_NCO) return fail(MSDR_STATUS_ARGUMENT_ERROR, "this chain has the Fs/4 mixer: no oscillator tables (Minimal-SDR.ino:546-558)");
    ChainCfgStore ed = c->store;
    const size_t bytes = ed.osc_i.size();
    ed.osc_i.assign((const char *)osc_i, (const char *)osc_i + bytes);
    ed.osc_q.assign((const char *)osc_q, (const char *)osc_q + bytes);
    return chain_rebuild_keep_folded(c, ed, [](uint32_t) { return true; }, true);
}

extern "C" int msdr_chain_set_node_coefficients(msdr_chain *c, uint32_t node, uint32_t stage, const int32_t coef[5])
{
    if (!c || !coef) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(c->ctx)) return rc;
    if (c->arith != MSDR_ARITH_Q15) return fail(MSDR_STATUS_ARGUMENT_ERROR, "AudioFilterBiquad nodes belong to Q15 chains (fp32: msdr_chain_set_biquad_coeffs)");
    if (node >= c->nnodes || !c->nodes[node]) return fail(MSDR_STATUS_ARGUMENT_ERROR, "node %u of %u", node, c->nnodes);
    if (stage >= 4) return 0;                                                 // filter_biquad.cpp:86
    if (int rc = msdr_biquad_q15_set_coefficients(c->nodes[node], stage, coef)) return rc;
    std::vector<int32_t> &h = c->store.node[node];                              // keep the stored configuration in step
    if (h.size() < 5 * (size_t)(stage + 1)) h.resize(5 * (size_t)(stage + 1), 0);
    memcpy(h.data() + 5 * stage, coef, 5 * sizeof(int32_t));
    c->store.cfg.node_stages[node] = std::max<uint32_t>(c->store.cfg.node_stages[node], stage + 1);
    return 0;
}

// arm_biquad_cascade_df1_f32's pCoeffs rewritten under the running stream: state kept in CMSIS terms (msdr_cascade_state.h)
extern "C" int msdr_chain_set_biquad_coeffs(msdr_chain *c, const float32_t *coeffs)
{
    if (!c || !coeffs) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    if (int rc = bind(c->ctx)) return rc;
    if (c->arith != MSDR_ARITH_F32) return fail(MSDR_STATUS_ARGUMENT_ERROR, "the fp32 cascade belongs to F32 chains (Q15: msdr_chain_set_node_coefficients)");
    const int S = (int)c->h_bq_stages, ny = 2 * S + 2;
    if (S == 0) return fail(MSDR_STATUS_ARGUMENT_ERROR, "this chain was created without a biquad cascade (numStages is fixed at creation, as in CMSIS)");
    for (int k = 0; k < 5 * S; k++) if (!std::isfinite(coeffs[k])) return fail(MSDR_STATUS_ARGUMENT_ERROR, "coefficient %d is not finite", k);
    cstate::Bridge br;
    br.build(c->h_bq.data(), coeffs, S);
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    // ---- every channel's state in CMSIS terms, and the cascade's true input history ----
    std::vector<double> Y((size_t)c->channels * ny, 0.0), D((size_t)c->channels * 8, 0.0);
    std::vector<int16_t> hist((size_t)c->channels * c->hist_len);
    HIP_TRY(hipMemcpy(hist.data(), c->d_hist[c->cur], hist.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
    if (c->seq_bq) {
        // (a stage object that runs block-parallel keeps its state in the coefficient-dependent basis too: the same refusal applies --
        //  build_to() leaves its matrices zero on failure, and reading through them would silently zero the filter's state)
        if (!c->seq_bq->sequential && !br.ok_to) return fail(MSDR_STATUS_ARGUMENT_ERROR, "the running cascade's block-parallel state has no unique CMSIS state (a numerator shares a root with an earlier denominator)");
        if (int rc = biquad_df1_read_cmsis(c->seq_bq, br, Y, D)) return rc;
    } else {
        if (!br.ok_to) return fail(MSDR_STATUS_ARGUMENT_ERROR, "the running cascade's block-parallel state has no unique CMSIS state (a numerator shares a root with an earlier denominator)");
        std::vector<float> st((size_t)c->channels * kBqStateFloats);
        HIP_TRY(hipMemcpy(st.data(), c->d_bq_state, st.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (uint32_t ch = 0; ch < c->channels; ch++) {
            double *d = D.data() + (size_t)ch * 8;
            chain_true_history(c, ch, hist.data() + (size_t)ch * c->hist_len, st.data() + (size_t)ch * kBqStateFloats, d);
            br.lib_to_cmsis(d, st.data() + (size_t)ch * kBqStateFloats, Y.data() + (size_t)ch * ny);
        }
    }
    // ---- the rebuilt chain ----
    const bool new_seq = cascade_needs_cmsis_order(coeffs, S);
    if (!new_seq && !br.ok_from) return fail(MSDR_STATUS_ARGUMENT_ERROR, "no block-parallel state reproduces the CMSIS state under the new coefficients");
    ChainCfgStore ed = c->store;
    ed.bq.assign(coeffs, coeffs + 5 * S);
    // what the rebuilt chain will be handed, measured: the section states in the new cascade's basis and the numerator history, over all channels
    double floor_d = 0.0, floor_sig = 0.0;
    if (!new_seq) {
        float w[16];
        for (uint32_t ch = 0; ch < c->channels; ch++) {
            br.cmsis_to_lib(Y.data() + (size_t)ch * ny, D.data() + (size_t)ch * 8, w);
            for (int k = 0; k < 2 * S; k++) {
                if (std::isfinite(w[k])) floor_sig = std::max(floor_sig, (double)std::fabs(w[k]));
                floor_d = std::max(floor_d, std::fabs(D[(size_t)ch * 8 + k]));
            }
        }
        const double inv = 1.5 / std::max((double)c->in_scale, 1e-300);      // (the bounds are kept in units of in_scale; half again for the first block's growth)
        floor_d *= inv; floor_sig *= inv;
    }
    msdr_biquad_df1_f32 *old_seq = c->seq_bq;          // (chain_rebuild would swap the old object in: the new one must stay)
    c->seq_bq = nullptr;
    const int rrc = chain_rebuild(c, ed, nullptr, floor_d, floor_sig);
    if (rrc) { c->seq_bq = old_seq; return rrc; }
    if (old_seq) msdr_biquad_df1_f32_destroy(old_seq);
    if (hist.size() != (size_t)c->channels * c->hist_len) {      // (the history length follows the cascade's place: read the carried-over copy)
        hist.resize((size_t)c->channels * c->hist_len);
        HIP_TRY(hipMemcpy(hist.data(), c->d_hist[c->cur], hist.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
    }
    if (c->seq_bq) {
        if (int rc = biquad_df1_write_cmsis(c->seq_bq, br, Y, D)) return rc;
    } else {
        std::vector<float> st((size_t)c->channels * kBqStateFloats, 0.0f);
        for (uint32_t ch = 0; ch < c->channels; ch++) {
            float *r = st.data() + (size_t)ch * kBqStateFloats;
            const double *d = D.data() + (size_t)ch * 8;
            br.cmsis_to_lib(Y.data() + (size_t)ch * ny, d, r + 8);
            if (chain_mode_folded(c, c->h_mode[ch])) {
                double Dn[8];
                chain_ssb_history(c, hist.data() + (size_t)ch * c->hist_len, c->h_mode[ch], c->h_tapset[ch], Dn);
                chain_folded_correction(c, d, Dn, r);
            } else for (int k = 0; k < 2 * S; k++) r[k] = (float)d[k];
            memcpy(c->dh_cache[ch].v, d, 8 * sizeof(double)); c->dh_gen[ch] = c->gen;
        }
        HIP_TRY(hipMemcpy(c->d_bq_state, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // the post channels' own cascade (rows f2 / f3 inside the fp32 chain)
    if (c->post_bq) if (int rc = msdr_biquad_df1_f32_set_coeffs(c->post_bq, coeffs)) return rc;
    return 0;
}

extern "C" int msdr_chain_set_anr(msdr_chain *c, const int32_t *anr_on, int32_t anr_on_all)
{
    if (!c) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null chain");
    if (int rc = bind(c->ctx)) return rc;
    if (c->arith != MSDR_ARITH_Q15) {
        // fp32 chain: the float flavour of the filter, between demodulator and cascade (chain_post_run)
        if (c->seq_bq && !c->h_bq_stages) return fail(MSDR_STATUS_ARGUMENT_ERROR, "internal: cascade coefficients not kept");
        c->h_anr.assign(c->channels, anr_on ? 0 : (int)anr_on_all);
        if (anr_on) for (uint32_t ch = 0; ch < c->channels; ch++) c->h_anr[ch] = (int)anr_on[ch];
        for (int &v : c->h_anr) if (v < 0) v = 0;
        c->anr_gen++;
        return 0;
    }
    if (!c->anr) if (int rc = msdr_anr_create(c->ctx, c->channels, &c->anr)) return rc;
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    hipFree(c->d_anr_on); c->d_anr_on = nullptr;
    c->anr_all = anr_on_all;
    if (anr_on) {
        std::vector<int> h(anr_on, anr_on + c->channels);
        if (int rc = upload(c->ctx, h, &c->d_anr_on)) return rc;
    }
    return 0;
}

extern "C" int msdr_chain_destroy(msdr_chain *c)
{
    if (!c) return 0;
    if (int rc = bind(c->ctx)) return rc;
    (void)hipStreamSynchronize(c->ctx->stream);
    chain_free(c);
    return 0;
}

extern "C" int msdr_chain_get_info(msdr_chain *c, msdr_chain_info *info)
{
    if (!c || !info) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null argument");
    *info = c->info;
    return 0;
}

extern "C" int msdr_chain_enable_timing(msdr_chain *c, int on)
{
    if (!c) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null chain");
    c->timing = on != 0;
    return 0;
}

extern "C" int msdr_chain_get_kernel_time(msdr_chain *c, double *total_ms, uint64_t *launches, int reset)
{
    if (!c) return fail(MSDR_STATUS_ARGUMENT_ERROR, "null chain");
    if (int rc = bind(c->ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    for (auto &e : c->events) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
        c->timed_ms += ms; c->timed_launches++;
        hipEventDestroy(e.first); hipEventDestroy(e.second);
    }
    c->events.clear();
    if (total_ms) *total_ms = c->timed_ms;
    if (launches) *launches = c->timed_launches;
    if (reset) { c->timed_ms = 0; c->timed_launches = 0; }
    return 0;
}
