// tools/probes/fir_stream_probe.hip -- third step of the round-3 stream study: the memory traffic of fir_f32tr_kernel alone (no
// arithmetic), at the kernel's own residency (8 waves per CU: two 256-thread workgroups held apart by an 80 KB LDS allocation), for
// the unit orders a persistent wave can take.  copy_order_probe.hip showed that the order of the chip's accesses decides (one compact
// front sweeping memory: 0.77-0.82 of 8 TB/s; the same accesses scrambled: 0.63); here the waves are long-lived (the taps stay in
// registers), so the front has to be kept compact by the unit -> wave map:
//   ORDER 0: today: wave = (channel, segment), walks its segment tile by tile (front = one tile in each of 2048 far-apart streams)
//   ORDER 1: units of T consecutive tiles, dealt to all 2048 waves round-robin (unit u of round r = r 2048 + wave): one front
//   ORDER 2: the same inside each eighth of the buffer for the workgroups with equal blockIdx % 8 (one XCD under round-robin
//            placement: a halo re-read finds its line in that XCD's L2): eight fronts
// A unit's first tile also fetches the 1 KB of halo in front of it (256 samples: the 256-tap filter), as the kernel would have to.
// A measurement aid, not product code.   hipcc --offload-arch=gfx950 -O3 -o fir_stream_probe fir_stream_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ f32x4 ld16(const f32x4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st16(f32x4 *p, f32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// ntiles = 4 KB tiles in the buffer (a multiple of 8 T waves); waves = gridDim.x * 4
template <int ORDER, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void fir_stream(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, long long ntiles, int T, float *__restrict__ sink)
{
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long waves = (long long)gridDim.x * 4;
    long long wave, nw, base, span;            // this wave's index among the nw waves that share the tile range [base, base + span)
    if (ORDER == 2) { wave = (long long)(blockIdx.x >> 3) * 4 + wv; nw = waves / 8; span = ntiles / 8; base = (long long)(blockIdx.x & 7) * span; }
    else { wave = (long long)blockIdx.x * 4 + wv; nw = waves; span = ntiles; base = 0; }
    const long long units = span / T;          // units of T tiles in the range
    // tile index of step s of this wave, and whether it opens a unit
    auto tile_of = [&](long long s) -> long long {
        if (ORDER == 0) return base + wave * (span / nw) + s;                        // one long unit per wave
        const long long r = s / T, k = s - r * T;
        return base + (r * nw + wave) * T + k;
    };
    const long long steps = ORDER == 0 ? span / nw : (units / nw) * T;
    f32x4 a[4], b[4], ha, hb;
    float acc = 0.0f;
    auto load = [&](f32x4 (&v)[4], f32x4 &h, long long s) {
        const long long t = tile_of(s);
        const bool opens = ORDER == 0 ? (s == 0) : (s % T == 0);
        if (opens && t > 0) h = ld16<NTL>(x + t * 256 - 64 + lane);                   // the 1 KB in front of the tile
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = ld16<NTL>(x + t * 256 + lane + 64 * j);
    };
    auto store = [&](const f32x4 (&v)[4], const f32x4 &h, long long s) {
        const long long t = tile_of(s);
        acc += h[0];
#pragma unroll
        for (int j = 0; j < 4; j++) st16<NTS>(y + t * 256 + lane + 64 * j, v[j]);
    };
    ha = hb = (f32x4)(0.0f);
    if (steps <= 0) return;
    load(a, ha, 0);
    for (long long s = 0; s < steps; s += 2) {
        if (s + 1 < steps) load(b, hb, s + 1);
        store(a, ha, s);
        if (s + 2 < steps) load(a, ha, s + 2);
        if (s + 1 < steps) store(b, hb, s + 1);
    }
    if (acc == 123.456f) sink[0] = acc + smem[0];
}


// ORDER 3 (dynamic): persistent waves, the WORKGROUP draws its next four consecutive tiles from the counter of its eighth of the
// buffer (one atomic per 16 KB; the counters of the eight fronts are 256 bytes apart), wave w takes the w-th of them; exactly one
// tile (+ halo) of loads in flight per wave: the next tile is requested when the current one has arrived, then the current one is
// stored (DEPTH 1), or the next request waits until the current tile's stores have been issued (DEPTH 0: nothing overlaps).
template <int DEPTH, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void fir_stream_queue(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, long long ntiles, unsigned *__restrict__ counters,
                                                        float *__restrict__ sink)
{
    extern __shared__ unsigned char smem[];
    unsigned *slot = reinterpret_cast<unsigned *>(smem);      // [2]: the group drawn for the next / the one after
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long span = ntiles / 8, base = (long long)(blockIdx.x & 7) * span;
    const unsigned ngroups = (unsigned)(span / 4);
    unsigned *ctr = counters + 64 * (blockIdx.x & 7);
    float acc = 0.0f;
    f32x4 a[4], b[4], ha = (f32x4)(0.0f), hb = (f32x4)(0.0f);
    auto draw = [&](int i) {                                   // all four waves call it together
        if (threadIdx.x == 0) slot[i] = atomicAdd(ctr, 1u);
        __syncthreads();
        const unsigned g = slot[i];
        return g;
    };
    auto load = [&](f32x4 (&v)[4], f32x4 &h, unsigned g) {
        const long long t = base + 4LL * g + wv;
        if (t > 0) h = ld16<NTL>(x + t * 256 - 64 + lane);
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = ld16<NTL>(x + t * 256 + lane + 64 * j);
    };
    auto store = [&](const f32x4 (&v)[4], const f32x4 &h, unsigned g) {
        const long long t = base + 4LL * g + wv;
        acc += h[0];
#pragma unroll
        for (int j = 0; j < 4; j++) st16<NTS>(y + t * 256 + lane + 64 * j, v[j]);
    };
    unsigned ga = draw(0), gb;
    if (ga < ngroups) {
        load(a, ha, ga);
        while (true) {
            gb = draw(1);
            if (DEPTH == 1) { if (gb < ngroups) load(b, hb, gb); store(a, ha, ga); }
            else { store(a, ha, ga); if (gb < ngroups) load(b, hb, gb); }
            if (gb >= ngroups) break;
            ga = draw(0);
            if (DEPTH == 1) { if (ga < ngroups) load(a, ha, ga); store(b, hb, gb); }
            else { store(b, hb, gb); if (ga < ngroups) load(a, ha, ga); }
            if (ga >= ngroups) break;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

// reference: ONE-SHOT, one 4 KB tile (+ halo) per wave, four waves per workgroup side by side, workgroups in address order inside
// each eighth (eight fronts), residency limited by the LDS allocation
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void fir_stream_oneshot(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, long long ntiles, float *__restrict__ sink)
{
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long span = ntiles / 8, base = (long long)(blockIdx.x & 7) * span;
    const long long t = base + 4LL * (blockIdx.x >> 3) + wv;
    f32x4 v[4], h = (f32x4)(0.0f);
    if (t > 0) h = ld16<NTL>(x + t * 256 - 64 + lane);
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = ld16<NTL>(x + t * 256 + lane + 64 * j);
#pragma unroll
    for (int j = 0; j < 4; j++) st16<NTS>(y + t * 256 + lane + 64 * j, v[j]);
    if (h[0] == 123.456f) sink[0] = h[1] + smem[0];
}

static hipEvent_t e0, e1;
template <typename F>
static void timeit(const char *family, const char *variant, const char *extra, double bytes, F &&launch)
{
    for (int w = 0; w < 3; w++) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 9; r++) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    CHECK(hipGetLastError());
    std::sort(t.begin(), t.end());
    const double tb = bytes / (t[t.size() / 2] * 1e-3) * 1e-12;
    printf("{\"family\": \"%s\", \"variant\": \"%s\"%s, \"ms_median\": %.4f, \"ms_min\": %.4f, \"TBps\": %.3f, \"frac_of_8TBps\": %.3f}\n", family, variant, extra,
           t[t.size() / 2], t[0], tb, tb / 8.0);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const bool only_dynamic = argc > 1 && atoi(argv[1]) == 1;
    const long long total = 1LL << 30;                   // floats: 4096 channels x 2^18 samples, 4 GiB in + 4 GiB out
    const long long ntiles = total / 1024;               // 2^20 tiles of 4 KB
    const double bytes = 8.0 * (double)total;            // algorithmic: the halo re-reads are overhead
    float *xb, *yb, *sink; unsigned *counters;
    CHECK(hipMalloc(&xb, total * 4)); CHECK(hipMalloc(&yb, total * 4)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&counters, 2048));
    CHECK(hipMemset(xb, 1, total * 4)); CHECK(hipMemset(yb, 0, total * 4));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const f32x4 *x = (const f32x4 *)xb; f32x4 *y = (f32x4 *)yb;
    char extra[256];
#define SETUP(K) CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SETUP((fir_stream<0, false, false>)) SETUP((fir_stream<0, false, true>)) SETUP((fir_stream<0, true, true>)) SETUP((fir_stream<0, true, false>))
    SETUP((fir_stream<1, false, false>)) SETUP((fir_stream<1, false, true>)) SETUP((fir_stream<1, true, true>)) SETUP((fir_stream<1, true, false>))
    SETUP((fir_stream<2, false, false>)) SETUP((fir_stream<2, false, true>)) SETUP((fir_stream<2, true, true>)) SETUP((fir_stream<2, true, false>))
    for (int bpc : {2, 1, 4}) {                         // workgroups per CU (2 = the kernel's residency), kept apart by LDS
        if (only_dynamic) break;
        const unsigned blocks = 256u * bpc;
        const size_t lds = (size_t)(160 * 1024 / bpc) - 1024;
#define RUN(ORDER, T, NTL, NTS, NAME) snprintf(extra, sizeof extra, ", \"order\": %d, \"tiles_per_unit\": %d, \"workgroups_per_cu\": %d", ORDER, T, bpc); \
        timeit("fir stream", NAME, extra, bytes, [&] { hipLaunchKernelGGL((fir_stream<ORDER, NTL, NTS>), dim3(blocks), dim3(256), lds, 0, x, y, ntiles, T, sink); });
#define ALLPOL(ORDER, T) RUN(ORDER, T, false, false, "plain loads, plain stores") RUN(ORDER, T, false, true, "plain loads, nt stores") \
        RUN(ORDER, T, true, true, "nt loads, nt stores") RUN(ORDER, T, true, false, "nt loads, plain stores")
        ALLPOL(0, 1)
        for (int T : {1, 2, 4, 8, 16, 64}) { ALLPOL(1, T) }
        for (int T : {1, 2, 4, 8, 16, 64}) { ALLPOL(2, T) }
    }
    // dynamic order (workgroup draws) and the one-shot reference, by residency
#define SETUPQ(K) CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SETUPQ((fir_stream_queue<1, false, false>)) SETUPQ((fir_stream_queue<1, false, true>)) SETUPQ((fir_stream_queue<1, true, true>))
    SETUPQ((fir_stream_queue<0, false, false>)) SETUPQ((fir_stream_queue<0, false, true>)) SETUPQ((fir_stream_queue<0, true, true>))
    SETUPQ((fir_stream_oneshot<false, false>)) SETUPQ((fir_stream_oneshot<false, true>)) SETUPQ((fir_stream_oneshot<true, true>))
    for (int bpc : {1, 2, 3, 4, 6, 8}) {
        const unsigned blocks = 256u * bpc;
        const size_t lds = bpc >= 8 ? 1024 : (size_t)(160 * 1024 / bpc) - 1024;
#define RUNQ(DEPTH, NTL, NTS, NAME) snprintf(extra, sizeof extra, ", \"order\": \"dynamic, workgroup draws 4 tiles\", \"loads_overlap_stores\": %d, \"workgroups_per_cu\": %d", DEPTH, bpc); \
        timeit("fir stream", NAME, extra, bytes, [&] { CHECK(hipMemsetAsync(counters, 0, 2048, 0)); hipLaunchKernelGGL((fir_stream_queue<DEPTH, NTL, NTS>), dim3(blocks), dim3(256), lds, 0, x, y, ntiles, counters, sink); });
        RUNQ(1, false, false, "plain loads, plain stores") RUNQ(1, false, true, "plain loads, nt stores") RUNQ(1, true, true, "nt loads, nt stores")
        RUNQ(0, false, false, "plain loads, plain stores") RUNQ(0, false, true, "plain loads, nt stores") RUNQ(0, true, true, "nt loads, nt stores")
#define RUNO(NTL, NTS, NAME) snprintf(extra, sizeof extra, ", \"order\": \"one-shot, eight fronts\", \"workgroups_per_cu\": %d", bpc); \
        timeit("fir stream", NAME, extra, bytes, [&] { hipLaunchKernelGGL((fir_stream_oneshot<NTL, NTS>), dim3((unsigned)(ntiles / 4)), dim3(256), lds, 0, x, y, ntiles, sink); });
        RUNO(false, false, "plain loads, plain stores") RUNO(false, true, "plain loads, nt stores") RUNO(true, true, "nt loads, nt stores")
    }
    return 0;
}
