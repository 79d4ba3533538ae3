// tools/probes/copy_order_probe.hip -- second step of the round-3 stream study (first: copy_ceiling_probe.hip).  The plain one-shot
// float4 copy reaches 0.79-0.82 of 8 TB/s where every persistent shape stays at 0.60-0.70.  Which property pays: the ADDRESS ORDER
// (dispatch order = address order: the chip sweeps memory as one compact front), the short LIFE of a wave, or the small number of
// bytes each CU keeps in flight?  Variants of the same 4 GiB -> 4 GiB copy:
//   one-shot, 1 KB per wave, block -> address map permuted (odd-multiplier scramble, 8 per-XCD fronts, bit reversal)
//   one-shot with the number of resident blocks per CU limited by an LDS allocation
//   one-shot, one 4 KB tile per wave (the FIR kernel's tile), 4 waves per block side by side
//   T tiles per wave one after the other (ping-pong prefetch written without register copies), units in address order
//   persistent waves that draw the next tile from a global counter (address order kept, waves long-lived)
// A measurement aid, not product code.   hipcc --offload-arch=gfx950 -O3 -o copy_order_probe copy_order_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ f32x4 ld16(const f32x4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st16(f32x4 *p, f32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// PERM 0: identity; 1: odd multiplier (a bijection mod 2^k); 2: eight fronts, blocks with equal b % 8 (one XCD under round-robin
// placement) sweep one eighth of the buffer; 3: bit reversal of the block index
__device__ __forceinline__ unsigned perm_block(unsigned b, unsigned nb_log2, int perm)
{
    if (perm == 1) return (b * 40503u + 12345u) & ((1u << nb_log2) - 1u);
    if (perm == 2) return ((b & 7u) << (nb_log2 - 3)) | (b >> 3);
    if (perm == 3) return __brev(b) >> (32 - nb_log2);
    return b;
}

// one-shot: a block of 256 threads moves 4 KB (1 KB per wave); `lds_bytes` of dynamic LDS only limit residency
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void oneshot_1k(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, unsigned nb_log2, int perm)
{
    extern __shared__ unsigned char smem[];
    const unsigned pb = perm_block(blockIdx.x, nb_log2, perm);
    const long long i = (long long)pb * 256 + threadIdx.x;
    if (threadIdx.x == 1023) smem[0] = 1;                       // (keeps the allocation)
    st16<NTS>(y + i, ld16<NTL>(x + i));
}

// one-shot: a wave moves one 4 KB tile (4 x 1 KB consecutive), the block's 4 waves take 4 consecutive tiles
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void oneshot_tile(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, unsigned nb_log2, int perm)
{
    const unsigned pb = perm_block(blockIdx.x, nb_log2, perm);
    const long long i = ((long long)pb * 4 + (threadIdx.x >> 6)) * 256 + (threadIdx.x & 63);
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = ld16<NTL>(x + i + 64 * j);
#pragma unroll
    for (int j = 0; j < 4; j++) st16<NTS>(y + i + 64 * j, v[j]);
}

// T tiles per wave, one after the other; wave-units in address order (unit u covers tiles [u T, (u + 1) T)); ping-pong without copies
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void run_tiles(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, int T, unsigned nb_log2, int perm)
{
    const unsigned pb = perm_block(blockIdx.x, nb_log2, perm);
    const long long u = (long long)pb * 4 + (threadIdx.x >> 6);
    const f32x4 *xs = x + u * T * 256 + (threadIdx.x & 63);
    f32x4 *ys = y + u * T * 256 + (threadIdx.x & 63);
    f32x4 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = ld16<NTL>(xs + 64 * j);
    for (int t = 0; t < T; t += 2) {
        if (t + 1 < T) {
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = ld16<NTL>(xs + (long long)(t + 1) * 256 + 64 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) st16<NTS>(ys + (long long)t * 256 + 64 * j, a[j]);
        if (t + 2 < T) {
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = ld16<NTL>(xs + (long long)(t + 2) * 256 + 64 * j);
        }
        if (t + 1 < T) {
#pragma unroll
            for (int j = 0; j < 4; j++) st16<NTS>(ys + (long long)(t + 1) * 256 + 64 * j, b[j]);
        }
    }
}

// persistent waves, each draws its next GROUP of G consecutive tiles from a global counter (the chip sweeps memory in address order
// whatever the waves' speeds); the next draw and its loads are issued before the current group's stores
// (one counter per b % 8 = per XCD under round-robin placement, 256 bytes apart: one word saturates at ~88 draws per microsecond;
// shard s sweeps the s-th eighth of the buffer; ngroups = groups per shard)
template <int G, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void queue_tiles(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, unsigned *__restrict__ counter, unsigned ngroups)
{
    const int lane = threadIdx.x & 63;
    const unsigned shard = blockIdx.x & 7u;
    x += (long long)shard * ngroups * (256 * G); y += (long long)shard * ngroups * (256 * G);
    auto draw = [&]() -> unsigned {
        unsigned g = 0;
        if (lane == 0) g = atomicAdd(counter + 64 * shard, 1u);
        return __builtin_amdgcn_readfirstlane(g);
    };
    auto load = [&](f32x4 (&v)[4 * G], unsigned g) {
#pragma unroll
        for (int j = 0; j < 4 * G; j++) v[j] = ld16<NTL>(x + (long long)g * (256 * G) + lane + 64 * j);
    };
    auto store = [&](const f32x4 (&v)[4 * G], unsigned g) {
#pragma unroll
        for (int j = 0; j < 4 * G; j++) st16<NTS>(y + (long long)g * (256 * G) + lane + 64 * j, v[j]);
    };
    f32x4 a[4 * G], b[4 * G];
    unsigned ga = draw(), gb;
    if (ga >= ngroups) return;
    load(a, ga);
    while (true) {
        gb = draw();
        if (gb < ngroups) load(b, gb);
        store(a, ga);
        if (gb >= ngroups) break;
        ga = draw();
        if (ga < ngroups) load(a, ga);
        store(b, gb);
        if (ga >= ngroups) break;
    }
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

int main()
{
    const long long total = 1LL << 30;                   // floats: 4 GiB in + 4 GiB out (the bench shape of the FIR stage)
    const long long nvec = total / 4;                    // 2^28 vectors of 16 bytes
    const double bytes = 8.0 * (double)total;
    float *xb, *yb; unsigned *counter;
    CHECK(hipMalloc(&xb, total * 4)); CHECK(hipMalloc(&yb, total * 4)); CHECK(hipMalloc(&counter, 2048));
    CHECK(hipMemset(xb, 1, total * 4)); CHECK(hipMemset(yb, 0, total * 4));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const f32x4 *x = (const f32x4 *)xb; f32x4 *y = (f32x4 *)yb;
    char extra[256];
    const char *perm_name[4] = {"address order", "scrambled (odd multiplier)", "eight fronts (b % 8)", "bit-reversed"};

    // 1. one-shot, 1 KB per wave, permuted block -> address map
    for (int perm = 0; perm < 4; perm++) {
        snprintf(extra, sizeof extra, ", \"order\": \"%s\"", perm_name[perm]);
        timeit("one-shot 1 KB per wave", "plain loads, plain stores", extra, bytes, [&] { hipLaunchKernelGGL((oneshot_1k<false, false>), dim3(1u << 20), dim3(256), 0, 0, x, y, 20u, perm); });
        timeit("one-shot 1 KB per wave", "nt loads, nt stores", extra, bytes, [&] { hipLaunchKernelGGL((oneshot_1k<true, true>), dim3(1u << 20), dim3(256), 0, 0, x, y, 20u, perm); });
    }
    // 2. one-shot, 1 KB per wave, residency limited by LDS (160 KB per CU)
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&oneshot_1k<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&oneshot_1k<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int bpc : {1, 2, 4, 6}) {
        const size_t lds = (size_t)(160 * 1024 / bpc) & ~(size_t)1023;
        snprintf(extra, sizeof extra, ", \"order\": \"address order\", \"blocks_per_cu_by_lds\": %d", bpc);
        timeit("one-shot 1 KB per wave", "plain loads, plain stores", extra, bytes, [&] { hipLaunchKernelGGL((oneshot_1k<false, false>), dim3(1u << 20), dim3(256), lds, 0, x, y, 20u, 0); });
        timeit("one-shot 1 KB per wave", "nt loads, nt stores", extra, bytes, [&] { hipLaunchKernelGGL((oneshot_1k<true, true>), dim3(1u << 20), dim3(256), lds, 0, x, y, 20u, 0); });
    }
    // 3. one-shot, one 4 KB tile per wave
    for (int perm = 0; perm < 3; perm++) {
        snprintf(extra, sizeof extra, ", \"order\": \"%s\"", perm_name[perm]);
        timeit("one-shot 4 KB tile per wave", "plain loads, plain stores", extra, bytes, [&] { hipLaunchKernelGGL((oneshot_tile<false, false>), dim3(1u << 18), dim3(256), 0, 0, x, y, 18u, perm); });
        timeit("one-shot 4 KB tile per wave", "nt loads, nt stores", extra, bytes, [&] { hipLaunchKernelGGL((oneshot_tile<true, true>), dim3(1u << 18), dim3(256), 0, 0, x, y, 18u, perm); });
        timeit("one-shot 4 KB tile per wave", "plain loads, nt stores", extra, bytes, [&] { hipLaunchKernelGGL((oneshot_tile<false, true>), dim3(1u << 18), dim3(256), 0, 0, x, y, 18u, perm); });
    }
    // 4. T tiles per wave, units in address order (and scrambled / eight fronts)
    for (int T : {2, 4, 8, 16, 64, 256}) {
        unsigned nbl = 18; for (int t = T; t > 1; t >>= 1) nbl--;
        for (int perm = 0; perm < 3; perm++) {
            snprintf(extra, sizeof extra, ", \"tiles_per_wave\": %d, \"order\": \"%s\"", T, perm_name[perm]);
            timeit("T tiles per wave", "plain loads, plain stores", extra, bytes, [&] { hipLaunchKernelGGL((run_tiles<false, false>), dim3(1u << nbl), dim3(256), 0, 0, x, y, T, nbl, perm); });
            timeit("T tiles per wave", "plain loads, nt stores", extra, bytes, [&] { hipLaunchKernelGGL((run_tiles<false, true>), dim3(1u << nbl), dim3(256), 0, 0, x, y, T, nbl, perm); });
            if (perm == 0) timeit("T tiles per wave", "nt loads, nt stores", extra, bytes, [&] { hipLaunchKernelGGL((run_tiles<true, true>), dim3(1u << nbl), dim3(256), 0, 0, x, y, T, nbl, perm); });
        }
    }
    // 5. persistent waves drawing groups of tiles from per-XCD counters
    for (int wps : {1, 2, 4}) {
        const unsigned blocks = 256u * wps;
#define RUN_Q(G, NTL, NTS, NAME) snprintf(extra, sizeof extra, ", \"waves_per_simd\": %d, \"tiles_per_draw\": %d", wps, G); \
        timeit("persistent, tile queue per XCD", NAME, extra, bytes, [&] { CHECK(hipMemsetAsync(counter, 0, 2048, 0)); hipLaunchKernelGGL((queue_tiles<G, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, x, y, counter, (unsigned)(nvec / 256 / G / 8)); });
        RUN_Q(1, false, false, "plain loads, plain stores") RUN_Q(1, false, true, "plain loads, nt stores") RUN_Q(1, true, true, "nt loads, nt stores")
        RUN_Q(2, false, false, "plain loads, plain stores") RUN_Q(2, false, true, "plain loads, nt stores")
        RUN_Q(4, false, false, "plain loads, plain stores") RUN_Q(4, false, true, "plain loads, nt stores")
    }
    return 0;
}
