// tools/probes/mfma_ceiling_probe.hip -- what the matrix pipe alone can deliver for the 256-tap split-fp16 Toeplitz FIR
// (msdr_fir_f32mf.hiph / msdr_fir_f32tr.hiph): the same MFMA count per output as the product kernel -- 54 v_mfma_f32_32x32x16_f16
// or 108 v_mfma_f32_16x16x32_f16 per 1024 outputs and wave -- with operands held in registers, no LDS, no VALU work, random data;
// then the same with the stage's HBM traffic (4 KB in + 4 KB out per 1024 outputs) beside it, and the traffic alone.
// It is a measurement aid, not product code.   hipcc --offload-arch=gfx950 -O3 -o mfma_ceiling_probe mfma_ceiling_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kTile = 1024;      // outputs per wave and iteration

// MODE 0: 54 x 32x32x16 on one accumulator; MODE 1: 108 x 16x16x32 (four accumulators); MODE 2 / 3: 54 x 32x32x16 alternating between
// two / three independent accumulators (summed at the end); IO: stream 4 KB in / 4 KB out per iteration; MFMA: issue the products
template <int MODE, bool IO, bool MFMA>
__global__ __launch_bounds__(256) void ceiling_kernel(const float *__restrict__ x, float *__restrict__ y, const _Float16 *__restrict__ ops,
                                                      int iters, long long *__restrict__ clk)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    f16x8 a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        a[i] = *reinterpret_cast<const f16x8 *>(ops + ((i * 64 + lane) * 8));
        b[i] = *reinterpret_cast<const f16x8 *>(ops + ((6 + i) * 64 + lane) * 8);
    }
    f32x16 acc32 = (f32x16)(0.0f), acc32b = (f32x16)(0.0f), acc32c = (f32x16)(0.0f);
    f32x4 acc16[4] = {(f32x4)(0.0f), (f32x4)(0.0f), (f32x4)(0.0f), (f32x4)(0.0f)};
    const float *xs = x + wave * (long long)iters * kTile;
    float *ys = y + wave * (long long)iters * kTile;
    f32x4 pre[4];
    if constexpr (IO) {
#pragma unroll
        for (int j = 0; j < 4; j++) pre[j] = *reinterpret_cast<const f32x4 *>(xs + 4 * (lane + 64 * j));
    }
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        f32x4 cur[4];
        if constexpr (IO) {
#pragma unroll
            for (int j = 0; j < 4; j++) cur[j] = pre[j];
            const int nx = it + 1 < iters ? it + 1 : it;
#pragma unroll
            for (int j = 0; j < 4; j++) pre[j] = *reinterpret_cast<const f32x4 *>(xs + (long long)nx * kTile + 4 * (lane + 64 * j));
        }
        if constexpr (MFMA) {
            if constexpr (MODE == 0) {
#pragma unroll
                for (int s = 0; s < 18; s++) {
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % 6], b[(s + 1) % 6], acc32, 0, 0, 0);
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + 2) % 6], b[(s + 3) % 6], acc32, 0, 0, 0);
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + 4) % 6], b[(s + 5) % 6], acc32, 0, 0, 0);
                }
            } else if constexpr (MODE == 2) {
#pragma unroll
                for (int s = 0; s < 18; s += 2) {
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % 6], b[(s + 1) % 6], acc32, 0, 0, 0);
                    acc32b = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + 1) % 6], b[(s + 2) % 6], acc32b, 0, 0, 0);
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + 2) % 6], b[(s + 3) % 6], acc32, 0, 0, 0);
                    acc32b = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + 3) % 6], b[(s + 4) % 6], acc32b, 0, 0, 0);
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + 4) % 6], b[(s + 5) % 6], acc32, 0, 0, 0);
                    acc32b = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + 5) % 6], b[s % 6], acc32b, 0, 0, 0);
                }
            } else if constexpr (MODE == 3) {
#pragma unroll
                for (int s = 0; s < 18; s++) {
                    acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % 6], b[(s + 1) % 6], acc32, 0, 0, 0);
                    acc32b = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + 2) % 6], b[(s + 3) % 6], acc32b, 0, 0, 0);
                    acc32c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s + 4) % 6], b[(s + 5) % 6], acc32c, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 9; s++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        acc16[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(s + q) % 6], b[(s + 1) % 6], acc16[q], 0, 0, 0);
                        acc16[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(s + q + 2) % 6], b[(s + 3) % 6], acc16[q], 0, 0, 0);
                        acc16[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(s + q + 4) % 6], b[(s + 5) % 6], acc16[q], 0, 0, 0);
                    }
            }
        }
        if constexpr (IO) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x4 v = cur[j];
                if constexpr (MFMA) {
                    if constexpr (MODE == 1) v[0] += acc16[j][0] * 1e-30f; else v[0] += (acc32[j] + acc32b[j] + acc32c[j]) * 1e-30f;
                }
                *reinterpret_cast<f32x4 *>(ys + (long long)it * kTile + 4 * (lane + 64 * j)) = v;
            }
        }
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { clk[2 * wave] = c1 - c0; clk[2 * wave + 1] = r1 - r0; }
    if constexpr (!IO) {            // keep the accumulators alive
        float s = 0.0f;
        if constexpr (MODE != 1) { for (int i = 0; i < 16; i++) s += acc32[i] + acc32b[i] + acc32c[i]; } else { for (int q = 0; q < 4; q++) for (int i = 0; i < 4; i++) s += acc16[q][i]; }
        if (s == 123.456f) ys[lane] = s;
    }
}

struct Result { double ms_med, ms_min, ghz; };

template <typename K>
static Result run(K kern, const char *name, int wps, long long total_tiles, const float *x, float *y, const _Float16 *ops, long long *clk, bool io)
{
    const int waves = 256 * 4 * wps, blocks = waves / 4, iters = (int)(total_tiles / waves);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 10; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, y, ops, iters, clk);       // warm-up: let the clock settle under load
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 15; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, y, ops, iters, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    std::vector<long long> h(2 * (size_t)waves);
    CHECK(hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<double> g;
    for (int w = 0; w < waves; w++) if (h[2 * w + 1] > 0) g.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);    // s_memrealtime: 100 MHz
    std::sort(g.begin(), g.end());
    Result res{t[t.size() / 2], t[0], g.empty() ? 0.0 : g[g.size() / 2]};
    const double samples = (double)iters * waves * kTile;
    printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"tiles\": %lld, \"ms_median\": %.4f, \"ms_min\": %.4f, \"in_kernel_clock_GHz\": %.3f, "
           "\"Gsamples_per_s\": %.1f, \"frac_of_8TBps_at_8B\": %.3f%s}\n",
           name, wps, (long long)iters * waves, res.ms_med, res.ms_min, res.ghz, samples / res.ms_med * 1e-6, samples * 8.0 / (res.ms_med * 1e-3) / 8e12,
           io ? ", \"hbm\": \"4 KB in + 4 KB out per tile\"" : "");
    fflush(stdout);
    return res;
}

int main(int argc, char **argv)
{
    const long long tiles = argc > 1 ? atoll(argv[1]) : (1LL << 20);       // 2^20 tiles of 1024 = the bench's 4096 channels x 2^18
    const size_t n = (size_t)tiles * kTile;
    float *x, *y; _Float16 *ops; long long *clk;
    CHECK(hipMalloc(&x, n * 4)); CHECK(hipMalloc(&y, n * 4));
    CHECK(hipMalloc(&ops, 12 * 64 * 8 * 2)); CHECK(hipMalloc(&clk, 2 * 8192 * sizeof(long long)));
    {
        std::vector<float> hx((size_t)1 << 24);
        unsigned s = 12345u;
        for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 8000.0f; }
        for (size_t o = 0; o < n; o += hx.size()) CHECK(hipMemcpy(x + o, hx.data(), std::min(hx.size(), n - o) * 4, hipMemcpyHostToDevice));
        std::vector<_Float16> ho(12 * 64 * 8);
        for (auto &v : ho) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23))); }
        CHECK(hipMemcpy(ops, ho.data(), ho.size() * 2, hipMemcpyHostToDevice));
    }
    for (int wps : {1, 2, 4}) {
        run(ceiling_kernel<0, false, true>, "mfma_only_32x32x16_f16 (54 per 1024 outputs)", wps, tiles, x, y, ops, clk, false);
        run(ceiling_kernel<1, false, true>, "mfma_only_16x16x32_f16 (108 per 1024 outputs)", wps, tiles, x, y, ops, clk, false);
        run(ceiling_kernel<2, false, true>, "mfma_only_32x32x16_f16, two accumulators alternating", wps, tiles, x, y, ops, clk, false);
        run(ceiling_kernel<3, false, true>, "mfma_only_32x32x16_f16, three accumulators alternating", wps, tiles, x, y, ops, clk, false);
    }
    for (int wps : {1, 2, 4, 8}) run(ceiling_kernel<0, true, false>, "stream_only", wps, tiles, x, y, ops, clk, true);
    for (int wps : {1, 2, 4}) {
        run(ceiling_kernel<0, true, true>, "mfma_32x32x16 + stream", wps, tiles, x, y, ops, clk, true);
        run(ceiling_kernel<1, true, true>, "mfma_16x16x32 + stream", wps, tiles, x, y, ops, clk, true);
    }
    return 0;
}
