// tools/probes/stream_pattern_probe.hip -- which order of 4 KB tiles lets a "read 4 KB, write 4 KB per wave and step" kernel
// (the access shape of every one-wave-per-stream kernel in this library) reach the HBM rate of a plain copy.
// A measurement aid, not product code.   hipcc --offload-arch=gfx950 -O3 -o stream_pattern_probe stream_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
constexpr int kTile = 1024;

// PAT 0: wave w owns tiles [w iters, (w+1) iters) in order          (one contiguous stream per wave)
// PAT 1: grid stride: step it of wave w = tile it waves + w
// PAT 2: as 0, each wave starting at a different point of its own stretch (rotation by skew(w)), same set of tiles
// PAT 3: as 0 with the write stream one tile behind the read stream irrelevant here; kept for symmetry: reads only
// PAT 4: writes only
template <int PAT, bool NT, int IN_BYTES>
__global__ __launch_bounds__(256) void stream_kernel(const float *__restrict__ x, float *__restrict__ y, int iters, int skew_mul)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
    auto tile_of = [&](int it) -> long long {
        if constexpr (PAT == 1) return (long long)it * waves + wave;
        else if constexpr (PAT == 2) return wave * iters + (it + (int)((wave * skew_mul) % iters)) % iters;
        else return wave * iters + it;
    };
    auto ld = [&](long long t, int j) -> f32x4 {
        if constexpr (IN_BYTES == 2) {          // int16 input: 2 KB per tile: two 8-byte loads per lane... modelled as one 16-byte load of half the lanes' span
            const f32x4 *p = reinterpret_cast<const f32x4 *>(reinterpret_cast<const short *>(x) + t * kTile) + (lane + 64 * (j & 1));
            return NT ? __builtin_nontemporal_load(p) : *p;
        } else {
            const f32x4 *p = reinterpret_cast<const f32x4 *>(x + t * kTile) + (lane + 64 * j);
            return NT ? __builtin_nontemporal_load(p) : *p;
        }
    };
    constexpr int NL = IN_BYTES == 2 ? 2 : 4;
    f32x4 pre[4];
    if constexpr (PAT != 4) {
#pragma unroll
        for (int j = 0; j < NL; j++) pre[j] = ld(tile_of(0), j);
    }
    f32x4 sink = (f32x4)(0.0f);
    for (int it = 0; it < iters; it++) {
        f32x4 cur[4];
        if constexpr (PAT != 4) {
#pragma unroll
            for (int j = 0; j < NL; j++) cur[j] = pre[j];
            if constexpr (NL == 2) { cur[2] = cur[0]; cur[3] = cur[1]; }
            const int nx = it + 1 < iters ? it + 1 : it;
#pragma unroll
            for (int j = 0; j < NL; j++) pre[j] = ld(tile_of(nx), j);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) cur[j] = (f32x4)((float)it);
        }
        if constexpr (PAT != 3) {
            const long long t = tile_of(it);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x4 *p = reinterpret_cast<f32x4 *>(y + t * kTile) + (lane + 64 * j);
                if (NT) __builtin_nontemporal_store(cur[j], p); else *p = cur[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) sink += cur[j];
        }
    }
    if constexpr (PAT == 3) if (sink[0] == 123.456f) y[lane] = sink[1];
}

template <typename K>
static void run(K kern, const char *name, int wps, long long total_tiles, const float *x, float *y, double bytes_per_tile, int skew_mul)
{
    const int waves = 256 * 4 * wps, blocks = waves / 4, iters = (int)(total_tiles / waves);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, y, iters, skew_mul);
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 9; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, y, iters, skew_mul);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double bytes = bytes_per_tile * (double)iters * waves;
    printf("{\"pattern\": \"%s\", \"waves_per_simd\": %d, \"skew_mul\": %d, \"ms_median\": %.4f, \"ms_min\": %.4f, \"TBps\": %.3f, \"frac_of_8TBps\": %.3f}\n",
           name, wps, skew_mul, t[t.size() / 2], t[0], bytes / (t[t.size() / 2] * 1e-3) * 1e-12, bytes / (t[t.size() / 2] * 1e-3) / 8e12);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const long long tiles = argc > 1 ? atoll(argv[1]) : (1LL << 20);
    const size_t n = (size_t)tiles * kTile;
    float *x, *y;
    CHECK(hipMalloc(&x, n * 4)); CHECK(hipMalloc(&y, n * 4));
    CHECK(hipMemset(x, 1, n * 4)); CHECK(hipMemset(y, 0, n * 4));
    for (int wps : {2, 4, 8}) {
        run(stream_kernel<0, false, 4>, "per-wave stream", wps, tiles, x, y, 8192.0, 0);
        run(stream_kernel<0, true, 4>, "per-wave stream, nt", wps, tiles, x, y, 8192.0, 0);
        run(stream_kernel<1, false, 4>, "grid stride", wps, tiles, x, y, 8192.0, 0);
        run(stream_kernel<1, true, 4>, "grid stride, nt", wps, tiles, x, y, 8192.0, 0);
        for (int sk : {1, 5, 37}) {
            run(stream_kernel<2, false, 4>, "per-wave stream, rotated start", wps, tiles, x, y, 8192.0, sk);
            run(stream_kernel<2, true, 4>, "per-wave stream, rotated start, nt", wps, tiles, x, y, 8192.0, sk);
        }
    }
    for (int wps : {4, 8}) {
        run(stream_kernel<3, false, 4>, "reads only, per-wave stream", wps, tiles, x, y, 4096.0, 0);
        run(stream_kernel<3, true, 4>, "reads only, per-wave stream, nt", wps, tiles, x, y, 4096.0, 0);
        run(stream_kernel<4, false, 4>, "writes only, per-wave stream", wps, tiles, x, y, 4096.0, 0);
        run(stream_kernel<4, true, 4>, "writes only, per-wave stream, nt", wps, tiles, x, y, 4096.0, 0);
        run(stream_kernel<0, false, 2>, "int16 in / fp32 out (6 B per sample), per-wave stream", wps, tiles, x, y, 6144.0, 0);
        run(stream_kernel<0, true, 2>, "int16 in / fp32 out, per-wave stream, nt", wps, tiles, x, y, 6144.0, 0);
        run(stream_kernel<2, true, 2>, "int16 in / fp32 out, rotated start, nt", wps, tiles, x, y, 6144.0, 5);
    }
    return 0;
}
