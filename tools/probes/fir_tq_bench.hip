// tools/probes/fir_tq_bench.hip -- fir_f32tq_kernel (minimal-sdr_amd/csrc/msdr_fir_f32tq.hiph) launched on its own, without the library:
// the product instantiation next to its diagnostic ablations (no matrix instructions / no global traffic / no tile queue), on the bench
// shape of the FIR stage (4096 channels x 2^18 samples, 256 taps, random fp32 samples).  Answers "what bounds the kernel": if the stream
// alone and the products alone are both well below the whole, the rest is the package power cap (their energies add, their times
// do not).  A measurement aid, not product code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I minimal-sdr_amd/csrc -o tools/probes/fir_tq_bench tools/probes/fir_tq_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "msdr_shared.h"
#include "msdr_fir_f32tq.hiph"
#include "fir_f32tq_abl_kernel.hip"      // the kernel with its ablation switches: a copy kept beside this probe (the product header has none)

using namespace msdr;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float *x, size_t n, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * 8000.0f;
    }
}

static double g_seconds = 0.0;       // > 0: loop the variant for that long (power / clock sampling from outside), then report
template <int ABL>
static double run(const char *name, TqParams q, unsigned grid, size_t lds, unsigned *d_ctr, int fronts)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fir_f32tq_abl_kernel<9, false, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<float> t;
    int flip = 0;
    CHECK(hipMemset(d_ctr, 0, 2 * kTqMaxFronts * kTqCtrStride * 4));
    const int reps = g_seconds > 0 ? (int)(g_seconds / 1.5e-3) : 60;
    for (int r = 0; r < reps; r++) {
        q.ctr = d_ctr + (size_t)flip * kTqMaxFronts * kTqCtrStride; q.ctr_next = d_ctr + (size_t)(flip ^ 1) * kTqMaxFronts * kTqCtrStride;
        flip ^= 1;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((fir_f32tq_abl_kernel<9, false, ABL>), dim3(grid), dim3(256), lds, 0, q);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 20 && t.size() < 4096) t.push_back(ms);
    }
    CHECK(hipGetLastError());
    std::sort(t.begin(), t.end());
    const double med = t[t.size() / 2];
    printf("{\"kernel\": \"fir_f32tq_kernel<9>\", \"variant\": \"%s\", \"fronts\": %d, \"ms_median\": %.4f, \"ms_min\": %.4f, \"frac_of_8TBps_at_8B\": %.3f}\n", name, fronts, med, t[0],
           8.0 * q.channels * (double)q.n / (med * 1e-3) / 8e12);
    fflush(stdout);
    return med;
}

int main(int argc, char **argv)
{
    const int channels = 4096, N = 256;
    const long long n = 1LL << 18;
    const int fronts = argc > 1 ? atoi(argv[1]) : 32;
    const int only = argc > 2 ? atoi(argv[2]) : -1;            // one variant (its ABL bits) ...
    g_seconds = argc > 3 ? atof(argv[3]) : 0.0;                // ... looped for this many seconds
    float *x, *y, *hist; char *tab; unsigned *ctr;
    CHECK(hipMalloc(&x, (size_t)channels * n * 4)); CHECK(hipMalloc(&y, (size_t)channels * n * 4));
    const int hist_len = 255;
    CHECK(hipMalloc(&hist, (size_t)channels * hist_len * 4)); CHECK(hipMemset(hist, 0, (size_t)channels * hist_len * 4));
    CHECK(hipMalloc(&ctr, 2 * kTqMaxFronts * kTqCtrStride * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, x, (size_t)channels * n, 12345u);
    // tap table as msdr_fir_f32_create builds it (lowpass, Kaiser window)
    std::vector<float> h(N);
    for (int k = 0; k < N; k++) { const double t = k - (N - 1) / 2.0, s = t == 0 ? 1.0 : std::sin(0.7330382858 * t) / (0.7330382858 * t); h[k] = (float)(0.2333 * s * (0.54 - 0.46 * std::cos(2 * M_PI * k / (N - 1)))); }
    const int H = fm_halo(N), trs = tr_steps(N);
    double maxabs = 0; for (float v : h) maxabs = std::max(maxabs, (double)std::fabs(v));
    int ex = 0; (void)std::frexp(maxabs, &ex); ex = 14 - ex;
    std::vector<char> tb(tr_table_bytes(trs), 0);
    _Float16 *tf = reinterpret_cast<_Float16 *>(tb.data() + kTrHdrBytes);
    for (int F = 0; F < 2; F++) for (int st = 0; st < trs; st++) for (int l = 0; l < 64; l++) for (int jj = 0; jj < 8; jj++) {
        const int kk = 32 * st + 8 * (l >> 4) + jj, d = H + 16 * F + (l & 15) - kk;
        const double val = (d >= 0 && d < N) ? std::ldexp((double)h[N - 1 - d], ex) : 0.0;
        const _Float16 vh = (_Float16)val;
        const size_t o = ((size_t)(F * trs + st) * 2) * 512 + l * 8 + jj;
        tf[o] = vh; tf[o + 512] = (_Float16)(val - (double)vh);
    }
    F32TrHeader hd; hd.ns = trs; hd.ex = ex; hd.fixed_k = 0; hd.use_fixed = 0; hd.skip1 = 0;
    memcpy(tb.data(), &hd, sizeof hd);
    CHECK(hipMalloc(&tab, tb.size())); CHECK(hipMemcpy(tab, tb.data(), tb.size(), hipMemcpyHostToDevice));
    if (trs != 9) { printf("unexpected step count %d\n", trs); return 1; }

    TqParams q;
    q.x = x; q.y = y; q.hist = hist; q.tab = tab; q.n = n; q.channels = channels; q.hist_len = hist_len;
    q.tpr = (unsigned)(n / kTrTile); q.tpr_shift = 8; q.total = q.tpr * channels; q.fronts = fronts; q.per_front = (q.total + fronts - 1) / fronts;
    q.all_aligned = 1;
    q.run_shift = getenv("MSDR_TQ_RUN_SHIFT") ? atoi(getenv("MSDR_TQ_RUN_SHIFT")) : 2;
    const unsigned grid = 512;
    const size_t lds = 4 * tr_wave_bytes(trs);
    CHECK(hipDeviceSynchronize());
    if (only >= 0) {
        switch (only) {
        case 0: run<0>("product", q, grid, lds, ctr, fronts); break;
        case 1: run<1>("no matrix instructions", q, grid, lds, ctr, fronts); break;
        case 2: run<2>("no global loads / stores in the steady state", q, grid, lds, ctr, fronts); break;
        case 3: run<3>("neither", q, grid, lds, ctr, fronts); break;
        case 8: run<8>("30 of 36 product triples (fast-FIR product count)", q, grid, lds, ctr, fronts); break;
        case 24: run<24>("30 of 36 product triples + the split's side work (fast-FIR upper bound)", q, grid, lds, ctr, fronts); break;
        case 10: run<10>("30 of 36 product triples, no global traffic", q, grid, lds, ctr, fronts); break;
        default: printf("variant %d not built\n", only); return 1;
        }
        return 0;
    }
    for (int rep = 0; rep < 2; rep++) {
        run<0>("product", q, grid, lds, ctr, fronts);
        run<8>("30 of 36 product triples (fast-FIR product count)", q, grid, lds, ctr, fronts);
        run<24>("30 of 36 product triples + the split's side work (fast-FIR upper bound)", q, grid, lds, ctr, fronts);
        run<10>("30 of 36 product triples, no global traffic", q, grid, lds, ctr, fronts);
        run<1>("no matrix instructions", q, grid, lds, ctr, fronts);
        run<2>("no global loads / stores in the steady state", q, grid, lds, ctr, fronts);
        run<3>("neither", q, grid, lds, ctr, fronts);
        run<4>("round-robin tiles instead of the queue", q, grid, lds, ctr, fronts);
    }
    return 0;
}
