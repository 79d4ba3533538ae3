#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    // mfma layout probe: D = A*B with A[i][k] = (k==0) ? i : 0, B[k][j] = (k==0) ? 1 : 0  -> D[i][j] = i
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    h8 av = (h8)(0), bv = (h8)(0);
    if (lane < 32) { av[0] = (_Float16)(float)lane; bv[0] = (_Float16)1.0f; }
    f16v c = (f16v)(0.0f);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
    for (int r2 = 0; r2 < 16; r2++) out[128 + lane * 16 + r2] = (unsigned)c[r2];
}
int main() {
    unsigned *d; hipMalloc(&d, (128 + 1024) * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128 + 1024]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("r0:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[i]); printf("\nr1:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 + i]);
    printf("\nlane0 rows:"); for (int r = 0; r < 16; r++) printf(" %u", h[128 + r]);
    printf("\nlane32 rows:"); for (int r = 0; r < 16; r++) printf(" %u", h[128 + 32 * 16 + r]);
    printf("\nlane5 rows:"); for (int r = 0; r < 16; r++) printf(" %u", h[128 + 5 * 16 + r]);
    printf("\n");
}
