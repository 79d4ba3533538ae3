// Probe (diagnostic, not product): operand pairing and accumulator wrap of v_mfma_i32_32x32x32_i8 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_i8_probe tools/probes/mfma_i8_probe.hip && /tmp/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const signed char *A, const signed char *B, int *D, int cinit)
{
    // hypothesis: lane l = (r = l & 31, h = l >> 5) holds A[r][16 h + j] and B[16 h + j][r], j = 0..15
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x4 a, b;
    signed char *ap = (signed char *)&a, *bp = (signed char *)&b;
    for (int j = 0; j < 16; j++) { ap[j] = A[r * 32 + 16 * h + j]; bp[j] = B[(16 * h + j) * 32 + r]; }
    i32x16 c;
    for (int i = 0; i < 16; i++) c[i] = cinit;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; i++) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}
int main()
{
    std::vector<signed char> A(1024), B(1024);
    std::vector<int> D(1024), R(1024);
    srand(5);
    for (auto &v : A) v = (signed char)(rand() % 256 - 128);
    for (auto &v : B) v = (signed char)(rand() % 256 - 128);
    signed char *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    for (int cinit : {0, 2147483000}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, cinit);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
            unsigned s = (unsigned)cinit;
            for (int k = 0; k < 32; k++) s += (unsigned)((int)A[i * 32 + k] * (int)B[k * 32 + j]);
            R[i * 32 + j] = (int)s;
            bad += (R[i * 32 + j] != D[i * 32 + j]);
        }
        printf("cinit %d: %d mismatches of 1024 (D[0][0..3] = %d %d %d %d, want %d %d %d %d)\n", cinit, bad, D[0], D[1], D[2], D[3], R[0], R[1], R[2], R[3]);
    }
    return 0;
}
