// Probe (diagnostic): is (int)v_sqrt_f32(f) == (int)sqrtf(f) for every INTEGER-valued float f in [0, 2^31] (the values (float)(I*I + Q*Q) takes)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void check(unsigned long long *bad, unsigned long long *first_bad)
{
    const unsigned long long total = (1ull << 23) + (0x4f000000ull - 0x4b000000ull + 1);     // integers < 2^23, then every float in [2^23, 2^31]
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = i < (1ull << 23) ? (float)(unsigned)i : __int_as_float((int)(0x4b000000ull + (i - (1ull << 23))));
        const int want = (int)sqrtf(x), got = (int)__builtin_amdgcn_sqrtf(x);
        if (want != got) { atomicAdd(bad, 1ull); atomicMin(first_bad, (unsigned long long)__float_as_uint(x)); }
    }
}
int main()
{
    unsigned long long *d, h[2] = {0, ~0ull};
    hipMalloc(&d, sizeof h); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, d + 1);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("integer-valued floats in [0, 2^31]: %llu truncated results differ (first at bits 0x%llx)\n", h[0], h[1]);
    return 0;
}
