// Probe (diagnostic): the lean correctly-rounded sqrt used by the Q15 envelope (no denormal scaling) against sqrtf, exhaustively
// over every float in [0, 2^31] (the range of (float)(I*I + Q*Q)); also the truncated integer results.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float sqrt_rn_lean(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
    const float vp = __builtin_fmaf(-sd, s, x), vs = __builtin_fmaf(-su, s, x);
    float r = (vp <= 0.0f) ? sd : s;
    r = (vs > 0.0f) ? su : r;
    return r;
}
__global__ void check(unsigned long long *bad, unsigned long long *badint, unsigned first, unsigned last)
{
    for (unsigned long long b = first + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= last; b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __int_as_float((int)b);
        const float want = sqrtf(x), got = sqrt_rn_lean(x);
        if (__float_as_int(want) != __float_as_int(got)) atomicAdd(bad, 1ull);
        if ((int)want != (int)got) atomicAdd(badint, 1ull);
    }
}
int main()
{
    unsigned long long *d, h[2] = {0, 0};
    hipMalloc(&d, sizeof h); hipMemset(d, 0, sizeof h);
    const unsigned first = 0x00800000u /* smallest normal */, last = 0x4f000000u /* 2^31 */;
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, d + 1, first, last);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("normal floats in (0, 2^31]: %llu differ from sqrtf, %llu differ after truncation to int\n", h[0], h[1]);
    hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(check, dim3(1), dim3(1), 0, 0, d, d + 1, 0u, 0u);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("x = 0: %llu differ\n", h[0]);
    return 0;
}
