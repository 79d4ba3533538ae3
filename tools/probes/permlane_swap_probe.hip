// tools/probes/permlane_swap_probe.hip -- what v_permlane16_swap_b32 / v_permlane32_swap_b32 do to (vdst, src) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    int a = threadIdx.x, b = 100 + threadIdx.x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
    int c = threadIdx.x, d = 100 + threadIdx.x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
    out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d;
}
int main()
{
    int *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane16_swap vdst", "permlane16_swap src ", "permlane32_swap vdst", "permlane32_swap src "};
    for (int r = 0; r < 4; r++) { printf("%s:", names[r]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%d", l, h[64 * r + l]); printf("\n"); }
    return 0;
}
