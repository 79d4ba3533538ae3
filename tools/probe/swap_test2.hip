#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void swap_asm(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__global__ void k(float *out, const float *in, int mode) {
    unsigned lane = threadIdx.x;
    float dv[16];
    for (int r = 0; r < 16; r++) dv[r] = in[lane * 16 + r] * 2.0f;
    if (mode == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, dv[r]), __builtin_bit_cast(unsigned, dv[r + 8]), false, false);
            dv[r] = __builtin_bit_cast(float, sw[0]);
            dv[r + 8] = __builtin_bit_cast(float, sw[1]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) swap_asm(dv[r], dv[r + 8]);
    }
    float d[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) d[k2] = dv[(k2 & 3) + ((k2 & 4) << 1) + ((k2 & 8) >> 1)];
    for (int r = 0; r < 16; r++) out[lane * 16 + r] = d[r];
}
int main() {
    float *d, *in; hipMalloc(&d, 1024 * 4); hipMalloc(&in, 1024 * 4);
    float h[1024];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) h[l * 16 + r] = (l & 31) * 32 + 4 * (l >> 5) + (r & 3) + 8 * (r >> 2);   // = sample index / 2... scaled by 2 in kernel
    for (int i = 0; i < 1024; i++) h[i] *= 0.5f;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; mode++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, in, mode);
        float o[1024]; hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) { float want = (l & 31) * 32 + 16 * (l >> 5) + r; if (o[l * 16 + r] != want) bad++; }
        printf("mode %d bad %d  lane0:", mode, bad); for (int r = 0; r < 16; r++) printf(" %g", o[r]); printf("\n");
    }
}
