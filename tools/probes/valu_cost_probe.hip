// Probe (diagnostic): issue cost in cycles of the integer instructions of the Teensy biquad inner loop, one wave alone on a SIMD:
// 256 independent copies of one instruction between two s_memtime reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
#define MEASURE(name, ASM)                                                                                   \
    {                                                                                                        \
        unsigned long long t0 = __builtin_readcyclecounter();                                                \
        asm volatile(REP256(ASM "\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "memory");   \
        unsigned long long t1 = __builtin_readcyclecounter();                                                \
        if (threadIdx.x == 0) out[idx] = (float)(t1 - t0) / 256.0f;                                          \
        idx++;                                                                                               \
    }
__global__ void probe(float *out, int *sink)
{
    int a = threadIdx.x, b = a * 3, c = a * 5, d = a * 7, e = a + 11, f = a ^ 5, idx = 0;
    MEASURE("v_add_u32", "v_add_u32 %0, %4, %5")
    MEASURE("v_mul_i32_i24", "v_mul_i32_i24 %0, %4, %5")
    MEASURE("v_mad_i32_i24", "v_mad_i32_i24 %0, %4, %5, %1")
    MEASURE("v_mul_i32_i24_sdwa", "v_mul_i32_i24_sdwa %0, %4, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
    MEASURE("v_add_u32_sdwa", "v_add_u32_sdwa %0, %4, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
    MEASURE("v_dot2c_i32_i16", "v_dot2c_i32_i16 %0, %4, %5")
    MEASURE("v_perm_b32", "v_perm_b32 %0, %4, %5, %1")
    MEASURE("v_med3_i32", "v_med3_i32 %0, %4, %5, %1")
    MEASURE("v_mul_hi_i32", "v_mul_hi_i32 %0, %4, %5")
    MEASURE("v_bfe_i32", "v_bfe_i32 %0, %4, 0, 16")
    MEASURE("dep v_add_u32", "v_add_u32 %0, %0, %5")
    MEASURE("dep v_add_u32_sdwa", "v_add_u32_sdwa %0, %0, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
    MEASURE("dep v_mul_i32_i24_sdwa", "v_mul_i32_i24_sdwa %0, %0, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
    MEASURE("dep v_dot2c", "v_dot2c_i32_i16 %0, %0, %5")
    MEASURE("dep v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %5, %1")
    sink[threadIdx.x] = a + b + c + d;
}
int main()
{
    float *o, h[16]; int *s;
    hipMalloc(&o, sizeof h); hipMalloc(&s, 64 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o, s);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o, s);
    hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"v_add_u32", "v_mul_i32_i24", "v_mad_i32_i24", "v_mul_i32_i24_sdwa", "v_add_u32_sdwa", "v_dot2c_i32_i16", "v_perm_b32", "v_med3_i32",
                           "v_mul_hi_i32", "v_bfe_i32", "dep v_add_u32", "dep v_add_u32_sdwa", "dep v_mul_i32_i24_sdwa", "dep v_dot2c", "dep v_mad_i32_i24"};
    for (int i = 0; i < 15; i++) printf("%-24s %.2f cycles\n", names[i], h[i]);
    return 0;
}
