// Probe (diagnostic): integer helper instructions the Q15 matrix-core kernel relies on (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void probe(int *out)
{
    const s16x2 m = {(short)-32768, (short)-32768};
    out[0] = __builtin_amdgcn_sdot2(m, m, 0, false);                         // want -2147483648 (wrap)
    const s16x2 p = __builtin_amdgcn_cvt_pk_i16(70000, -70000);              // want 32767, -32768
    out[1] = p[0]; out[2] = p[1];
    const s16x2 q = __builtin_amdgcn_cvt_pk_i16(-5, 1234);
    out[3] = q[0]; out[4] = q[1];
    unsigned w = 0x80007fffu, k = 0xffffffffu, r;                            // (-32768, 32767) * (-1, -1) per half, wrap
    asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(k));
    out[5] = (int)r;                                                         // want 0x80008001
    unsigned a = 0x11223344u, b = 0x55667788u, sel = 0x07050301u, pr;        // v_perm_b32: D.byte[i] = {S0,S1}.byte[sel.byte[i]], S1 = bytes 0-3, S0 = bytes 4-7
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pr) : "v"(a), "v"(b), "v"(sel));
    out[6] = (int)pr;
    int acc = 100, t = (int)0xfffe1234;                                      // SDWA: acc + sext(t.word1) = 100 + (-2) = 98
    asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(out[7]) : "v"(acc), "v"(t));
    int o8;
    asm volatile("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(o8) : "v"(acc), "v"(t));
    out[8] = o8;
}
int main()
{
    int *d, h[9];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("sdot2 wrap: %d (want -2147483648)\ncvt_pk_i16: %d %d (want 32767 -32768), %d %d (want -5 1234)\npk_mul_lo_u16: 0x%08x (want 0x80008001)\n"
           "perm: 0x%08x\nsdwa zext word1: %d, sext word1: %d (want 98)\n", h[0], h[1], h[2], h[3], h[4], (unsigned)h[5], (unsigned)h[6], h[7], h[8]);
    return 0;
}
