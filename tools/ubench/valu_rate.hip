// Micro-benchmark: issue cost (shader cycles per wave64 instruction) of the VALU ops the GroupNorm+SiLU prologue uses.
// One wave per SIMD (grid = 1 block of 256 threads), N independent ops per timed region, s_memtime around it.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define TIMED(name, idx, body)                                                         \
    {                                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
        for (int it = 0; it < 64; ++it) { REP8(body) }                                 \
        asm volatile("s_nop 0" ::: "memory");                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
        if (threadIdx.x == 0) out[idx] = t1 - t0;                                      \
    }
__global__ void k(unsigned long long* out, float* sink, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    TIMED("exp", 0, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    TIMED("rcp", 1, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    TIMED("fma", 2, asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    TIMED("pkfma", 3, asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    TIMED("floor", 4, asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    TIMED("ldexp", 5, asm volatile("v_ldexp_f32 %0, %0, 1\n v_ldexp_f32 %1, %1, 1\n v_ldexp_f32 %2, %2, 1\n v_ldexp_f32 %3, %3, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    TIMED("cvt_pk_bf16", 6, asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    TIMED("exp_f16", 7, asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    TIMED("cvt_i32", 8, asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    TIMED("fract", 9, asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    TIMED("med3", 10, asm volatile("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1];
}
int main()
{
    unsigned long long* d; float* s;
    hipMalloc(&d, 16 * 8); hipMalloc(&s, 256 * 4);
    const char* names[] = {"v_exp_f32", "v_rcp_f32", "v_fma_f32", "v_pk_fma_f32", "v_floor_f32", "v_ldexp_f32", "v_cvt_pk_bf16_f32", "v_exp_f16", "v_cvt_i32_f32", "v_fract_f32", "v_med3_f32"};
    for (int nthreads : {64, 256, 512}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, s, 1.5f);
        unsigned long long h[16];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("block of %d threads (%d wave(s) per SIMD):\n", nthreads, nthreads > 256 ? 2 : 1);
        for (int i = 0; i < 11; ++i) printf("  %-20s %6.2f memtime-ticks per instruction (2048 instrs)\n", names[i], (double)h[i] / 2048.0);
    }
    return 0;
}
