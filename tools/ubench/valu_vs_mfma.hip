// Micro-benchmark: cost of VALU ops in one wave while ANOTHER wave on the same SIMD streams MFMAs back to back
// (the situation of a producer wave next to a consumer wave).  Block of 512 threads: waves 0-3 issue
// v_mfma_f32_32x32x16_bf16 in a loop, waves 4-7 (one per SIMD) time 2048 instructions of each kind.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP8(x) x x x x x x x x
#define TIMED(idx, body)                                                               \
    {                                                                                  \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
        for (int it = 0; it < 64; ++it) { REP8(body) }                                 \
        asm volatile("s_nop 0" ::: "memory");                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
        if (threadIdx.x == 256) out[idx] = t1 - t0;                                    \
    }
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, float seed, int mfma_iters, int which, int delay)
{
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < mfma_iters; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) { out[30] = t1 - t0; }
        sink[threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
        return;
    }
    if (delay) { const unsigned long long t = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t < (unsigned long long)delay) __builtin_amdgcn_s_sleep(8); }
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a0, a3}, p3 = {a1, a2};
    if (which == 0) TIMED(0, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 1) TIMED(1, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 2) TIMED(2, asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 3) TIMED(3, asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    if (which == 4) TIMED(4, asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    if (which == 5) TIMED(5, asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 6) TIMED(6, asm volatile("v_lshlrev_b32 %0, 16, %0\n v_and_b32 %1, 0xffff0000, %1\n v_lshlrev_b32 %2, 16, %2\n v_and_b32 %3, 0xffff0000, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 7) TIMED(7, asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 8) TIMED(8, asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    if (which == 9) TIMED(9, asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 10) TIMED(10, asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %0\n v_pk_mul_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %2" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    if (which == 11) TIMED(11, asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 13) TIMED(13, asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0" : "+v"(a0));)
    if (which == 14) TIMED(14, asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1" : "+v"(a0), "+v"(a1));)
    if (which == 15) TIMED(15, asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %0, %0, %0" : "+v"(p0));)
    if (which == 16) TIMED(16, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0" : "+v"(a0));)
    if (which == 17) TIMED(17, asm volatile("v_exp_f32 %0, %0\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(a1));)
    if (which == 18) TIMED(18, asm volatile("v_mul_f32 %0, %1, %2\n v_mul_f32 %1, %2, %3\n v_mul_f32 %2, %3, %0\n v_mul_f32 %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 19) TIMED(19, asm volatile("v_pk_mul_f32 %0, %1, %2\n v_pk_mul_f32 %1, %2, %3\n v_pk_mul_f32 %2, %3, %0\n v_pk_mul_f32 %3, %0, %1" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    if (which == 20) TIMED(20, asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    if (which == 21) TIMED(21, asm volatile("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %0\n v_fma_f32 %2, %3, %0, %1\n v_fma_f32 %3, %0, %1, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 22) TIMED(22, asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 23) TIMED(23, asm volatile("v_mul_f32 %0, s4, %0\n v_mul_f32 %1, s5, %1\n v_mul_f32 %2, s6, %2\n v_mul_f32 %3, s7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 24) TIMED(24, asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %1, %2, %3\n v_cvt_pk_bf16_f32 %2, %3, %0\n v_cvt_pk_bf16_f32 %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 25) TIMED(25, asm volatile("v_lshlrev_b32 %0, 16, %1\n v_and_b32 %1, 0xffff0000, %2\n v_lshlrev_b32 %2, 16, %3\n v_and_b32 %3, 0xffff0000, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    if (which == 12) TIMED(12, asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0] + p3[1];
}
int main()
{
    unsigned long long* d; float* s;
    hipMalloc(&d, 32 * 8); hipMalloc(&s, 512 * 4);
    const char* names[] = {"v_exp_f32", "v_rcp_f32", "v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_cvt_pk_bf16_f32", "v_lshl/v_and", "v_mul_f32", "v_pk_add_f32", "v_add_f32", "pk_mul+pk_add dep", "v_fmac_f32", "v_add_u32/v_xor", "v_mul_f32 1 chain", "v_mul_f32 2 chains", "v_pk_mul_f32 1 chain", "v_exp_f32 1 chain", "exp,add,rcp,mul chain", "v_mul_f32 d,a,b distinct", "v_pk_mul_f32 d,a,b distinct", "v_pk_mul_f32 d,d,a", "v_fma_f32 4 distinct", "v_mul_f32 d,d,a", "v_mul_f32 d,sgpr,d", "v_cvt_pk_bf16 d,a,b", "v_lshl/v_and d,a"};
    for (int delay : {0, 20000, 100000, 300000})
    for (int iters : {0, 4000}) {
        if (delay && !iters) continue;
        printf("delay %d cycles before the timed section:\n", delay);
        hipMemset(d, 0, 32 * 8);
        unsigned long long h[32];
        for (int w = 0; w < 26; ++w) {       // one launch per op: the MFMA waves (4000 x 4 x 32 cycles) outlast every timed section but the starved one
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, s, 1.5f, iters, w, delay);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s (mfma loop: %.1f ticks per MFMA):\n", iters ? "next to an MFMA-streaming wave" : "alone on the SIMD", iters ? (double)h[30] / (4.0 * iters) : 0.0);
        for (int i = 0; i < 26; ++i) printf("  %-20s %6.2f ticks per instruction\n", names[i], (double)h[i] / 2048.0);
    }
    return 0;
}
