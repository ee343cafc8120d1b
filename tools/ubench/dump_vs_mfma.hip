// Micro-benchmark: the producers' per-item work (GroupNorm-apply + SiLU on 16 bytes, mask, ds_write_b128) timed in one
// wave per SIMD, alone and next to a wave that streams MFMAs.  Variants isolate the pieces.
#include "../../clip-neural-image-conpression_amd/csrc/ccn_device.h"
#include <cstdio>
using namespace ccn;
template <int VAR, bool SWAP>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, const float2* ab, float seed, int mfma_iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (SWAP ? wave >= 4 : wave < 4) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
        for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        sink[threadIdx.x] = s;
        return;
    }
    GnCoef<__bf16> gk;
    gk.load(ab + (lane & 7) * 8, true);
    u32x4 areg[11];
    for (int i = 0; i < 11; ++i) areg[i] = u32x4{0x3f803f80u + lane + i, 0x3f003f00u + i, 0x40004000u + lane, 0x3e803e80u};
    const bool ok = (lane & 31) != 5;
    unsigned char* const As = smem + (wave & 3) * 16384;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            u32x4 v = areg[i];
            asm volatile("" : "+v"(v));
            if (VAR == 10) {
                // epilogue-style statistics on packed bf16 pairs with v_dot2c_f32_bf16
                float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16x2 pk = __builtin_bit_cast(bf16x2, v[e]);
                    acc0 = __builtin_amdgcn_fdot2_f32_bf16(pk, pk, acc0, false);
                    acc1 = __builtin_amdgcn_fdot2_f32_bf16(pk, __builtin_bit_cast(bf16x2, 0x3f803f80u), acc1, false);
                }
                v = u32x4{__float_as_uint(acc0), __float_as_uint(acc1), v[2], v[3]};
            } else if (VAR >= 5) {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned u = v[e];
                    f32x2 x = {bf_lo(u), bf_hi(u)};
                    if (VAR >= 6) x = x * gk.a[e];
                    if (VAR >= 7) x = x + gk.c[e];
                    if (VAR >= 8) o[e] = pack_bf2(x[0], x[1]);
                    else o[e] = __float_as_uint(x[0]) ^ (__float_as_uint(x[1]) >> 16);
                    if (VAR >= 9) o[e] = ok ? o[e] : 0u;
                }
                v = o;
            } else if (VAR != 1) {
                const u32x4 tr = VAR == 2 ? gk.template apply<false>(v) : gk.template apply<true>(v);
                if (VAR != 3) v = u32x4{ok ? tr[0] : 0u, ok ? tr[1] : 0u, ok ? tr[2] : 0u, ok ? tr[3] : 0u};
                else v = tr;
            }
            if (VAR != 4) *(u32x4*)(As + (i * 64 + lane) * 16) = v;
            else asm volatile("" :: "v"(v));
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == (SWAP ? 0 : 256)) out[VAR] = t1 - t0;
}
template <int VAR, bool SWAP> void run(unsigned long long* d, float* s, float2* ab, int iters)
{
    hipLaunchKernelGGL((k<VAR, SWAP>), dim3(1), dim3(512), 65536, 0, d, s, ab, 1.5f, iters);
    hipDeviceSynchronize();
}
int main()
{
    unsigned long long* d; float* s; float2* ab;
    hipMalloc(&d, 32 * 8); hipMalloc(&s, 512 * 4); hipMalloc(&ab, 4096);
    hipMemset(ab, 0x3f, 4096);
    const char* names[] = {"full item (affine+SiLU, mask, ds_write)", "ds_write only", "affine only, mask, ds_write", "SiLU, no mask, ds_write", "SiLU, mask, no ds_write", "unpack", "unpack, pk_mul", "unpack, pk_mul, pk_add", "unpack, pk_mul, pk_add, cvt_pk", "unpack, pk_mul, pk_add, cvt_pk, cndmask", "8 x v_dot2c_f32_bf16"};
    for (int swap = 0; swap < 2; ++swap)
    for (int iters : {0, 6000}) {
        hipMemset(d, 0, 32 * 8);
        printf("timed waves are the %s ones of the block; ", swap ? "OLDER (0-3)" : "younger (4-7)");
        if (swap) continue;
        run<0, false>(d, s, ab, iters); run<1, false>(d, s, ab, iters); run<2, false>(d, s, ab, iters); run<3, false>(d, s, ab, iters); run<4, false>(d, s, ab, iters);
        run<5, false>(d, s, ab, iters); run<6, false>(d, s, ab, iters); run<7, false>(d, s, ab, iters); run<8, false>(d, s, ab, iters); run<9, false>(d, s, ab, iters); run<10, false>(d, s, ab, iters);
        unsigned long long h[32];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s:\n", iters ? "next to an MFMA-streaming wave" : "alone on the SIMD");
        for (int i = 0; i < 11; ++i) printf("  %-44s %7.1f cycles per item\n", names[i], (double)h[i] / (64.0 * 11.0));
    }
    return 0;
}
