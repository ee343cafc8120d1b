// Device-side helpers shared by the convolution kernels (types, 16-byte pack/unpack, MFMA wrappers,
// the GroupNorm+SiLU prologue transform).
#pragma once
#include "ccn_internal.h"

namespace ccn {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { A_NHWC = 0, A_IM2COL = 1 };
enum { EPI_NHWC = 0, EPI_HEAD = 1 };

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32, RNE
}

template <typename T> __device__ __forceinline__ float silu_f(float v);
template <> __device__ __forceinline__ float silu_f<float>(float v) { return v / (1.0f + expf(-v)); }
template <> __device__ __forceinline__ float silu_f<__bf16>(float v) { return __fdividef(v, 1.0f + __expf(-v)); }

// 16 bytes of T -> EPC floats and back
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int EPC = 4;
    static __device__ __forceinline__ void unpack(const u32x4& r, float* v) {
        v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
    }
    static __device__ __forceinline__ u32x4 pack(const float* v) {
        return u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    }
};
template <> struct Vec16<__bf16> {
    static constexpr int EPC = 8;
    static __device__ __forceinline__ void unpack(const u32x4& r, float* v) {
        v[0] = bf_lo(r.x); v[1] = bf_hi(r.x); v[2] = bf_lo(r.y); v[3] = bf_hi(r.y);
        v[4] = bf_lo(r.z); v[5] = bf_hi(r.z); v[6] = bf_lo(r.w); v[7] = bf_hi(r.w);
    }
    static __device__ __forceinline__ u32x4 pack(const float* v) {
        return u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    }
};

template <typename T> __device__ __forceinline__ void mfma16(f32x16& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mfma16<float>(f32x16& acc, const u32x4& a, const u32x4& b) {
    const f32x4 a4 = __builtin_bit_cast(f32x4, a), b4 = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], b4[q], acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mfma16<__bf16>(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// v_mfma_f32_16x16x32_bf16 on quarter q (registers 4q .. 4q+3) of a 16-register accumulator tuple
typedef float f32x4q __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma16q(f32x16& acc, const int q, const u32x4& a, const u32x4& b) {
    f32x4q t = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), t, 0, 0, 0);
    acc[4 * q] = t[0]; acc[4 * q + 1] = t[1]; acc[4 * q + 2] = t[2]; acc[4 * q + 3] = t[3];
}


// ---- GroupNorm-apply (+ SiLU) on one 16-byte chunk while staging the A operand ----------------------------
// fp32 (parity mode): y = fma(x, a, c); silu(y) = y / (1 + expf(-y)) with full-precision expf and a true division.
// bf16 (throughput mode): packed fp32 math, exp2 with log2(e) folded into a second scale/shift pair, v_rcp.
template <typename T> struct GnCoef;
template <> struct GnCoef<float> {
    float a[4], c[4];
    // vector loads issued together, ONE branch (a per-element select makes hipcc wait for each load in turn)
    __device__ __forceinline__ void load(const float2* ab, bool valid) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = 1.f; c[e] = 0.f; }
        if (valid) {
            // table layout per channel pair: {scale(2p), scale(2p+1), shift(2p), shift(2p+1)}
            const f32x4 v0 = *(const f32x4*)ab, v1 = *(const f32x4*)(ab + 2);
            a[0] = v0[0]; a[1] = v0[1]; c[0] = v0[2]; c[1] = v0[3];
            a[2] = v1[0]; a[3] = v1[1]; c[2] = v1[2]; c[3] = v1[3];
        }
    }
    template <bool SILU> __device__ __forceinline__ u32x4 apply(const u32x4& raw) const {
        float v[4];
        Vec16<float>::unpack(raw, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = fmaf(v[e], a[e], c[e]);
            if (SILU) y = y / (1.0f + expf(-y));
            v[e] = y;
        }
        return Vec16<float>::pack(v);
    }
};
// NO packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) here or anywhere in a producer wave: next to a
// wave that streams MFMAs they are starved outright -- tools/ubench/dump_vs_mfma.hip: this transform goes from 150 to
// >2300 cycles per 16 bytes as soon as one v_pk_mul_f32 is in it, while v_fma_f32 / v_mul_f32 / v_exp_f32 lose ~25 %.
// The file is built with -fno-slp-vectorize so that scalar code stays scalar.
template <> struct GnCoef<__bf16> {
    float a[8], c[8];
    __device__ __forceinline__ void load(const float2* ab, bool valid) {
        f32x4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f32x4{1.f, 1.f, 0.f, 0.f};
        if (valid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = *(const f32x4*)(ab + 2 * e);        // 4 x 16 B, all in flight together
        }
        // pair-interleaved table {scale, scale, shift, shift}: pure renaming, nothing here waits for the loads
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[2 * e] = v[e][0]; a[2 * e + 1] = v[e][1]; c[2 * e] = v[e][2]; c[2 * e + 1] = v[e][3]; }
    }
    // scale = gamma * rstd, shift = beta - mean * scale from raw affine parameters and a group's statistics
    __device__ __forceinline__ void from_raw(const f32x4* g, const f32x4* bt, float mean, float rstd, bool valid) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = g[e >> 2][e & 3] * rstd;
            a[e] = valid ? sc : 1.f;
            c[e] = valid ? fmaf(-mean, sc, bt[e >> 2][e & 3]) : 0.f;
        }
    }
    template <bool SILU> __device__ __forceinline__ u32x4 apply(const u32x4& raw) const {
        const float nl2e = -1.4426950408889634f;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned u = raw[e];
            float y0 = fmaf(bf_lo(u), a[2 * e], c[2 * e]), y1 = fmaf(bf_hi(u), a[2 * e + 1], c[2 * e + 1]);
            if (SILU) {
                const float d0 = __builtin_amdgcn_exp2f(y0 * nl2e) + 1.0f, d1 = __builtin_amdgcn_exp2f(y1 * nl2e) + 1.0f;
                y0 *= __builtin_amdgcn_rcpf(d0); y1 *= __builtin_amdgcn_rcpf(d1);
            }
            o[e] = pack_bf2(y0, y1);
        }
        return o;
    }
};


// ---- GroupNorm partial sums: write-through publish + last-arriver finalize ---------------------------------------------
// Hand-off between workgroups of ONE launch, placement-independent (cdna_hip_programming.md Guideline 16, R1 with an
// arrival counter): every partial is ONE 8-byte agent-scope (sc1, write-through) store; every storing wave drains its
// stores; workgroup barrier; one lane adds to the sample's counter; the workgroup whose add returns fin_blocks-1 is the
// last one: one agent-scope acquire, barrier, sc1 loads of every partial of that sample in a fixed order.
__device__ __forceinline__ void part_store(float2* p, float s1, float s2)
{
    const unsigned long long v = ((unsigned long long)__float_as_uint(s2) << 32) | __float_as_uint(s1);
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 part_load(const float2* p)
{
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

// mean and 1/sqrt(var + eps) of group g of sample b from its partial sums; one wave per group, fixed summation order;
// every lane returns the result
__device__ __forceinline__ void gn_group_stats(const float2* part, int b, int g, int G, int nslot, int n_nt, int bn, int cpg,
                                               double count, float eps, int lane, double& mean, double& rstd)
{
    const int jlo = (g * cpg) / bn, jhi = ((g + 1) * cpg - 1) / bn, nj = jhi - jlo + 1;
    const int n_sp = nslot / n_nt, ne = n_sp * nj;
    const float2* base = part + (size_t)(b * G + g) * nslot;
    double s1 = 0.0, s2 = 0.0;
    for (int e = lane; e < ne; e += 64) {
        const int sp = e / nj, j = jlo + (e - sp * nj);
        const float2 v = part_load(base + (size_t)sp * n_nt + j);
        s1 += (double)v.x; s2 += (double)v.y;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { s1 += __shfl_xor(s1, s); s2 += __shfl_xor(s2, s); }
    mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    rstd = 1.0 / sqrt(var + (double)eps);
}

// scale/shift of every channel of group g of sample b from its partial sums
__device__ __forceinline__ void gn_reduce_group(const float2* part, int b, int g, int G, int nslot, int n_nt, int bn, int cpg, int C,
                                                double count, const float* gamma, const float* beta, float eps, float2* ab, int lane)
{
    double mean, rstd;
    gn_group_stats(part, b, g, G, nslot, n_nt, bn, cpg, count, eps, lane, mean, rstd);
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
        const double sc = (double)gamma[c] * rstd;
        // pair-interleaved: channels (2p, 2p+1) -> {scale, scale, shift, shift} (see GnCoef::load)
        float* const row = (float*)(ab + (size_t)b * C) + 4 * (c >> 1) + (c & 1);
        row[0] = (float)sc; row[2] = (float)((double)beta[c] - mean * sc);
    }
}

// call with ALL threads of the workgroup after its part_store()s; `flag` is a free LDS word
template <int NT>
__device__ __forceinline__ void gn_fused_finalize(const ConvArgs& a, int b, unsigned* flag, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.fin_counter + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = (old == (unsigned)a.fin_blocks - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int g = wave; g < a.G; g += NT / 64)
        gn_reduce_group(a.part, b, g, a.G, a.nslot, a.n_nt, a.bn, a.cpg, a.Cout, a.fin_count, a.fin_gamma, a.fin_beta, 1e-5f, a.fin_ab, lane);
    if (tid == 0) __hip_atomic_store(a.fin_counter + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}

}  // namespace ccn
