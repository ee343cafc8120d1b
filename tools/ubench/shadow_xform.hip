// Micro-benchmark for the persistent conv kernel's role split: WHO should run the GroupNorm + SiLU staging
// transform?  One workgroup per CU, 4 consumer waves (the kernel's own 120-step chunk loop on v_mfma_f32_32x32x16_bf16: 288 MFMAs, 120
// ds_read_b128, 36 weight loads from L2) + 4 producer waves, coupled by workgroup barriers like the kernel, on random operands.
//   MODE 0 (the kernel today): producers transform 11 items per chunk while staging (load -> GN+SiLU -> ds_write) and run the previous
//          tile's epilogue (8 items per chunk); consumers only MFMA.  ONE barrier per chunk.
//   MODE 1 (candidate): producers dump the items RAW (load -> ds_write, no VALU) and run the epilogue; each consumer wave transforms a
//          quarter of the next chunk IN PLACE in LDS in the shadow of its own MFMAs (per group of 24 MFMAs: ds_read_b128 of one item in
//          the one-MFMA step hh = 0, one element (unpack, fma, mul, exp2, add, rcp, mul, half a cvt_pk) in each of the steps hh = 1..8,
//          ds_write_b128 in hh = 9).  TWO barriers per chunk (raw visible; transformed visible).
//   MODE 2: MODE 1 without the producers' epilogue (what the consumers' loop alone can do);  MODE 3: MODE 0 without it.
// Reports consumer cycles per chunk (ideal 288 x 32 = 9216), wall per launch and the in-kernel clock.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 shadow_xform.hip -o shadow_xform && ./shadow_xform [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned rnd_pair(unsigned h, int ebias)
{
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned lo = ((h & 1u) << 15) | ((unsigned)(ebias + 125 + ((h >> 1) & 3u)) << 7) | ((h >> 3) & 0x7fu);
    const unsigned hi = (((h >> 10) & 1u) << 15) | ((unsigned)(ebias + 125 + ((h >> 11) & 3u)) << 7) | ((h >> 13) & 0x7fu);
    return lo | (hi << 16);
}
__device__ __forceinline__ unsigned pack_bf2(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
// one bf16 pair: GroupNorm affine + SiLU, scalar fp32 ops only (the kernel's GnCoef<bf16>::apply)
__device__ __forceinline__ unsigned xform_pair(unsigned u, float a0, float c0, float a1, float c1)
{
    const float nl2e = -1.4426950408889634f;
    float y0 = fmaf(__uint_as_float(u << 16), a0, c0), y1 = fmaf(__uint_as_float(u & 0xffff0000u), a1, c1);
    const float d0 = __builtin_amdgcn_exp2f(y0 * nl2e) + 1.0f, d1 = __builtin_amdgcn_exp2f(y1 * nl2e) + 1.0f;
    y0 *= __builtin_amdgcn_rcpf(d0); y1 *= __builtin_amdgcn_rcpf(d1);
    return pack_bf2(y0, y1);
}

constexpr int A_BYTES = 10 * 34 * 128;       // one 64-channel chunk of an 8-row halo tile
constexpr int STG = 8 * 32 * 272;            // bf16 staging tile
template <int MODE>
__global__ __launch_bounds__(512) void kx(const u32x4* __restrict__ w, const u32x4* __restrict__ gin, u32x4* __restrict__ gout, unsigned long long* out,
                                          float* sink, int chunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (2 * A_BYTES + STG) / 16; i += 512) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = rnd_pair((unsigned)(i * 4 + q) * 2654435761u, 0);
        ((u32x4*)smem)[i] = v;
    }
    __syncthreads();
    const int ptid = tid & 255, ck = ptid & 7, pcol = ptid >> 3;
    float ca[8], cc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ca[e] = 0.9f + 0.01f * (float)((ck * 8 + e) & 15); cc[e] = 0.05f * (float)((e + ck) & 7) - 0.1f; }
    const auto gsrd = __builtin_amdgcn_make_buffer_rsrc((void*)gin, 0, 1 << 26, 0x00020000);
    const auto osrd = __builtin_amdgcn_make_buffer_rsrc((void*)gout, 0, 1 << 26, 0x00020000);
    auto item_addr = [&](int buf, int i) __attribute__((always_inline)) { return buf * A_BYTES + (i * 34 + pcol) * 128 + (((ck ^ (pcol >> 1)) & 7) << 4); };

    if (wave >= 4) {
        // ---------------------------------------------------------------- producers
        u32x4 areg[11];
        const unsigned gbase = (unsigned)(((blockIdx.x * 256 + ptid) * 16) & ((1 << 22) - 1));
        auto issue = [&](int k) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 11; ++i) areg[i] = __builtin_amdgcn_raw_buffer_load_b128(gsrd, gbase + ((unsigned)((k * 11 + i) & 15) << 22), 0, 0);
        };
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        issue(0);
        for (int k = 0; k < chunks; ++k) {
            // stage chunk k+1 into the other buffer
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                u32x4 v = areg[i];
                if (MODE == 0 || MODE == 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = xform_pair(v[q], ca[2 * q], cc[2 * q], ca[2 * q + 1], cc[2 * q + 1]);
                }
                *(u32x4*)(smem + item_addr((k + 1) & 1, i % 10)) = v;
                if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 1 || MODE == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }   // raw chunk visible
            issue(k + 1);
            if (MODE == 0 || MODE == 1) {
                // half a tile's epilogue per chunk (2-chunk layers): 8 items of staging -> FiLM affine + residual -> store, statistics
                const int sb = 2 * A_BYTES + (ptid >> 4) * 272 + (ptid & 15) * 16;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const u32x4 sv = *(const u32x4*)(smem + sb + ((it >> 1) * 32 + (it & 1) * 16) * 272);
                    const u32x4 rv = __builtin_amdgcn_raw_buffer_load_b128(gsrd, gbase + (unsigned)(it << 18), 0, 0);
                    u32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x0 = fmaf(__uint_as_float(sv[q] << 16), ca[2 * q], cc[2 * q]) + __uint_as_float(rv[q] << 16);
                        const float x1 = fmaf(__uint_as_float(sv[q] & 0xffff0000u), ca[2 * q + 1], cc[2 * q + 1]) + __uint_as_float(rv[q] & 0xffff0000u);
                        s1[2 * q] += x0; s2[2 * q] = fmaf(x0, x0, s2[2 * q]); s1[2 * q + 1] += x1; s2[2 * q + 1] = fmaf(x1, x1, s2[2 * q + 1]);
                        o[q] = pack_bf2(x0, x1);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o, osrd, gbase + (unsigned)(it << 18), 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // end of chunk
        }
        float r = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) r += s1[e] + s2[e];
        sink[blockIdx.x * 512 + tid] = r + __uint_as_float(areg[0][0]);
        return;
    }
    // -------------------------------------------------------------------- consumers
    __builtin_amdgcn_s_setprio(2);
    constexpr int TH = 8, HR = 10, WIN = 6, NS = 120, PF = 4, HPITCH = 34;
    f32x16 acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = (f32x16){0};
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 24, 0x00020000);
    const int voff = ((blockIdx.x & 3) * 4096 + wave * 1024 + lane) * 16;
    const int r = lane & 31, h = lane >> 5;
    u32x4 bq[3][3], rw[WIN];
    int b16x[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { const int hx = r + d; b16x[d] = hx * 128 + (((hx >> 1) & 6) << 4) + (((h ^ (hx >> 1)) & 1) << 4); }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) bq[g][dy] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (g * 3 + dy) * 1024, 0));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < chunks; ++k) {
        const int bufoff = (k & 1) * A_BYTES, nbuf = (k + 1) & 1;
        const int sbase = (k & 8) * 1024;
        int ag = 0;
        auto rload = [&](int s_) __attribute__((always_inline)) {
            const int g = s_ / HR, hh = s_ % HR;
            if (hh == 0) ag = (bufoff + b16x[g / 4]) ^ ((g & 3) << 5);
            rw[s_ % WIN] = *(const u32x4*)(smem + ag + hh * HPITCH * 128);
        };
        u32x4 xr = {0, 0, 0, 0}, xo = {0, 0, 0, 0};
        float ya = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < PF; ++s_) rload(s_);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int g = s_ / HR, hh = s_ % HR;
            __builtin_amdgcn_sched_barrier(0);
            if ((MODE == 1 || MODE == 2) && s_ == 10) {              // after the first group: the producers' raw dump of chunk k+1 is complete
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            const int pg = g + 2;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
                if (hh == 2 + 2 * dy) bq[pg % 3][dy] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + ((pg % 6) * 3 + dy) * 1024, 0));
            if (s_ + PF < NS) rload(s_ + PF);
            // the in-shadow transform: group g (1..11) handles item g-1 of this thread's quarter of the next chunk
            int nv = 0;
            if ((MODE == 1 || MODE == 2) && g >= 1) {
                const int it = g - 1;
                if (hh == 0) xr = *(const u32x4*)(smem + item_addr(nbuf, it % 10));
                else if (hh <= 8) {
                    // one element per step: unpack, fma, mul, exp2, add, rcp, mul (+ the pair's cvt_pk in the odd step)
                    const int e = hh - 1, q = e >> 1;
                    const float nl2e = -1.4426950408889634f;
                    const float xin = (e & 1) ? __uint_as_float(xr[q] & 0xffff0000u) : __uint_as_float(xr[q] << 16);
                    float y = fmaf(xin, ca[e], cc[e]);
                    y *= __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y * nl2e) + 1.0f);
                    if (e & 1) xo[q] = pack_bf2(ya, y); else ya = y;
                    nv = 9;
                } else *(u32x4*)(smem + 2 * A_BYTES + (it * 256 + tid) * 16) = xo;   // (written beside the buffers: the MFMA operands stay random)
            }
            int nm = 0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int i = hh - dy;
                if (i >= 0 && i < TH) { acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[g % 3][dy]), __builtin_bit_cast(bf16x8, rw[s_ % WIN]), acc[i], 0, 0, 0); ++nm; }
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (m < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (m == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (m == 1 && (hh == 2 || hh == 4 || hh == 6)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (nv && nm == 3) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                else if (nv && nm == 2) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                else if (nv) __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
                else __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // end of chunk
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x == 0) { out[wave] = t1 - t0; out[4 + wave] = q1 - q0; }
    float rr = 0.f;
#pragma unroll
    for (int m = 0; m < TH; ++m) rr += acc[m][0] + acc[m][7];
    sink[blockIdx.x * 512 + tid] = rr;
}

template <int MODE> static void run(const char* name, const u32x4* w, const u32x4* gin, u32x4* gout, unsigned long long* out, float* sink, double seconds)
{
    const int chunks = 512, lds = 2 * A_BYTES + STG;
    (void)hipFuncSetAttribute((const void*)kx<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double total = 0; long n = 0;
    while (total < seconds * 1e3) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((kx<MODE>), dim3(256), dim3(512), lds, 0, w, gin, gout, out, sink, chunks);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); total += ms; n += 10;
    }
    unsigned long long h[8]; (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-64s %.3f ms per launch; %.0f cycles per chunk (ideal 9216: %.1f %% of the MFMA rate), clock %.3f GHz\n", name, total / n, (double)h[0] / chunks,
           100.0 * 9216 * chunks / (double)h[0], (double)h[0] / (double)h[4] * 0.1);
}

int main(int argc, char** argv)
{
    const double sec = argc > 1 ? atof(argv[1]) : 1.5;
    u32x4 *w, *gin, *gout; unsigned long long* out; float* sink;
    (void)hipMalloc(&w, 1 << 24); (void)hipMalloc(&gin, 1 << 26); (void)hipMalloc(&gout, 1 << 26); (void)hipMalloc(&out, 256); (void)hipMalloc(&sink, 256 * 512 * 4);
    std::vector<unsigned> hw((1 << 26) / 4);
    unsigned st = 12345u;
    for (auto& x : hw) { st = st * 1664525u + 1013904223u; x = rnd_pair(st ^ (st >> 13), -3); }
    (void)hipMemcpy(w, hw.data(), 1 << 24, hipMemcpyHostToDevice);
    (void)hipMemcpy(gin, hw.data(), 1 << 26, hipMemcpyHostToDevice);
    for (int rd = 0; rd < 3; ++rd) {
        run<0>("MODE 0  producers transform + epilogue (today)", w, gin, gout, out, sink, sec);
        run<1>("MODE 1  consumers transform in their MFMA shadow, producers raw + epilogue", w, gin, gout, out, sink, sec);
        run<3>("MODE 3  producers transform, no epilogue", w, gin, gout, out, sink, sec);
        run<2>("MODE 2  consumers transform in shadow, producers raw only", w, gin, gout, out, sink, sec);
    }
    return 0;
}
