// Micro-benchmark: the consumer wave's step of the persistent conv kernel in isolation.  One group = 8 MFMAs
// (v_mfma_f32_32x32x16_bf16, eight accumulators = a 256-pixel x 32-channel wave tile), NDS `ds_read_b128` A fragments and
// NLOAD 16-byte-per-lane weight loads from an L2-resident array through a three-deep register ring.  Question: what does a
// weight load cost the wave (DESIGN finding 10 fitted ~86 cycles), and does the kind of load or the tile shape change it?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 consumer_loop.hip -o consumer_loop && ./consumer_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned raw4;
// MT accumulators per wave (8: 256 x 32 tile; 16: 512 x 32), NLOAD weight loads per group of MT MFMAs, NDS ds_reads per group
template <int MT, int NLOAD, int NDS, int KIND>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ w, unsigned long long* out, float* sink, int iters, int wstride)
{
    __shared__ u32x4 lds[64 * 64];                                 // 64 KB of "A fragments"
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) { u32x4 v = {0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; lds[i] = v; }
    __syncthreads();
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x16){0};
    const u32x4* wp = w + (size_t)(blockIdx.x & 3) * 4096 + wave * 1024 + lane;          // a few hundred KB in all: L2-resident
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 24, 0x00020000);
    int voff = ((blockIdx.x & 3) * 4096 + wave * 1024 + lane) * 16;
    u32x4 ring[3][NLOAD > 0 ? NLOAD : 1];
    auto issue = [&](int slot, int it) __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NLOAD; ++l) {
            if (KIND == 0) ring[slot][l] = wp[((it * NLOAD + l) & 15) * 64];
            else           ring[slot][l] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((it * NLOAD + l) & 15) * 1024, 0));
        }
    };
    u32x4 bconst = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    for (int s = 0; s < 3; ++s) { if (s == 0) issue(0, 0); if (s == 1) issue(1, 1); if (s == 2) issue(2, 2); }
    // A fragments are fetched one group ahead (two register sets), as the kernel's row-fragment prefetch does
    u32x4 a[2][NDS > 0 ? NDS : 1];
    a[0][0] = bconst; a[1][0] = bconst;
    auto fetch_a = [&](int buf, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < NDS; ++m) a[buf][m] = lds[(g & 7) * 512 + (m & 7) * 64 + lane];
    };
    fetch_a(0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 6) {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            u32x4 b[NLOAD > 0 ? NLOAD : 1];
            b[0] = bconst;
            if (NLOAD > 0) {
                // wait for this slot (two younger groups stay in flight)
                if (NLOAD == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                if (NLOAD == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
                for (int l = 0; l < NLOAD; ++l) b[l] = ring[s % 3][l];
            }
            fetch_a((s + 1) & 1, it + s + 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int ai = NDS > 0 ? m % NDS : 0, bi = NLOAD > 1 ? m / (MT / NLOAD) : 0;
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s & 1][ai]), __builtin_bit_cast(bf16x8, b[bi]), acc[m], 0, 0, 0);
            }
            if (NLOAD > 0) issue(s % 3, it + s + 3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) r += acc[m][0] + acc[m][7];
    sink[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MT, int NLOAD, int NDS, int KIND>
static void run(const char* name, const u32x4* w, unsigned long long* out, float* sink, int grid)
{
    const int iters = 6 * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MT, NLOAD, NDS, KIND>), dim3(grid), dim3(256), 0, 0, w, out, sink, iters, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MT, NLOAD, NDS, KIND>), dim3(grid), dim3(256), 0, 0, w, out, sink, iters, 0);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / iters;                       // s_memtime ticks (100 MHz) -> use the event time for cycles
    const double us_per_group = ms * 1e3 / iters;
    printf("%-44s grid %3d  %.4f us/group  = %.1f cycles @2.0GHz per %d MFMAs (ideal %d)  [memtime %.3f ticks]\n", name, grid, us_per_group,
           us_per_group * 2000.0, MT, MT * 32, cyc);
}

// The persistent kernel's real group: HR = TH + 2 halo-row fragments (ds_read_b128), 3 weight fragments (one per dy), 3 * TH MFMAs
// (output row r takes halo rows r, r+1, r+2).  A fragments one group ahead, weights through a three-deep ring.
template <int TH, int NDSX, int NWL>
__global__ __launch_bounds__(256) void kg(const u32x4* __restrict__ w, unsigned long long* out, float* sink, int iters)
{
    __shared__ u32x4 lds[64 * 64];
    constexpr int HR = TH + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) { u32x4 v = {0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; lds[i] = v; }
    __syncthreads();
    f32x16 acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = (f32x16){0};
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 24, 0x00020000);
    const int voff = ((blockIdx.x & 3) * 4096 + wave * 1024 + lane) * 16;
    u32x4 ring[3][3];
    u32x4 bconst = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    auto issue = [&](int slot, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (l < NWL) ring[slot][l] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((g * 3 + l) & 15) * 1024, 0));
            else ring[slot][l] = bconst;
        }
    };
    u32x4 a[2][HR];
    auto fetch_a = [&](int buf, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < HR; ++m) { if (m < NDSX) a[buf][m] = lds[(g & 3) * 1024 + m * 64 + lane]; else a[buf][m] = bconst; }
    };
    issue(0, 0); issue(1, 1); issue(2, 2);
    fetch_a(0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 6) {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            if (NWL == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if (NWL == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            fetch_a((s + 1) & 1, it + s + 1);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int r = 0; r < TH; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s & 1][r + dy]), __builtin_bit_cast(bf16x8, ring[s % 3][dy]), acc[r], 0, 0, 0);
            if (NWL > 0) issue(s % 3, it + s + 3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < TH; ++m) r += acc[m][0] + acc[m][7];
    sink[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int TH, int NDSX, int NWL>
static void rung(const char* name, const u32x4* w, unsigned long long* out, float* sink, int grid)
{
    const int iters = 6 * 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kg<TH, NDSX, NWL>), dim3(grid), dim3(256), 0, 0, w, out, sink, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kg<TH, NDSX, NWL>), dim3(grid), dim3(256), 0, 0, w, out, sink, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s grid %3d  %.4f us/group  [%.1f cycles per %d MFMAs, ideal %d: %.1f %%]\n", name, grid, ms * 1e3 / iters, (double)h[0] / iters, 3 * TH, 96 * TH,
           100.0 * 96 * TH / ((double)h[0] / iters));
}

// The kernel's own step order: halo row hh feeds output rows hh, hh-1, hh-2 (dy = 0, 1, 2) -> an accumulator is reused after
// 2-3 MFMAs; row fragments PF rows ahead through a WIN-deep window, one weight load in each of three steps of a group.
template <int PF, int ORDER>
__global__ __launch_bounds__(256) void kk(const u32x4* __restrict__ w, unsigned long long* out, float* sink, int iters, int rnd = 0)
{
    __shared__ u32x4 lds[64 * 64];
    constexpr int TH = 8, HR = 10, WIN = 6, NS = 60;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        u32x4 v = {0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        if (rnd) {                                               // random bf16 pairs of magnitude ~0.25..2 (operand toggling like real activations)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h = (unsigned)(i * 4 + q) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                const unsigned lo = ((h & 1u) << 15) | ((125u + ((h >> 1) & 3u)) << 7) | ((h >> 3) & 0x7fu);
                const unsigned hi = (((h >> 10) & 1u) << 15) | ((125u + ((h >> 11) & 3u)) << 7) | ((h >> 13) & 0x7fu);
                v[q] = lo | (hi << 16);
            }
        }
        lds[i] = v;
    }
    __syncthreads();
    f32x16 acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = (f32x16){0};
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 24, 0x00020000);
    const int voff = ((blockIdx.x & 3) * 4096 + wave * 1024 + lane) * 16;
    u32x4 bq[3][3], rw[WIN];
    int sbase = 0;
    auto rload = [&](int s_) __attribute__((always_inline)) { const int g = (s_ / HR) % 6, hh = s_ % HR; rw[s_ % WIN] = lds[(g & 3) * 1024 + hh * 64 + lane]; };
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) bq[g][dy] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (g * 3 + dy) * 1024, 0));
#pragma unroll
    for (int s_ = 0; s_ < PF; ++s_) rload(s_);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 6) {
        sbase = (it & 8) * 1024;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int g = s_ / HR, hh = s_ % HR;
            __builtin_amdgcn_sched_barrier(0);
            const int pg = g + 2;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
                if (hh == 2 + 2 * dy) bq[pg % 3][dy] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + ((pg % 6) * 3 + dy) * 1024, 0));
            rload(s_ + PF);
            int nm = 0;
            if (ORDER == 0) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int i = hh - dy;
                    if (i >= 0 && i < TH) { acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[g % 3][dy]), __builtin_bit_cast(bf16x8, rw[s_ % WIN]), acc[i], 0, 0, 0); ++nm; }
                }
            } else {
#pragma unroll
                for (int dy = 2; dy >= 0; --dy) {
                    const int i = hh - dy;
                    if (i >= 0 && i < TH) { acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[g % 3][dy]), __builtin_bit_cast(bf16x8, rw[s_ % WIN]), acc[i], 0, 0, 0); ++nm; }
                }
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (m < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (m == 1 && (hh == 2 || hh == 4 || hh == 6)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < TH; ++m) r += acc[m][0] + acc[m][7];
    sink[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int PF, int ORDER>
static void runk(const char* name, const u32x4* w, unsigned long long* out, float* sink, int grid)
{
    const int iters = 6 * 512;
    hipLaunchKernelGGL((kk<PF, ORDER>), dim3(grid), dim3(256), 0, 0, w, out, sink, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((kk<PF, ORDER>), dim3(grid), dim3(256), 0, 0, w, out, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[4]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s grid %3d  [%.1f cycles per 24 MFMAs, ideal 768: %.1f %%]\n", name, grid, (double)h[0] / iters, 100.0 * 768 / ((double)h[0] / iters));
}

// The same wave tile (8 rows x 32 pixels x 32 channels), the same LDS and weight bytes, on the OTHER bf16 MFMA shape:
// v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md, DVFS give-back item 7; cdna_hip_programming.md rule 28).  The tile is cut into
// pixel halves p and channel halves c (acc[row][p][c], 4 registers each = the same 128 accumulator registers); one group is a
// (dx, 32-channel K slice) pair = 6 weight fragments (dy x c) + 20 row fragments (halo row x p) + 96 MFMAs of 16 cycles = TWO groups of
// the 32x32x16 form (2 x (3 + 10 + 24 x 32 cycles)).  Halo row fragment (hh, p) feeds output rows hh, hh-1, hh-2 x both channel halves the
// moment it arrives; weights through a ring of two groups (the next group's six loads go out in the six-MFMA steps of this one).
// F16 = 1: the f16 forms of both shapes (same cycles; what the operand type does to the clock).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// random operand bits, magnitude ~0.25..2 (bf16: 8-bit exponent 125..128; f16: 5-bit exponent 13..16), random sign and mantissa
__host__ __device__ inline unsigned rnd_pair(unsigned h, int f16, int ebias)
{
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    if (f16) {
        const unsigned lo = ((h & 1u) << 15) | ((unsigned)(ebias + 13 + ((h >> 1) & 3u)) << 10) | ((h >> 3) & 0x3ffu);
        const unsigned hi = (((h >> 13) & 1u) << 15) | ((unsigned)(ebias + 13 + ((h >> 14) & 3u)) << 10) | ((h >> 16) & 0x3ffu);
        return lo | (hi << 16);
    }
    const unsigned lo = ((h & 1u) << 15) | ((unsigned)(ebias + 125 + ((h >> 1) & 3u)) << 7) | ((h >> 3) & 0x7fu);
    const unsigned hi = (((h >> 10) & 1u) << 15) | ((unsigned)(ebias + 125 + ((h >> 11) & 3u)) << 7) | ((h >> 13) & 0x7fu);
    return lo | (hi << 16);
}
template <int F16> __device__ __forceinline__ void mfma32(f32x16& c, const u32x4& a, const u32x4& b)
{
    if (F16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int F16> __device__ __forceinline__ void mfma16s(f32x4v& c, const u32x4& a, const u32x4& b)
{
    if (F16) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// out[0..3] = s_memtime cycles of waves 0-3 of block 0, out[4..7] = s_memrealtime ticks (100 MHz): clock = cycles / ticks * 100 MHz
// SHAPE 32: the kernel's own loop (kk above) with the operand type a template parameter; SHAPE 16: the 16x16x32 form.
// `iters` counts 768-cycle units (24 MFMAs of 32x32x16 = 48 of 16x16x32) in both.
template <int SHAPE, int F16>
__global__ __launch_bounds__(256) void kshape(const u32x4* __restrict__ w, unsigned long long* out, float* sink, int iters, int rnd)
{
    __shared__ u32x4 lds[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        u32x4 v = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        if (F16) v = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        if (rnd) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = rnd_pair((unsigned)(i * 4 + q) * 2654435761u, F16, 0);
        }
        lds[i] = v;
    }
    __syncthreads();
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 24, 0x00020000);
    const int voff = ((blockIdx.x & 3) * 4096 + wave * 1024 + lane) * 16;
    constexpr int WIN = 6, PF = 4;
    u32x4 rw[WIN];
    int sbase = 0;
    float r = 0.f;
    unsigned long long t0 = 0, t1 = 0, q0 = 0, q1 = 0;
    if constexpr (SHAPE == 32) {
        constexpr int TH = 8, HR = 10, NS = 60;
        f32x16 acc[TH];
#pragma unroll
        for (int m = 0; m < TH; ++m) acc[m] = (f32x16){0};
        u32x4 bq[3][3];
        auto rload = [&](int s_) __attribute__((always_inline)) { const int g = (s_ / HR) % 6, hh = s_ % HR; rw[s_ % WIN] = lds[(g & 3) * 1024 + hh * 64 + lane]; };
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) bq[g][dy] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (g * 3 + dy) * 1024, 0));
#pragma unroll
        for (int s_ = 0; s_ < PF; ++s_) rload(s_);
        t0 = __builtin_amdgcn_s_memtime(); q0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; it += 6) {
            sbase = (it & 8) * 1024;
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                const int g = s_ / HR, hh = s_ % HR;
                __builtin_amdgcn_sched_barrier(0);
                const int pg = g + 2;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
                    if (hh == 2 + 2 * dy) bq[pg % 3][dy] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + ((pg % 6) * 3 + dy) * 1024, 0));
                rload(s_ + PF);
                int nm = 0;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int i = hh - dy;
                    if (i >= 0 && i < TH) { mfma32<F16>(acc[i], bq[g % 3][dy], rw[s_ % WIN]); ++nm; }
                }
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    if (m < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (m == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (m == 1 && (hh == 2 || hh == 4 || hh == 6)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime(); q1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int m = 0; m < TH; ++m) r += acc[m][0] + acc[m][7];
    } else {
        constexpr int TH = 8, HR = 10, NS = 40;                   // two groups of 20 (halo row, pixel half) steps = 4 units of 768 cycles
        // one 16-register tuple per output row, the MFMAs work on its quarters (p, c): 32 separate 4-register tuples make the register
        // allocator permute them over the loop's back edge (~190 v_accvgpr copies per iteration)
        f32x16 acc[TH];
#pragma unroll
        for (int m = 0; m < TH; ++m) acc[m] = (f32x16){0};
        u32x4 bq[2][3][2];
        auto rload = [&](int s_) __attribute__((always_inline)) { const int g = (s_ / 20) & 1, hp = s_ % 20; rw[s_ % WIN] = lds[g * 2048 + hp * 64 + lane]; };
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int c = 0; c < 2; ++c) bq[0][dy][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (dy * 2 + c) * 1024, 0));
#pragma unroll
        for (int s_ = 0; s_ < PF; ++s_) rload(s_);
        t0 = __builtin_amdgcn_s_memtime(); q0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; it += 4) {
            sbase = (it & 8) * 1024;
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                const int g = s_ / 20, hh = (s_ % 20) >> 1, p = s_ & 1;
                __builtin_amdgcn_sched_barrier(0);
                // the next group's fragment (dy, c) = hh - 2 goes out in the six-MFMA step (hh, p = 0), hh = 2..7
                const bool wl = p == 0 && hh >= 2 && hh <= 7;
                if (wl) {
                    const int f = hh - 2;
                    bq[(g + 1) & 1][f >> 1][f & 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + ((((it >> 1) + g + 1) % 3) * 6 + f) * 1024, 0));
                }
                rload(s_ + PF);
                int nm = 0;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int i = hh - dy;
                    if (i >= 0 && i < TH) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const int q = p * 2 + c;
                            f32x4v t = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                            mfma16s<F16>(t, bq[g & 1][dy][c], rw[s_ % WIN]);
                            acc[i][4 * q] = t[0]; acc[i][4 * q + 1] = t[1]; acc[i][4 * q + 2] = t[2]; acc[i][4 * q + 3] = t[3];
                            ++nm;
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    if (m < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (m == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (m == 2 && wl) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (m == 1 || m == 3 || m == 5) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime(); q1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int m = 0; m < TH; ++m) r += acc[m][0] + acc[m][7] + acc[m][9] + acc[m][14];
    }
    if (lane == 0 && blockIdx.x == 0) { out[wave] = t1 - t0; out[4 + wave] = q1 - q0; }
    sink[blockIdx.x * 256 + threadIdx.x] = r;
}

// Co-execution: waves 0-3 run the kernel-order consumer loop (kk), waves 4-7 (one per SIMD, like the producers) issue VALU
// work until the consumers finish: VPI VALU instructions, then a sleep of GAP*64 cycles, repeated.  How many VALU instructions
// fit next to a group of 24 MFMAs, and what does each cost the MFMA stream?
template <int KIND, int PRIO, int NVS = 0>
__global__ __launch_bounds__(512) void kco(const u32x4* __restrict__ w, unsigned long long* out, float* sink, int iters, int gap)
{
    __shared__ u32x4 lds[64 * 64];
    __shared__ volatile int done;
    constexpr int TH = 8, HR = 10, WIN = 6, NS = 60, PF = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 512) { u32x4 v = {0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; lds[i] = v; }
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave >= 4) {
        float a0 = 1.0f + lane * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
        unsigned long long n = 0;
        while (!done) {
            if (KIND == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else {
                // the GroupNorm + SiLU mix per element: shift, fma, mul, exp, add, rcp, mul (+ half a cvt_pk)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_lshlrev_b32 %0, 16, %1\n v_fma_f32 %0, %0, %2, %3\n v_mul_f32 %1, 0xbfb8aa3b, %0\n v_exp_f32 %1, %1\n v_add_f32 %1, 1.0, %1\n v_rcp_f32 %1, %1\n v_mul_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            }
            n += 64;
            for (int q = 0; q < gap; ++q) __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) out[8 + wave] = n;
        sink[threadIdx.x] = a0 + a1 + a2 + a3;
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    f32x16 acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = (f32x16){0};
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 24, 0x00020000);
    const int voff = ((blockIdx.x & 3) * 4096 + wave * 1024 + lane) * 16;
    u32x4 bq[3][3], rw[WIN];
    float ov[4] = {1.0f + lane, 2.0f, 3.0f, 4.0f};
    int sbase = 0;
    auto rload = [&](int s_) __attribute__((always_inline)) { const int g = (s_ / HR) % 6, hh = s_ % HR; rw[s_ % WIN] = lds[(g & 3) * 1024 + hh * 64 + lane]; };
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) bq[g][dy] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (g * 3 + dy) * 1024, 0));
#pragma unroll
    for (int s_ = 0; s_ < PF; ++s_) rload(s_);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 6) {
        sbase = (it & 8) * 1024;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int g = s_ / HR, hh = s_ % HR;
            __builtin_amdgcn_sched_barrier(0);
            const int pg = g + 2;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
                if (hh == 2 + 2 * dy) bq[pg % 3][dy] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + ((pg % 6) * 3 + dy) * 1024, 0));
            rload(s_ + PF);
            int nm = 0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int i = hh - dy;
                if (i >= 0 && i < TH) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[g % 3][dy]), __builtin_bit_cast(bf16x8, rw[s_ % WIN]), acc[i], 0, 0, 0); ++nm;
#pragma unroll
                    for (int q = 0; q < NVS; ++q) { ov[q & 3] = __builtin_fmaf(ov[q & 3], ov[(q + 1) & 3], 1.0f); }
                }
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (m < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (m == 1 && (hh == 2 || hh == 4 || hh == 6)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (m < nm) __builtin_amdgcn_sched_group_barrier(0x002, NVS + 1, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[wave] = t1 - t0; sink[0] = ov[0] + ov[1] + ov[2] + ov[3]; }
    __builtin_amdgcn_s_setprio(0);
    if (threadIdx.x == 0) done = 1;
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < TH; ++m) r += acc[m][0] + acc[m][7];
    sink[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int KIND, int PRIO, int NVS = 0>
static void runco(const char* name, const u32x4* w, unsigned long long* out, float* sink, int gap)
{
    const int iters = 6 * 512;
    hipMemset(out, 0, 128);
    hipLaunchKernelGGL((kco<KIND, PRIO, NVS>), dim3(1), dim3(512), 0, 0, w, out, sink, iters, gap);
    hipDeviceSynchronize();
    unsigned long long h[16]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / iters, nv = (double)h[12] / iters;
    printf("own %d/MFMA, %-30s sleep %2d: %.1f cycles per group (%.1f %% of MFMA rate), %.1f VALU instr per group next to it (%.2f per MFMA); extra MFMA-wave cycles per VALU instr %.2f\n",
           NVS, name, gap, cyc, 100.0 * 768 / cyc, nv, nv / 24, nv > 0 ? (cyc - 771.0) / nv : 0.0);
}

// The consumer wave issuing VALU work itself in the shadow of its own MFMAs: NV v_fma_f32 after every MFMA (asm volatile keeps
// program order).  No other wave on the SIMD.
template <int NV>
__global__ __launch_bounds__(256) void kself(unsigned long long* out, float* sink, int iters, int rnd = 0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = (f32x16){0};
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    if (rnd) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h = (unsigned)(threadIdx.x * 8 + q) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            a[q] = ((h & 1u) << 15) | ((125u + ((h >> 1) & 3u)) << 7) | ((h >> 3) & 0x7fu) | ((((h >> 10) & 1u) << 15 | ((125u + ((h >> 11) & 3u)) << 7) | ((h >> 13) & 0x7fu)) << 16);
            h = h * 1664525u + 1013904223u;
            b[q] = ((h & 1u) << 15) | ((122u + ((h >> 1) & 3u)) << 7) | ((h >> 3) & 0x7fu) | ((((h >> 10) & 1u) << 15 | ((122u + ((h >> 11) & 3u)) << 7) | ((h >> 13) & 0x7fu)) << 16);
        }
    }
    float v0 = 1.0f + lane, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if ((q & 3) == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v0));
                if ((q & 3) == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v1));
                if ((q & 3) == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v2));
                if ((q & 3) == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v3));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    float r = v0 + v1 + v2 + v3;
#pragma unroll
    for (int m = 0; m < 8; ++m) r += acc[m][0] + acc[m][7];
    sink[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NV>
static void runself(unsigned long long* out, float* sink)
{
    const int iters = 2048;
    hipLaunchKernelGGL((kself<NV>), dim3(1), dim3(256), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[4]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("own wave: %d v_fma after each MFMA: %.1f cycles per 24 MFMAs (ideal 768: %.1f %%)\n", NV, (double)h[0] / iters, 100.0 * 768 / ((double)h[0] / iters));
}

// power mode: `consumer_loop <variant> [seconds]` keeps one variant running on all CUs (tools/power_probe_ubench.sh samples rocm-smi beside it)
template <typename F> static void spin(F launch, double seconds)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double total = 0; long n = 0;
    while (total < seconds * 1e3) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); total += ms; n += 20;
    }
    printf("ran %ld launches in %.1f ms (%.3f ms each)\n", n, total, total / n);
}

// Symmetric alternative to warp specialisation: WPS waves per SIMD, EVERY wave issues MFMAs and NV independent VALU instructions after each
// of them (its share of staging / epilogue work in the shadow of its own MFMAs).  Reports cycles per MFMA issued on a SIMD.
template <int NV, int WPS>
__global__ __launch_bounds__(256 * WPS) void ksym(unsigned long long* out, float* sink, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = (f32x16){0};
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float v0 = 1.0f + lane, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if ((q & 3) == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v0));
                if ((q & 3) == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v1));
                if ((q & 3) == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v2));
                if ((q & 3) == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v3));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;          // (t0 was taken behind a workgroup barrier: common start)
    float r = v0 + v1 + v2 + v3;
#pragma unroll
    for (int m = 0; m < 8; ++m) r += acc[m][0] + acc[m][7];
    sink[blockIdx.x * 256 * WPS + threadIdx.x] = r;
}
template <int NV, int WPS>
static void runsym(unsigned long long* out, float* sink)
{
    const int iters = 1024;
    hipLaunchKernelGGL((ksym<NV, WPS>), dim3(1), dim3(256 * WPS), 0, 0, out, sink, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int i = 0; i < 4 * WPS; ++i) mx = h[i] > mx ? h[i] : mx;    // the SIMD's last wave (the older wave wins the MFMA issue)
    const double cyc = (double)mx / (iters * 24.0) / WPS;          // cycles per MFMA issued on one SIMD
    printf("symmetric: %d wave(s) per SIMD, %d v_fma after each MFMA: %.1f cycles per MFMA on the SIMD (%.1f %% of the MFMA rate), %d VALU per MFMA\n", WPS, NV, cyc, 3200.0 / cyc, NV);
}

static void fill_w(u32x4* w, int f16)
{
    std::vector<unsigned> hw((1 << 24) / 4);
    unsigned st = 12345u;
    for (auto& x : hw) { st = st * 1664525u + 1013904223u; x = rnd_pair(st ^ (st >> 13), f16, -3); }
    (void)hipMemcpy(w, hw.data(), 1 << 24, hipMemcpyHostToDevice);
}
static void report_clock(unsigned long long* out, int iters)
{
    unsigned long long h[8]; (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("    in-kernel: %.1f cycles per 768-cycle unit (%.1f %% of the MFMA rate), clock %.3f GHz\n", (double)h[0] / iters, 100.0 * 768 * iters / (double)h[0],
           (double)h[0] / (double)h[4] * 0.1);
}

int main(int argc, char** argv)
{
    if (argc > 1) {
        u32x4* w; unsigned long long* out; float* sink;
        (void)hipMalloc(&w, 1 << 24); (void)hipMemset(w, 0x3f, 1 << 24); (void)hipMalloc(&out, 256); (void)hipMalloc(&sink, 256 * 1024 * 4);
        const double sec = argc > 2 ? atof(argv[2]) : 4.0;
        const int iters = 6 * 2048;
        const std::string v = argv[1];
        if (v == "mfma") spin([&] { hipLaunchKernelGGL((kself<0>), dim3(256), dim3(256), 0, 0, out, sink, iters / 3, 0); }, sec);
        else if (v == "mfma_rnd") spin([&] { hipLaunchKernelGGL((kself<0>), dim3(256), dim3(256), 0, 0, out, sink, iters / 3, 1); }, sec);
        else if (v == "sym2") spin([&] { hipLaunchKernelGGL((ksym<0, 2>), dim3(256), dim3(512), 0, 0, out, sink, iters / 3); }, sec);
        else if (v == "sym1") spin([&] { hipLaunchKernelGGL((ksym<0, 1>), dim3(256), dim3(256), 0, 0, out, sink, iters / 3); }, sec);
        else if (v == "mfma_valu4") spin([&] { hipLaunchKernelGGL((kself<4>), dim3(256), dim3(256), 0, 0, out, sink, iters / 3); }, sec);
        else if (v == "loop") spin([&] { hipLaunchKernelGGL((kk<4, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, 0); }, sec);
        else if (v == "loop_rnd") {
            std::vector<unsigned> hw((1 << 24) / 4);
            unsigned st = 12345u;
            for (auto& x : hw) {
                st = st * 1664525u + 1013904223u; const unsigned h = st ^ (st >> 13);
                const unsigned lo = ((h & 1u) << 15) | ((122u + ((h >> 1) & 3u)) << 7) | ((h >> 3) & 0x7fu);
                const unsigned hi = (((h >> 10) & 1u) << 15) | ((122u + ((h >> 11) & 3u)) << 7) | ((h >> 13) & 0x7fu);
                x = lo | (hi << 16);
            }
            (void)hipMemcpy(w, hw.data(), 1 << 24, hipMemcpyHostToDevice);
            spin([&] { hipLaunchKernelGGL((kk<4, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, 1); }, sec);
        }
        else if (v == "loop_lds_only") spin([&] { hipLaunchKernelGGL((kg<8, 10, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters); }, sec);
        else if (v == "loop_w_only") spin([&] { hipLaunchKernelGGL((kg<8, 0, 3>), dim3(256), dim3(256), 0, 0, w, out, sink, iters); }, sec);
        else if (v == "loop_sibling") spin([&] { hipLaunchKernelGGL((kco<1, 0, 0>), dim3(256), dim3(512), 0, 0, w, out, sink, iters, 0); }, sec);
        else if (v == "loop_rnd16" || v == "loop16" || v == "loop_rnd_f16" || v == "loop_rnd16_f16" || v == "loop32") {
            // one shape / operand type on all CUs (power_probe_ubench.sh samples rocm-smi beside it)
            const int f16 = v.find("f16") != std::string::npos, rnd = v.find("rnd") != std::string::npos, s16 = v.find("16") != std::string::npos && v != "loop_rnd_f16";
            if (rnd) fill_w(w, f16);
            auto launch = [&] {
                if (s16 && f16) hipLaunchKernelGGL((kshape<16, 1>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, rnd);
                else if (s16) hipLaunchKernelGGL((kshape<16, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, rnd);
                else if (f16) hipLaunchKernelGGL((kshape<32, 1>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, rnd);
                else hipLaunchKernelGGL((kshape<32, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, rnd);
            };
            spin(launch, sec);
            report_clock(out, iters);
        }
        else if (v == "shape_ab") {
            // rule 24: the variants interleaved in ONE process on one device, several rounds; random operands (rule 25); cycles AND wall AND clock
            const char* names[4] = {"32x32x16 bf16", "16x16x32 bf16", "32x32x16 f16 ", "16x16x32 f16 "};
            const int rounds = argc > 3 ? atoi(argv[3]) : 4;
            fill_w(w, 0);
            spin([&] { hipLaunchKernelGGL((kshape<32, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, 1); }, 2.0);   // warm: clocks settle under load
            for (int rd = 0; rd < rounds; ++rd)
                for (int k = 0; k < 4; ++k) {
                    fill_w(w, k >> 1);
                    printf("round %d  %s  ", rd, names[k]);
                    auto launch = [&] {
                        if (k == 0) hipLaunchKernelGGL((kshape<32, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, 1);
                        if (k == 1) hipLaunchKernelGGL((kshape<16, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, 1);
                        if (k == 2) hipLaunchKernelGGL((kshape<32, 1>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, 1);
                        if (k == 3) hipLaunchKernelGGL((kshape<16, 1>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, 1);
                    };
                    spin(launch, sec);
                    report_clock(out, iters);
                }
            // zero-toggle control: constants rank the shapes by cycles only (give-back item 7)
            for (int k = 0; k < 2; ++k) {
                printf("constants %s  ", names[k]);
                spin([&] { if (k == 0) hipLaunchKernelGGL((kshape<32, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, 0);
                           else hipLaunchKernelGGL((kshape<16, 0>), dim3(256), dim3(256), 0, 0, w, out, sink, iters, 0); }, sec);
                report_clock(out, iters);
            }
        }
        else { printf("unknown variant\n"); return 1; }
        return 0;
    }
    {
    u32x4* w; unsigned long long* out; float* sink;
    hipMalloc(&w, 1 << 24); hipMemset(w, 0x3f, 1 << 24); hipMalloc(&out, 256); hipMalloc(&sink, 256 * 1024 * 4);
    for (int grid : {1, 256}) {
        run<8, 0, 0, 0>("8 MFMA", w, out, sink, grid);
        run<8, 0, 8, 0>("8 MFMA + 8 ds_read", w, out, sink, grid);
        run<8, 1, 0, 0>("8 MFMA + 1 global_load", w, out, sink, grid);
        run<8, 1, 0, 1>("8 MFMA + 1 buffer_load", w, out, sink, grid);
        run<8, 1, 8, 0>("8 MFMA + 8 ds_read + 1 global_load", w, out, sink, grid);
        run<8, 1, 8, 1>("8 MFMA + 8 ds_read + 1 buffer_load", w, out, sink, grid);
        run<8, 2, 8, 1>("8 MFMA + 8 ds_read + 2 buffer_load", w, out, sink, grid);
        run<16, 1, 16, 1>("16 MFMA + 16 ds_read + 1 buffer_load", w, out, sink, grid);
        run<16, 2, 8, 1>("16 MFMA + 8 ds_read + 2 buffer_load", w, out, sink, grid);
        run<16, 0, 16, 1>("16 MFMA + 16 ds_read", w, out, sink, grid);
    }
    for (int grid : {1, 256}) {
        rung<8, 0, 0>("group 8 rows: 24 MFMA", w, out, sink, grid);
        rung<8, 10, 0>("group 8 rows: 24 MFMA + 10 ds", w, out, sink, grid);
        rung<8, 0, 3>("group 8 rows: 24 MFMA + 3 wload", w, out, sink, grid);
        rung<8, 10, 3>("group 8 rows: 24 MFMA + 10 ds + 3 wload", w, out, sink, grid);
        rung<8, 10, 1>("group 8 rows: 24 MFMA + 10 ds + 1 wload", w, out, sink, grid);
        rung<4, 6, 3>("group 4 rows: 12 MFMA + 6 ds + 3 wload", w, out, sink, grid);
        rung<4, 6, 0>("group 4 rows: 12 MFMA + 6 ds", w, out, sink, grid);
    }
    for (int grid : {1, 256}) {
        runk<4, 0>("kernel order, PF 4", w, out, sink, grid);
        runk<5, 0>("kernel order, PF 5", w, out, sink, grid);
        runk<3, 0>("kernel order, PF 3", w, out, sink, grid);
        runk<4, 1>("kernel order reversed dy, PF 4", w, out, sink, grid);
    }
    for (int gap : {0, 4, 8, 16, 64}) {
        runco<1, 0, 0>("GN+SiLU mix sibling", w, out, sink, gap);
        runco<1, 0, 1>("GN+SiLU mix sibling", w, out, sink, gap);
        runco<1, 0, 2>("GN+SiLU mix sibling", w, out, sink, gap);
        runco<1, 0, 3>("GN+SiLU mix sibling", w, out, sink, gap);
        runco<1, 0, 4>("GN+SiLU mix sibling", w, out, sink, gap);
    }
    runsym<0, 2>(out, sink); runsym<2, 2>(out, sink); runsym<3, 2>(out, sink); runsym<4, 2>(out, sink); runsym<5, 2>(out, sink); runsym<6, 2>(out, sink); runsym<7, 2>(out, sink); runsym<8, 2>(out, sink);
    runself<0>(out, sink); runself<1>(out, sink); runself<2>(out, sink); runself<3>(out, sink); runself<4>(out, sink);
    runself<5>(out, sink); runself<6>(out, sink); runself<7>(out, sink); runself<8>(out, sink);
    return 0;
    }
}
