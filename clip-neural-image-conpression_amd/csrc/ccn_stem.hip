// Stem convolution (models/unet.py:55,88: Conv2d(img_ch, base, 3, padding=1) on the NCHW fp32 image) for the bf16 mode.
//
// K = img_ch*9 = 27 is one MFMA K-block (32): the contraction is 2 x v_mfma_f32_32x32x16_bf16 per 32 pixels x 32 channels
// and costs nothing; the layer is bound by the 128 x H x W NHWC tensor it WRITES (134 MB at C2 / batch 8) and by the VALU
// work around it.  The generic implicit-GEMM kernel staged an im2col tile through LDS and transposed the result through
// LDS again (112 us, 1.2 TB/s).  Here only the output transpose touches LDS:
//   * the weights (A operand, [channel][k]) sit in registers for the whole kernel, host-packed in fragment order, with the
//     bias folded in as k = 27 against a constant-one im2col element;
//   * a lane builds its im2col fragment (B operand: pixel r, k = 16s + 8h .. +7) with 16 masked 4-byte loads straight from
//     the image (6 MB, L1/L2 resident); the per-lane k -> (channel, dy, dx) table is computed once;
//   * with the operands in this order a lane's accumulator quad is 4 consecutive channels of ONE pixel; the tile goes
//     through a per-wave LDS strip so that the NHWC stores are whole 256-byte pixel rows, and the next GroupNorm's
//     statistics accumulate per lane over the wave's pixels and are reduced once.
#include "ccn_device.h"
#include <cstdlib>

namespace ccn {

// G16: GroupNorm groups of the output are multiples of 16 channels -> statistics kept per 16-channel pair of register quads (half the
// registers); the kernel then fits three waves per SIMD, which is what hides its load -> MFMA -> LDS -> store latency chain
template <int NT, bool G16>
__global__ __launch_bounds__(256, (G16 || NT < 4) ? 3 : 2) void stem_kernel(const ConvArgs a, const int upw)
{
    constexpr int TRP = NT * 64 + 16;                              // strip pitch in bytes (C bf16 + pad: conflict-free 8-byte writes)
    __shared__ __attribute__((aligned(16))) unsigned char tr[4 * 32 * TRP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int H = a.Hin, W = a.Win, C = a.Cout, HW = H * W;
    const int n_tx = (W + 31) >> 5, units = H * n_tx;
    const int K = a.Cin * 9;                                        // <= 31 (checked by the launcher): k = K is the bias column

    u32x4 wf[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) wf[j][s] = *(const u32x4*)((const unsigned char*)a.wfrag + (size_t)((j * 2 + s) * 64 + lane) * 16);

    // element e (0..15) of this lane: k = 16*(e>>3) + 8h + (e&7) -> input channel c, tap (dy, dx).  The byte offset of (c, dy, dx)
    // depends on the lane only through h: both variants are wave-uniform (SGPRs), the lane selects at the load (16 VGPRs less)
    auto koff = [&](int e) __attribute__((always_inline)) -> int {
        int o[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int k = 16 * (e >> 3) + 8 * hh + (e & 7);
            const int c = k / 9, rem = k - c * 9, dy = rem / 3 - 1, dx = rem - (rem / 3) * 3 - 1;
            o[hh] = __builtin_amdgcn_readfirstlane((c * HW + dy * W + dx) * 4);
        }
        return h ? o[1] : o[0];
    };
    unsigned m_k = 0, m_one = 0, m_dy0 = 0, m_dy2 = 0, m_dx0 = 0, m_dx2 = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = 16 * (e >> 3) + 8 * h + (e & 7);
        const int c = k / 9, rem = k - c * 9, dy = rem / 3 - 1, dx = rem - (rem / 3) * 3 - 1;
        if (k < K) m_k |= 1u << e;
        if (k == K) m_one |= 1u << e;
        if (dy < 0) m_dy0 |= 1u << e;
        if (dy > 0) m_dy2 |= 1u << e;
        if (dx < 0) m_dx0 |= 1u << e;
        if (dx > 0) m_dx2 |= 1u << e;
    }
    constexpr unsigned OOB = 0x7FFFFFF0u;
    const unsigned in_bytes = (unsigned)((size_t)a.B * a.Cin * HW * 4);
    const auto isrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, in_bytes, 0x00020000);
    const unsigned out_bytes = (unsigned)((size_t)a.B * HW * C * 2);
    const auto osrd = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, out_bytes, 0x00020000);

    constexpr int NS = G16 ? 2 : 4;                                 // statistics slots per n-tile: quads, or pairs of quads (16 channels)
    float s1[NT][NS], s2[NT][NS];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < NS; ++g) { s1[j][g] = 0.f; s2[j][g] = 0.f; }

    const int u0 = (blockIdx.x * 4 + wave) * upw;
    // software pipeline: the 16 loads of unit uu+1 are in flight while unit uu is multiplied and stored
    float vn[16];
    auto fetch = [&](int u, float* dst) __attribute__((always_inline)) {
        const bool uv = u < units;                                  // wave-uniform
        const int y = (uv ? u : 0) / n_tx, x = ((uv ? u : 0) - y * n_tx) * 32 + r;
        // elements this lane must not read: outside its k range, or a tap that leaves the image
        unsigned bad = ~m_k;
        if (y == 0) bad |= m_dy0;
        if (y == H - 1) bad |= m_dy2;
        if (x == 0) bad |= m_dx0;
        if (x >= W - 1) bad |= m_dx2;
        if (x >= W || !uv) bad = 0xffffu;
        const int base = ((b * a.Cin) * HW + y * W + x) * 4;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            dst[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(isrd, (unsigned)(base + koff(e)) | (((bad >> e) & 1u) ? OOB : 0u), 0, 0));
    };
    fetch(u0, vn);
    for (int uu = 0; uu < upw; ++uu) {
        const int u = u0 + uu;
        if (u >= units) break;                                      // wave-uniform
        const int y = u / n_tx, x0 = (u - y * n_tx) * 32, x = x0 + r;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = vn[e];
        if (uu + 1 < upw) fetch(u + 1, vn);
        u32x4 bf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int e = s * 8 + 2 * p;
                const float lo = ((m_one >> e) & 1u) ? 1.0f : v[e], hi = ((m_one >> (e + 1)) & 1u) ? 1.0f : v[e + 1];
                bf[s][p] = pack_bf2(lo, hi);
            }
        const bool pv = x < W;
        const float mk = pv ? 1.0f : 0.0f;
        // accumulators -> the wave's LDS strip [32 pixels][C bf16 (+16 B pad)] -> 16-byte stores, 256 contiguous bytes per pixel:
        // direct 8-byte stores from the accumulator layout reached only 2.6 TB/s (32 scattered 16-byte pieces per instruction).
        // One N tile at a time: its accumulator is dead once packed (a quarter of the registers of NT live accumulators).
        unsigned char* const strip = tr + wave * (32 * TRP);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[j][s]), __builtin_bit_cast(bf16x8, bf[s]), acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float c0 = acc[g * 4], c1 = acc[g * 4 + 1], c2 = acc[g * 4 + 2], c3 = acc[g * 4 + 3];
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 pk = {pack_bf2(c0, c1), pack_bf2(c2, c3)};
                *(u32x2*)(strip + r * TRP + (j * 32 + g * 8 + 4 * h) * 2) = pk;
                const float t = (c0 + c1) + (c2 + c3);
                constexpr int SH = G16 ? 1 : 0;
                s1[j][g >> SH] = fmaf(t, mk, s1[j][g >> SH]);
                s2[j][g >> SH] = fmaf(fmaf(c0, c0, c1 * c1) + fmaf(c2, c2, c3 * c3), mk, s2[j][g >> SH]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // same wave: LDS is in order
        constexpr int SL = NT * 4;                                     // 16-byte slices per pixel
#pragma unroll
        for (int it = 0; it < (32 * SL) / 64; ++it) {
            const int idx = it * 64 + lane, pxl = idx / SL, sl = idx - pxl * SL;
            const u32x4 v16 = *(const u32x4*)(strip + pxl * TRP + sl * 16);
            const bool ok = x0 + pxl < W && sl * 8 < C;
            __builtin_amdgcn_raw_buffer_store_b128(v16, osrd, ok ? (unsigned)(((b * H + y) * W + x0 + pxl) * C + sl * 8) * 2u : OOB, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // strip read before the next unit rewrites it
    }
    if (!a.part) return;
    // wave reduction over the 32 pixel lanes of each half, then quads -> GroupNorm groups through a few LDS words
    __shared__ float red[4][NT * 8][2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // (G16: the pair's sum sits in its first quad's slot, the second quad's slot holds zero)
            float t1 = G16 ? ((g & 1) ? 0.f : s1[j][g >> 1]) : s1[j][g], t2 = G16 ? ((g & 1) ? 0.f : s2[j][g >> 1]) : s2[j][g];
#pragma unroll
            for (int s = 1; s < 32; s <<= 1) { t1 += __shfl_xor(t1, s); t2 += __shfl_xor(t2, s); }
            if (r == 0) { red[wave][j * 8 + g * 2 + h][0] = t1; red[wave][j * 8 + g * 2 + h][1] = t2; }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // same wave wrote them: LDS is in order per wave
    if (lane < a.G) {
        const int qpg = a.cpg / 4;                                  // channel quads per group (cpg % 4 == 0, checked by the launcher)
        float t1 = 0.f, t2 = 0.f;
        for (int q = lane * qpg; q < (lane + 1) * qpg && q < NT * 8; ++q) { t1 += red[wave][q][0]; t2 += red[wave][q][1]; }
        part_store(a.part + (size_t)(b * a.G + lane) * a.nslot + blockIdx.x * 4 + wave, t1, t2);
    }
}

bool stem2_supported(int dtype, int cin, int cout, int G)
{
    const int g = cout < G ? cout : G;
    return dtype == 1 && cin * 9 <= 31 && (cout == 32 || cout == 64 || cout == 128 || cout == 192) && cout % g == 0 && (cout / g) % 4 == 0;
}

// blocks per image for the chosen units-per-wave; nslot of the output's GroupNorm partials = 4 * that
int stem2_blocks(int H, int W, int* upw_out)
{
    const int units = H * ((W + 31) / 32);
    int upw = 8;                                                   // (sweep at C2, HIP events: 48.6 / 39.0 / 35.8 / 31.2 / 36.6 us for 1 / 2 / 4 / 8 / 16)
    if (units < 4 * 8 * 16) upw = 4;
    if (units < 4 * 4 * 16) upw = 1;
    static const char* e = diag_env("CCN_STEM_UPW");                // diagnostics build only: units per wave
    if (e && atoi(e) > 0) upw = atoi(e);
    if (upw_out) *upw_out = upw;
    return (units + 4 * upw - 1) / (4 * upw);
}

hipError_t launch_stem2(const ConvArgs& a, hipStream_t s)
{
    int upw = 1;
    const int blocks = stem2_blocks(a.Hin, a.Win, &upw);
    if (!a.wfrag || (size_t)a.B * a.Cin * a.Hin * a.Win * 4 >= 0x7FFFFFF0u || (size_t)a.B * a.Hin * a.Win * a.Cout * 2 >= 0x7FFFFFF0u)
        return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks, (unsigned)a.B);
    switch (a.Cout) {
        case 192: hipLaunchKernelGGL((stem_kernel<6, false>), grid, dim3(256), 0, s, a, upw); break;   // (C4's width)
        case 128:
            if (a.cpg % 16 == 0) hipLaunchKernelGGL((stem_kernel<4, true>), grid, dim3(256), 0, s, a, upw);
            else hipLaunchKernelGGL((stem_kernel<4, false>), grid, dim3(256), 0, s, a, upw);
            break;
        case 64: hipLaunchKernelGGL((stem_kernel<2, false>), grid, dim3(256), 0, s, a, upw); break;
        case 32: hipLaunchKernelGGL((stem_kernel<1, false>), grid, dim3(256), 0, s, a, upw); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ccn
