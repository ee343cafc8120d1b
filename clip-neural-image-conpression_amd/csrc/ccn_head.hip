// Head of the UNet for the bf16 mode: out_norm (GroupNorm, no activation: models/unet.py:105) -> Conv2d(C, img_ch, 3,
// padding=1) (models/unet.py:79,106) -> eps and/or the DDIM update of the NCHW fp32 state (diffusion/ddim.py:34-45).
//
// Cout = 3.  As an implicit GEMM with the taps in K the MFMA N tile is 29/32 padding (38.6 GFLOP issued for 3.6 useful)
// and every tap re-reads its input fragment from LDS.  Here the nine taps sit in N instead:
//     G[p][tap*3 + co] = sum_c x[p][c] * W'[c][tap*3 + co]           one 32-wide N tile = 9 taps x 3 channels, K = C
//     out[p][co]       = bias[co] + sum_tap ( G[p + d_tap][tap*3 + co] + S[tap*3 + co] )   over the taps inside the image
// with the GroupNorm affine folded into the weights per SAMPLE: W'[c][n] = a_b[c] W[c][n], S[n] = sum_c c_b[c] W[c][n]
// (head_prep_kernel, which replaces the gn_finalize launch of out_norm).  A workgroup computes G for the 10 x 34 halo
// of its 8 x 32 output tile straight from global memory (input read once, 134 MB at C2 / batch 8), parks it in LDS as fp32 and then
// every thread gathers the 27 terms of its pixel and applies the update.  HBM-bound instead of LDS/latency-bound.
#include "ccn_device.h"

namespace ccn {

namespace {
constexpr int HG_PITCH = 27;                                       // floats per halo pixel in LDS: the 27 useful columns (odd: conflict-free
                                                                   // gathers); 36.7 KB per workgroup -> four workgroups per CU
constexpr int HG_ROWS = 10, HG_COLS = 34, HG_PIX = HG_ROWS * HG_COLS;
}

// Per-sample head weights from out_norm's finalized (scale, shift) table (pair-interleaved, see GnCoef): fragments
// [sample][k-step][lane][8 bf16] (lane (n = lane & 31, hh = lane >> 5) holds W'[16 s + 8 hh + e][n]) and S.  Grid (k-steps + 1, B).
__global__ __launch_bounds__(512) void head_prep_kernel(const float2* __restrict__ ab, int C, const float* __restrict__ w, int Cout,
                                                         unsigned short* __restrict__ wq, float* __restrict__ sq, float* __restrict__ carry, int step)
{
    const int b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x, ksteps = C / 16;
    const float* abf = (const float*)(ab + (size_t)b * C);         // channels (2p, 2p+1) -> {a, a, c, c}
    const int N = Cout * 9;                                        // <= 32; n = tap*Cout + co; w: reference layout (Cout, C, 3, 3) fp32
    if (s < ksteps) {
        const int e = tid & 7, ln = tid >> 3;
        const int n = ln & 31, c = 16 * s + 8 * (ln >> 5) + e;
        float v = 0.f;
        if (n < N) { const int tap = n / Cout, co = n - tap * Cout; v = abf[4 * (c >> 1) + (c & 1)] * w[((size_t)co * C + c) * 9 + tap]; }
        // bf16 rounding with the error fed forward from DDIM step to DDIM step (the conv weights get the same treatment at commit
        // time, diffuse_round_phases in ccn_api.hip): W' changes slowly along a trajectory, so independent rounding would repeat
        // nearly the same error in all 50 steps; with the carry the running sum of the rounded weights tracks the running sum of W'.
        // Step 0 (and ccn_forward) starts from a zero carry: plain round-to-nearest-even.
        const size_t ci = ((size_t)b * ksteps + s) * 512 + tid;
        const float t = v + (step > 0 ? carry[ci] : 0.f);
        const unsigned u = __float_as_uint(t);
        const unsigned hi = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        carry[ci] = t - __uint_as_float(hi);
        wq[ci] = (unsigned short)(hi >> 16);
    } else {
        // S[n] = sum_c shift[c] W[c][n]: 16 threads per n, fixed order
        const int n = tid >> 4, part = tid & 15;
        float acc = 0.f;
        if (n < N) {
            const int tap = n / Cout, co = n - tap * Cout;
            for (int c = part; c < C; c += 16) acc = fmaf(abf[4 * (c >> 1) + 2 + (c & 1)], w[((size_t)co * C + c) * 9 + tap], acc);
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) acc += __shfl_xor(acc, m);
        if (part == 0) sq[b * 32 + n] = acc;
    }
}

template <int KSTEPS>
__global__ __launch_bounds__(256, 4) void head_kernel(const ConvArgs a, const unsigned short* __restrict__ wq, const float* __restrict__ sq,
                                                   const float* __restrict__ bias)
{
    typedef __bf16 T;
    __shared__ float Gs[HG_PIX * HG_PITCH];
    __shared__ float Ss[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int H = a.Hin, W = a.Win, C = a.Cin;
    const int n_tx = (W + 31) >> 5, n_ty = (H + 7) >> 3;
    int bid = blockIdx.x;
    const int tx = bid % n_tx; bid /= n_tx;
    const int ty = bid % n_ty; const int b = bid / n_ty;
    const int y0 = ty * 8 - 1, x0 = tx * 32 - 1;

    if (tid < 32) Ss[tid] = sq[b * 32 + tid];

    constexpr unsigned OOB = 0x7FFFFFF0u;
    const unsigned in_bytes = (unsigned)((size_t)a.B * H * W * C * sizeof(T));
    const auto isrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, in_bytes, 0x00020000);
    // 340 halo pixels in M tiles of 32: wave w takes tiles w, w+4, w+8 (11 tiles, the last one partial).  K in passes of at most 8
    // k-steps (128 channels): all loads of a pass first, then its MFMAs (C = 192, C4's width: two passes of 6 -- the register budget
    // of one pass is what keeps three workgroups per CU)
    constexpr int MT = 3, NPASS = KSTEPS > 8 ? 2 : 1, KP = KSTEPS / NPASS;
    static_assert(KSTEPS % NPASS == 0, "k-steps split evenly over the passes");
    unsigned base[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int p = (wave + 4 * t) * 32 + r;                      // halo pixel of this lane (A operand row)
        const int hy = p / HG_COLS, hx = p - hy * HG_COLS;
        const int iy = y0 + hy, ix = x0 + hx;
        const bool ok = p < HG_PIX && iy >= 0 && iy < H && ix >= 0 && ix < W;
        base[t] = ok ? (unsigned)(((b * H + iy) * W + ix) * C + 8 * h) * 2u : OOB;
    }
    if constexpr (NPASS == 1) {
        // all loads first, then tile by tile: MFMAs, and the tile's G rows go to LDS while the next tile's loads are still landing
        u32x4 wf[KSTEPS];                                          // B operand: W'_b fragments, [k-step][lane]
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) wf[s] = *(const u32x4*)(wq + ((size_t)b * KSTEPS + s) * 512 + lane * 8);
        u32x4 af[MT][KSTEPS];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) af[t][s] = __builtin_amdgcn_raw_buffer_load_b128(isrd, base[t], s * 32, 0);   // out of image -> zeros
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int mt = wave + 4 * t;
            if (mt * 32 >= HG_PIX) break;                           // wave-uniform
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[t][s]), __builtin_bit_cast(bf16x8, wf[s]), acc, 0, 0, 0);
            // D: lane r = column n, register q = pixel row (q & 3) + 8 (q >> 2) + 4 h of the M tile
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pp = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (pp < HG_PIX && r < HG_PITCH) Gs[pp * HG_PITCH + r] = acc[q];
            }
        }
    } else {
        f32x16 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            u32x4 wf[KP];
#pragma unroll
            for (int s = 0; s < KP; ++s) wf[s] = *(const u32x4*)(wq + ((size_t)b * KSTEPS + ps * KP + s) * 512 + lane * 8);
            u32x4 af[MT][KP];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int s = 0; s < KP; ++s) af[t][s] = __builtin_amdgcn_raw_buffer_load_b128(isrd, base[t], (ps * KP + s) * 32, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if ((wave + 4 * t) * 32 >= HG_PIX) break;           // wave-uniform
#pragma unroll
                for (int s = 0; s < KP; ++s)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[t][s]), __builtin_bit_cast(bf16x8, wf[s]), acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int mt = wave + 4 * t;
            if (mt * 32 >= HG_PIX) break;                           // wave-uniform
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pp = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (pp < HG_PIX && r < HG_PITCH) Gs[pp * HG_PITCH + r] = acc[t][q];
            }
        }
    }
    __syncthreads();
    // gather: thread -> output pixel (row tid >> 5, column tid & 31) of the tile
    const int oy = tid >> 5, ox = tid & 31;
    const int y = ty * 8 + oy, x = tx * 32 + ox;
    if (y >= H || x >= W) return;
    const int Cout = a.Cout;
    float e3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;                  // halo coordinates of the tap: (oy + dy, ox + dx)
        const int iy = y + dy - 1, ix = x + dx - 1;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const float* g = Gs + ((oy + dy) * HG_COLS + ox + dx) * HG_PITCH + tap * Cout;
#pragma unroll
            for (int co = 0; co < 4; ++co)
                if (co < Cout) e3[co] += g[co] + Ss[tap * Cout + co];
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        if (co >= Cout) break;
        const float e = e3[co] + bias[co];
        const size_t idx = ((size_t)(b * Cout + co) * H + y) * W + x;
        if (a.eps_out) a.eps_out[idx] = e;
        if (a.do_ddim) {
            // diffusion/ddim.py:38-43 -- every op rounds to fp32 separately, like the torch ops
            const float xs = a.x_state[idx];
            float x0v = __fdiv_rn(__fsub_rn(xs, __fmul_rn(a.c0, e)), a.c1);
            x0v = fminf(fmaxf(x0v, -1.0f), 1.0f);
            float xn = __fadd_rn(__fmul_rn(a.c2, x0v), __fmul_rn(a.c3, e));
            if (a.noise) xn = __fadd_rn(xn, __fmul_rn(a.sigma, a.noise[idx]));      // eta > 0 (diffusion/ddim.py:44-45)
            a.x_state[idx] = xn;
        }
    }
}

bool head2_supported(int dtype, int cin, int cout, int G)
{
    const int g = cin < G ? cin : G;
    return dtype == 1 && cout * 9 <= 27 && cout <= 3 && (cin == 32 || cin == 64 || cin == 128 || cin == 192) && cin % g == 0;
}

// fragments (bf16) | rounding carry (fp32) | S
size_t head2_scratch_bytes(int B, int C) { return (size_t)B * (C / 16) * 512 * 6 + (size_t)B * 32 * 4 + 512; }

// `scratch`: head2_scratch_bytes(); `ab`: out_norm's finalized scale/shift table; `w_f32` the fp32 (Cout, C, 3, 3) weights on the device;
// `step`: DDIM step index (0 restarts the rounding carry)
hipError_t launch_head2(const ConvArgs& a, const float2* ab, const float* w_f32, void* scratch, int step, hipStream_t s)
{
    if ((size_t)a.B * a.Hin * a.Win * a.Cin * 2 >= 0x7FFFFFF0u || a.Cin > 512) return hipErrorInvalidValue;
    const int ks = a.Cin / 16;
    const size_t nfrag = (size_t)a.B * ks * 512;
    unsigned short* wq = (unsigned short*)scratch;
    float* carry = (float*)((char*)scratch + ((nfrag * 2 + 255) & ~(size_t)255));
    float* sq = carry + nfrag;
    hipLaunchKernelGGL(head_prep_kernel, dim3(ks + 1, a.B), dim3(512), 0, s, ab, a.Cin, w_f32, a.Cout, wq, sq, carry, step);
    const unsigned grid = (unsigned)(a.B * ((a.Hin + 7) / 8) * ((a.Win + 31) / 32));
    switch (ks) {
        case 12: hipLaunchKernelGGL(head_kernel<12>, dim3(grid), dim3(256), 0, s, a, wq, sq, a.bias); break;
        case 8: hipLaunchKernelGGL(head_kernel<8>, dim3(grid), dim3(256), 0, s, a, wq, sq, a.bias); break;
        case 4: hipLaunchKernelGGL(head_kernel<4>, dim3(grid), dim3(256), 0, s, a, wq, sq, a.bias); break;
        case 2: hipLaunchKernelGGL(head_kernel<2>, dim3(grid), dim3(256), 0, s, a, wq, sq, a.bias); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ccn
