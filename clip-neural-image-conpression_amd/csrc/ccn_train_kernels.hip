// gfx950 kernels of the training step (train/diffusion_train.py:119-124,137-140): the backward pass of CLIPCondUNet.forward.
//
// The data gradients of the convolutions are convolutions again and run on the forward implicit-GEMM kernels with
// re-packed weights (pack_group_kernel): 3x3 s1 -> 3x3 s1 with flipped taps; 3x3 s2 -> the ConvTranspose kernel with a
// zero-padded 4x4 kernel; ConvTranspose 4x4 s2 -> the stride-2 kernel run with 16 taps.  New here:
//   wgrad_kernel         dW[tap][co][ci] = sum over pixels dY[p][co] * act(GN(x))[p + d_tap][ci]: the contraction runs over
//                        PIXELS, so with channel-contiguous (NHWC) tiles in LDS each lane of v_mfma_f32_32x32x2_f32 reads one
//                        float per operand (lane = channel, k = pixel) -- conflict-free without any transpose.  GroupNorm+SiLU
//                        of the forward input is redone while staging (the activated tensor is never stored).  Pixel tiles are
//                        split over workgroups; partial sums go to scratch and are reduced in a fixed order (deterministic).
//   gn_bwd_*             GroupNorm (+SiLU) backward in two HBM passes: per-channel sums, then the apply pass, which also folds
//                        the residual-gradient add and the FiLM backward (scale by 1+s, sums for d scale / d shift).
//   im2col27 / nchw_to_nhwc_pad   layout changes that let the stem / head weights use the same weight-gradient kernel.
//   tlinear_*            conditioning MLP / FiLM linears, fp32.
//   adamw_kernel, mse_loss_grad
#include "ccn_device.h"
#include "ccn_train.h"
#include <cstdlib>

namespace ccn {

namespace {
template <typename T> __device__ __forceinline__ T to_elem(float v);
template <> __device__ __forceinline__ float to_elem<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 to_elem<__bf16>(float v) { return (__bf16)v; }
template <typename T> __device__ __forceinline__ float from_elem(T v) { return (float)v; }

// d silu(y) / dy
template <typename T> __device__ __forceinline__ float dsilu(float y);
template <> __device__ __forceinline__ float dsilu<float>(float y) { const float sg = 1.0f / (1.0f + expf(-y)); return sg * (1.0f + y * (1.0f - sg)); }
template <> __device__ __forceinline__ float dsilu<__bf16>(float y) { const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y)); return sg * (1.0f + y * (1.0f - sg)); }

template <typename T> struct Chunk {                     // one pixel's channel slice: EPC8 = 8 consecutive channels
    static constexpr int E = 8;
    static __device__ __forceinline__ void load(const T* p, float* v);
    static __device__ __forceinline__ void store(T* p, const float* v);
};
template <> __device__ __forceinline__ void Chunk<float>::load(const float* p, float* v) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <> __device__ __forceinline__ void Chunk<float>::store(float* p, const float* v) {
    *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]}; *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void Chunk<__bf16>::load(const __bf16* p, float* v) { Vec16<__bf16>::unpack(*(const u32x4*)p, v); }
template <> __device__ __forceinline__ void Chunk<__bf16>::store(__bf16* p, const float* v) { *(u32x4*)p = Vec16<__bf16>::pack(v); }

// scale / shift of channel c of one sample from the pair-interleaved table (see GnCoef)
__device__ __forceinline__ void ab_of(const float* abf, int c, float& a, float& cc) { a = abf[4 * (c >> 1) + (c & 1)]; cc = abf[4 * (c >> 1) + 2 + (c & 1)]; }
}  // namespace

// ---- weight repacking --------------------------------------------------------------------------------------------------
// thread -> one (n, k) pair: its taps are contiguous in the reference layouts (9 or 16 floats), so the reads are whole
// segments and the writes, one per tap plane, are coalesced across the threads' consecutive k
template <typename T>
__global__ __launch_bounds__(256) void pack_group_kernel(const float* __restrict__ params, const PackDesc* __restrict__ descs)
{
    const PackDesc d = descs[blockIdx.y];
    const float* w = params + d.src_off;
    T* dst = (T*)d.dst;
    const int O = d.O, I = d.I, Np = d.Np, Kp = d.Kp;
    const size_t plane = (size_t)Np * Kp;
    if (d.mode == PK_FRAG_STEM || d.mode == PK_FRAG_STEM_HEAD_DG) {
        // ccn_stem.hip's order: [N / 32][k-step 2][lane = h * 32 + r][8]: lane (r, h) holds output channel j * 32 + r, k = 16 s + 8 h + e
        const int N = d.mode == PK_FRAG_STEM ? O : I, K = (d.mode == PK_FRAG_STEM ? I : O) * 9;
        for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)N * 32; idx += (size_t)gridDim.x * blockDim.x) {
            const int k = (int)(idx & 31), n = (int)(idx >> 5);
            float v = 0.f;
            if (d.mode == PK_FRAG_STEM) v = k < K ? w[(size_t)n * K + k] : (k == K ? params[d.aux_off + n] : 0.f);
            else if (k < K) { const int co = k / 9, tp = k - co * 9; v = w[((size_t)co * I + n) * 9 + (8 - tp)]; }
            dst[((((size_t)(n >> 5) * 2 + (k >> 4)) * 64) + (size_t)((((k >> 3) & 1) << 5) | (n & 31))) * 8 + (k & 7)] = to_elem<T>(v);
        }
        return;
    }
    if (d.mode == PK_STEM || d.mode == PK_HEAD_DG) {
        for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < plane; idx += (size_t)gridDim.x * blockDim.x) {
            const int k = (int)(idx % Kp), n = (int)(idx / Kp);
            float v = 0.f;
            if (d.mode == PK_STEM) { if (n < O && k < I * 9) v = w[(size_t)n * I * 9 + k]; }
            else if (n < I && k < O * 9) { const int co = k / 9, tp = k - co * 9; v = w[((size_t)co * I + n) * 9 + (8 - tp)]; }
            dst[idx] = to_elem<T>(v);
        }
        return;
    }
    if (sizeof(T) == 2 && (d.mode == PK_FRAG3 || d.mode == PK_FRAG3_DG || d.mode == PK_FRAG_S2 || d.mode == PK_FRAG_CT || d.mode == PK_FRAG_CT_DG ||
                           d.mode == PK_FRAG_P4_DG)) {
        // fragment orders: thread -> (output channel n, eight consecutive input channels k0 .. k0 + 7) with n fastest -- the eight values of
        // a (tap / pass) slot are one 16-byte piece of a fragment and the pieces of neighbouring n are neighbours, so every store
        // instruction writes whole 1-KiB fragment rows (one thread per (n, k) wrote 2-byte elements in eight places per instruction)
        const int K8 = Kp >> 3, mode = d.mode;
        for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)Np * K8; idx += (size_t)gridDim.x * blockDim.x) {
            const int n = (int)(idx % Np), k0 = (int)(idx / Np) * 8;
            float v[8][16];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
#pragma unroll
                for (int t = 0; t < 16; ++t) v[e][t] = 0.f;
                if (mode == PK_FRAG3 || mode == PK_FRAG_S2) {
                    if (n < O && k < I) { const float* p = w + ((size_t)n * I + k) * 9;
#pragma unroll
                        for (int t = 0; t < 9; ++t) v[e][t] = p[t]; }
                } else if (mode == PK_FRAG3_DG) {
                    if (n < I && k < O) { const float* p = w + ((size_t)k * I + n) * 9;
#pragma unroll
                        for (int t = 0; t < 9; ++t) v[e][t] = p[8 - t]; }
                } else if (mode == PK_FRAG_CT) {
                    if (n < O && k < I) { const float* p = w + ((size_t)k * O + n) * 16;
#pragma unroll
                        for (int t = 0; t < 16; ++t) v[e][t] = p[t]; }
                } else if (mode == PK_FRAG_CT_DG) {
                    if (n < I && k < O) { const float* p = w + ((size_t)k * I + n) * 9;
#pragma unroll
                        for (int t = 0; t < 9; ++t) v[e][(t / 3) * 4 + (t % 3)] = p[t]; }
                } else {                                         // PK_FRAG_P4_DG
                    if (n < I && k < O) { const float* p = w + ((size_t)n * O + k) * 16;
#pragma unroll
                        for (int t = 0; t < 16; ++t) v[e][t] = p[t]; }
                }
            }
            auto piece = [&](int tp) __attribute__((always_inline)) -> u32x4 {     // slot <- kernel element tp (< 0: empty slot)
                u32x4 r = {0u, 0u, 0u, 0u};
                if (tp >= 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = pack_bf2(v[2 * q][tp], v[2 * q + 1][tp]);
                }
                return r;
            };
            unsigned short* const db = (unsigned short*)dst;
            if (mode == PK_FRAG3 || mode == PK_FRAG3_DG) {
#pragma unroll
                for (int t = 0; t < 9; ++t) *(u32x4*)(db + pr3_frag_index(n, k0, t, Np)) = piece(t);
            } else if (mode == PK_FRAG_S2) {
#pragma unroll
                for (int ps = 0; ps < 10; ++ps) *(u32x4*)(db + prs2_frag_index(ps >> 1, ps & 1, n, k0, Np)) = piece(prs2_tap(ps >> 1, ps & 1));
            } else if (mode == PK_FRAG_P4_DG) {
#pragma unroll
                for (int pt = 0; pt < 16; ++pt) *(u32x4*)(db + prp4_frag_index(pt >> 2, pt & 3, n, k0, Np)) = piece(prp4_tap(pt >> 2, pt & 3));
            } else {
#pragma unroll
                for (int pt = 0; pt < 16; ++pt) *(u32x4*)(db + prct_frag_index(pt >> 2, pt & 3, n, k0, Np, Kp >> 6)) = piece(prct_tap(pt >> 2, pt & 3));
            }
        }
        return;
    }
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < plane; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % Kp), n = (int)(idx / Kp);
        float v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = 0.f;
        switch (d.mode) {
            case PK_CONV3:
                if (n < O && k < I) { const float* p = w + ((size_t)n * I + k) * 9;
#pragma unroll
                    for (int t = 0; t < 9; ++t) v[t] = p[t]; }
                break;
            case PK_CONVT:
                if (n < O && k < I) { const float* p = w + ((size_t)k * O + n) * 16;
#pragma unroll
                    for (int t = 0; t < 16; ++t) v[t] = p[t]; }
                break;
            case PK_DG3S1:
                if (n < I && k < O) { const float* p = w + ((size_t)k * I + n) * 9;
#pragma unroll
                    for (int t = 0; t < 9; ++t) v[t] = p[8 - t]; }
                break;
            case PK_DG3S2:
                if (n < I && k < O) { const float* p = w + ((size_t)k * I + n) * 9;
#pragma unroll
                    for (int t = 0; t < 16; ++t) v[t] = ((t >> 2) < 3 && (t & 3) < 3) ? p[(t >> 2) * 3 + (t & 3)] : 0.f; }
                break;
            case PK_DGT:
                if (n < I && k < O) { const float* p = w + ((size_t)n * O + k) * 16;
#pragma unroll
                    for (int t = 0; t < 16; ++t) v[t] = p[t]; }
                break;
            case PK_FRAG3:
                if (n < O && k < I) { const float* p = w + ((size_t)n * I + k) * 9;
#pragma unroll
                    for (int t = 0; t < 9; ++t) v[t] = p[t]; }
                break;
            case PK_FRAG3_DG:
                if (n < I && k < O) { const float* p = w + ((size_t)k * I + n) * 9;
#pragma unroll
                    for (int t = 0; t < 9; ++t) v[t] = p[8 - t]; }
                break;
            case PK_FRAG_S2:
                if (n < O && k < I) { const float* p = w + ((size_t)n * I + k) * 9;
#pragma unroll
                    for (int t = 0; t < 9; ++t) v[t] = p[t]; }
                break;
            case PK_FRAG_CT:
                if (n < O && k < I) { const float* p = w + ((size_t)k * O + n) * 16;
#pragma unroll
                    for (int t = 0; t < 16; ++t) v[t] = p[t]; }
                break;
            case PK_FRAG_P4_DG:
                if (n < I && k < O) { const float* p = w + ((size_t)n * O + k) * 16;
#pragma unroll
                    for (int t = 0; t < 16; ++t) v[t] = p[t]; }
                break;
            case PK_FRAG_CT_DG:
                if (n < I && k < O) { const float* p = w + ((size_t)k * I + n) * 9;
#pragma unroll
                    for (int t = 0; t < 16; ++t) v[t] = ((t >> 2) < 3 && (t & 3) < 3) ? p[(t >> 2) * 3 + (t & 3)] : 0.f; }
                break;
        }
        if (d.mode == PK_FRAG_S2) {
            // plane-pass order of the persistent kernel's stride-2 form (prs2_frag_index, ccn_internal.h; pack_conv3 in ccn_api.hip)
#pragma unroll
            for (int ps = 0; ps < 10; ++ps) {
                const int tp = prs2_tap(ps >> 1, ps & 1);
                float val = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) val = t == tp ? v[t] : val;
                dst[prs2_frag_index(ps >> 1, ps & 1, n, k, Np)] = to_elem<T>(val);
            }
            continue;
        }
        if (d.mode == PK_FRAG_P4_DG) {
            // plane passes of the 4x4 stride-2 form (prp4_frag_index)
#pragma unroll
            for (int pt = 0; pt < 16; ++pt) {
                const int tp = prp4_tap(pt >> 2, pt & 3);
                float val = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) val = t == tp ? v[t] : val;
                dst[prp4_frag_index(pt >> 2, pt & 3, n, k, Np)] = to_elem<T>(val);
            }
            continue;
        }
        if (d.mode == PK_FRAG_CT || d.mode == PK_FRAG_CT_DG) {
            // parity / tap order of its ConvTranspose form (prct_frag_index; pack_convT in ccn_api.hip)
#pragma unroll
            for (int pt = 0; pt < 16; ++pt) {
                const int tp = prct_tap(pt >> 2, pt & 3);
                float val = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) val = t == tp ? v[t] : val;
                dst[prct_frag_index(pt >> 2, pt & 3, n, k, Np, Kp >> 6)] = to_elem<T>(val);
            }
            continue;
        }
        if (d.mode == PK_FRAG3 || d.mode == PK_FRAG3_DG) {
            // fragment order of the persistent kernel's 3x3 form (pr3_frag_index, ccn_internal.h; pack_conv3 in ccn_api.hip)
#pragma unroll
            for (int t = 0; t < 9; ++t) dst[pr3_frag_index(n, k, t, Np)] = to_elem<T>(v[t]);
            continue;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < d.taps) dst[(size_t)t * plane + idx] = to_elem<T>(v[t]);
    }
}
hipError_t launch_pack_group(int dtype, const float* params, const PackDesc* descs_dev, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    if (dtype == 0) hipLaunchKernelGGL(pack_group_kernel<float>, dim3(512, n), dim3(256), 0, s, params, descs_dev);
    else hipLaunchKernelGGL(pack_group_kernel<__bf16>, dim3(512, n), dim3(256), 0, s, params, descs_dev);
    return hipGetLastError();
}

// ---- grouped FiLM linears ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void film_group_fwd_kernel(const float* __restrict__ params, const LinDesc* __restrict__ descs, const float* __restrict__ h,
                                                              float* __restrict__ film, int B, int K, int F)
{
    const LinDesc d = descs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = blockIdx.x * 4 + wave;
    if (n >= d.N) return;
    const float* wrow = params + d.w_off + (size_t)n * K;
    const float bias = params[d.b_off + n];
    for (int r = 0; r < B; ++r) {
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) acc = fmaf(h[(size_t)r * K + k], wrow[k], acc);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
        if (lane == 0) film[(size_t)r * F + d.out_off + n] = acc + bias;
    }
}
hipError_t launch_film_group_fwd(const float* params, const LinDesc* d, int n, int maxN, const float* h, float* film, int B, int K, int F, hipStream_t s)
{
    hipLaunchKernelGGL(film_group_fwd_kernel, dim3((maxN + 3) / 4, n), dim3(256), 0, s, params, d, h, film, B, K, F);
    return hipGetLastError();
}
__global__ void film_group_dw_kernel(float* __restrict__ grads, const LinDesc* __restrict__ descs, const float* __restrict__ dfilm, const float* __restrict__ h,
                                     int B, int K, int F)
{
    const LinDesc d = descs[blockIdx.y];
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)d.N * K) return;
    const int k = (int)(idx % K), n = (int)(idx / K);
    float acc = 0.f, sb = 0.f;
    for (int r = 0; r < B; ++r) { const float dv = dfilm[(size_t)r * F + d.out_off + n]; acc = fmaf(dv, h[(size_t)r * K + k], acc); sb += dv; }
    grads[d.w_off + idx] += acc;
    if (k == 0) grads[d.b_off + n] += sb;
}
hipError_t launch_film_group_dw(float* grads, const LinDesc* d, int n, int maxN, const float* dfilm, const float* h, int B, int K, int F, hipStream_t s)
{
    hipLaunchKernelGGL(film_group_dw_kernel, dim3((unsigned)(((size_t)maxN * K + 255) / 256), n), dim3(256), 0, s, grads, d, dfilm, h, B, K, F);
    return hipGetLastError();
}
// dh[r][k] += sum_n dfilm[r][off + n] W[n][k] for every linear (dh zeroed by the caller); block = 64 k x 4 waves over n, 4 rows
__global__ __launch_bounds__(256) void film_group_dx_kernel(const float* __restrict__ params, const LinDesc* __restrict__ descs, const float* __restrict__ dfilm,
                                                             float* __restrict__ dh, int B, int K, int F)
{
    __shared__ float red[4][4][64];
    const LinDesc d = descs[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane, r0 = blockIdx.y * 4, kk = k < K ? k : K - 1;
    const float* W = params + d.w_off;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int n = wave; n < d.N; n += 4) {
        const float w = W[(size_t)n * K + kk];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int r = r0 + j < B ? r0 + j : B - 1; acc[j] = fmaf(dfilm[(size_t)r * F + d.out_off + n], w, acc[j]); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    if (wave == 0 && k < K)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + j < B) atomicAdd(dh + (size_t)(r0 + j) * K + k, (red[0][j][lane] + red[1][j][lane]) + (red[2][j][lane] + red[3][j][lane]));
}
hipError_t launch_film_group_dx(const float* params, const LinDesc* d, int n, const float* dfilm, float* dh, int B, int K, int F, hipStream_t s)
{
    hipLaunchKernelGGL(film_group_dx_kernel, dim3((K + 63) / 64, (B + 3) / 4, n), dim3(256), 0, s, params, d, dfilm, dh, B, K, F);
    return hipGetLastError();
}

// ---- GroupNorm statistics ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float2* __restrict__ part, int G, int n_sp, int n_nt, int bn, int cpg, int C,
                                                        double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, float2* __restrict__ ab, float2* __restrict__ stats)
{
    __shared__ double red[4][2];
    const int bg = blockIdx.x, b = bg / G, g = bg % G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jlo = (g * cpg) / bn, jhi = ((g + 1) * cpg - 1) / bn, nj = jhi - jlo + 1, ne = n_sp * nj;
    const float2* base = part + (size_t)(b * G + g) * n_sp * n_nt;
    double s1 = 0.0, s2 = 0.0;
    for (int e = tid; e < ne; e += 256) {
        const int sp = e / nj, j = jlo + (e - sp * nj);
        const float2 v = base[(size_t)sp * n_nt + j];
        s1 += (double)v.x; s2 += (double)v.y;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { s1 += __shfl_xor(s1, s); s2 += __shfl_xor(s2, s); }
    if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; }
    __syncthreads();
    s1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    s2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    if (tid == 0) stats[bg] = make_float2((float)mean, (float)rstd);
    for (int c = g * cpg + tid; c < (g + 1) * cpg; c += 256) {
        const double sc = (double)gamma[c] * rstd;
        float* const row = (float*)(ab + (size_t)b * C) + 4 * (c >> 1) + (c & 1);
        row[0] = (float)sc; row[2] = (float)((double)beta[c] - mean * sc);
    }
}
hipError_t launch_gn_stats(const float2* part, int B, int G, int n_sp, int n_nt, int bn, int cpg, int C, double count, const float* gamma,
                           const float* beta, float eps, float2* ab, float2* stats, hipStream_t s)
{
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * G), dim3(256), 0, s, part, G, n_sp, n_nt, bn, cpg, C, count, gamma, beta, eps, ab, stats);
    return hipGetLastError();
}

// ---- GroupNorm (+SiLU) backward -----------------------------------------------------------------------------------------
// Thread -> a fixed slice of 8 channels (coefficients in registers) x a strided set of the block's pixels.
GnBwdGeom gn_bwd_geom(int dtype, int B, int HW, int C)
{
    (void)dtype;
    GnBwdGeom g{};
    const int nsl = C / 8;
    int nslb = nsl <= 256 ? nsl : 256;
    while (nsl % nslb) --nslb;                                   // largest divisor of nsl that fits a workgroup
    g.nslb = nslb; g.zblocks = nsl / nslb; g.pstep = 256 / nslb;
    // pixels per workgroup: ~512 workgroups per launch over the batch (two per CU: enough loads in flight for an HBM-bound pass, and
    // every workgroup's fixed cost -- coefficient loads, the LDS reduction, its row of partial sums -- is paid half as often as with
    // the 1024 of round 2: 684 vs 643 images/s on the training step, sweep in docs/EXPERIMENTS.md R3.12), at most 64 passes
    static const int env_minblk = diag_env("CCN_GNB_MINBLK") ? atoi(diag_env("CCN_GNB_MINBLK")) : 0;
    const int minblk = env_minblk > 0 ? env_minblk : (512 / (B > 0 ? B : 1) > 32 ? 512 / (B > 0 ? B : 1) : 32);
    int iters = HW / (g.pstep * minblk);
    static const int cap = diag_env("CCN_GNB_ITERS") ? atoi(diag_env("CCN_GNB_ITERS")) : 64;
    iters = iters < 1 ? 1 : (iters > cap ? cap : iters);
    g.ppb = g.pstep * iters;
    g.nblk = (HW + g.ppb - 1) / g.ppb;
    return g;
}

// block-level reduction of per-thread (s1[8], s2[8]) over the pixel lanes; result valid in threads with pp == 0
__device__ __forceinline__ void block_reduce16(float* sh, int tid, int sl, int pp, int nslb, int pstep, float* s1, float* s2)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[tid * 17 + e] = s1[e]; sh[tid * 17 + 8 + e] = s2[e]; }
    __syncthreads();
    if (pp == 0) {
        for (int q = 1; q < pstep; ++q) {
            const float* o = sh + (q * nslb + sl) * 17;
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += o[e]; s2[e] += o[8 + e]; }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dA, const float2* __restrict__ ab,
                                                             const float2* __restrict__ stats, float2* __restrict__ part, int HW, int C,
                                                             int cpg, int G, int silu, int nslb, int pstep, int ppb, int nblk)
{
    __shared__ float sh[256 * 17];
    const int tid = threadIdx.x, b = blockIdx.y, blk = blockIdx.x;
    const int sl = tid % nslb, pp = tid / nslb;
    const bool act = pp < pstep;
    const int ch0 = (blockIdx.z * nslb + sl) * 8;
    float a[8], c[8], mu[8], rs[8], s1[8], s2[8];
    const float* abf = (const float*)(ab + (size_t)b * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ab_of(abf, ch0 + e, a[e], c[e]);
        const float2 st = stats[b * G + (ch0 + e) / cpg];
        mu[e] = st.x; rs[e] = st.y; s1[e] = 0.f; s2[e] = 0.f;
    }
    const int pend = min(HW, (blk + 1) * ppb);
    if (act)
        for (int p0 = blk * ppb + pp; p0 < pend; p0 += 4 * pstep) {       // four pixels' loads in flight
            float xv[4][8], dv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * pstep;
                if (p < pend) { const size_t off = ((size_t)b * HW + p) * C + ch0; Chunk<T>::load(x + off, xv[u]); Chunk<T>::load(dA + off, dv[u]); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p0 + u * pstep >= pend) break;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float da = dv[u][e];
                    if (silu) da *= dsilu<T>(fmaf(xv[u][e], a[e], c[e]));
                    const float xh = (xv[u][e] - mu[e]) * rs[e];
                    s1[e] += da; s2[e] = fmaf(da, xh, s2[e]);
                }
            }
        }
    block_reduce16(sh, tid, sl, pp, nslb, pstep, s1, s2);
    if (pp == 0) {
        float2* o = part + ((size_t)b * nblk + blk) * C + ch0;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = make_float2(s1[e], s2[e]);
    }
}
hipError_t launch_gn_bwd_reduce(int dtype, const void* x, const void* dA, const float2* ab, const float2* stats, float2* part, int B, int HW,
                                int C, int cpg, int G, int silu, hipStream_t s)
{
    if (C % 8) return hipErrorInvalidValue;
    const GnBwdGeom g = gn_bwd_geom(dtype, B, HW, C);
    const dim3 grid(g.nblk, B, g.zblocks);
    if (dtype == 0) hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)dA, ab, stats, part, HW, C, cpg, G, silu, g.nslb, g.pstep, g.ppb, g.nblk);
    else hipLaunchKernelGGL(gn_bwd_reduce_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)x, (const __bf16*)dA, ab, stats, part, HW, C, cpg, G, silu, g.nslb, g.pstep, g.ppb, g.nblk);
    return hipGetLastError();
}

template <int CL>
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float2* __restrict__ part, int nblk, int C, int cpg, int G, double count,
                                                               const float* __restrict__ gamma, float2* __restrict__ gstat,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    // thread -> (channel of a CL-channel slab, one of 256/CL block lanes): coalesced rows, fixed summation order
    constexpr int RL = 256 / CL;
    __shared__ float red[RL][CL][2];
    __shared__ double mm[CL][2];
    const int bg = blockIdx.x, b = bg / G, g = bg % G, tid = threadIdx.x, cl = tid % CL, rl = tid / CL;
    double m1 = 0.0, m2 = 0.0;
    for (int c0 = g * cpg; c0 < (g + 1) * cpg; c0 += CL) {
        const int c = c0 + cl;
        const bool valid = c < (g + 1) * cpg;
        float p1 = 0.f, p2 = 0.f;
        if (valid) {
            int k = rl;
            for (; k + 7 * RL < nblk; k += 8 * RL) {                  // eight loads in flight, fixed order
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = part[((size_t)b * nblk + k + u * RL) * C + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) { p1 += v[u].x; p2 += v[u].y; }
            }
            for (; k < nblk; k += RL) { const float2 v = part[((size_t)b * nblk + k) * C + c]; p1 += v.x; p2 += v.y; }
        }
        red[rl][cl][0] = p1; red[rl][cl][1] = p2;
        __syncthreads();
        if (rl == 0 && valid) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int q = 0; q < RL; ++q) { s1 += (double)red[q][cl][0]; s2 += (double)red[q][cl][1]; }
            atomicAdd(dgamma + c, (float)s2); atomicAdd(dbeta + c, (float)s1);
            m1 += (double)gamma[c] * s1; m2 += (double)gamma[c] * s2;
        }
        __syncthreads();
    }
    if (rl == 0) { mm[cl][0] = m1; mm[cl][1] = m2; }
    __syncthreads();
    if (tid == 0) {
        m1 = 0.0; m2 = 0.0;
        for (int q = 0; q < CL; ++q) { m1 += mm[q][0]; m2 += mm[q][1]; }
        gstat[bg] = make_float2((float)(m1 / count), (float)(m2 / count));
    }
}
hipError_t launch_gn_bwd_finalize(const float2* part, int nblk, int B, int C, int cpg, int G, double count, const float* gamma, float2* gstat,
                                  float* dgamma, float* dbeta, hipStream_t s)
{
    if (cpg >= 64) hipLaunchKernelGGL(gn_bwd_finalize_kernel<64>, dim3(B * G), dim3(256), 0, s, part, nblk, C, cpg, G, count, gamma, gstat, dgamma, dbeta);
    else if (cpg >= 32) hipLaunchKernelGGL(gn_bwd_finalize_kernel<32>, dim3(B * G), dim3(256), 0, s, part, nblk, C, cpg, G, count, gamma, gstat, dgamma, dbeta);
    else hipLaunchKernelGGL(gn_bwd_finalize_kernel<16>, dim3(B * G), dim3(256), 0, s, part, nblk, C, cpg, G, count, gamma, gstat, dgamma, dbeta);
    return hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* dA, const float2* __restrict__ ab,
                                                            const float2* __restrict__ stats, const float2* __restrict__ gstat,
                                                            const T* __restrict__ addend, T* out, const float* __restrict__ film,
                                                            int film_bstride, float2* __restrict__ fpart, int HW, int C, int cpg, int G,
                                                            int silu, int nslb, int pstep, int ppb, int nblk)
{
    __shared__ float sh[256 * 17];
    const int tid = threadIdx.x, b = blockIdx.y, blk = blockIdx.x;
    const int sl = tid % nslb, pp = tid / nslb;
    const bool act = pp < pstep;
    const int ch0 = (blockIdx.z * nslb + sl) * 8;
    float a[8], c[8], mu[8], rs[8], m1[8], m2[8], fs[8], s1[8], s2[8];
    const float* abf = (const float*)(ab + (size_t)b * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ab_of(abf, ch0 + e, a[e], c[e]);
        const int g = (ch0 + e) / cpg;
        const float2 st = stats[b * G + g], gs = gstat[b * G + g];
        mu[e] = st.x; rs[e] = st.y; m1[e] = gs.x; m2[e] = gs.y;
        fs[e] = film ? 1.0f + film[(size_t)b * film_bstride + ch0 + e] : 1.0f;
        s1[e] = 0.f; s2[e] = 0.f;
    }
    const int pend = min(HW, (blk + 1) * ppb);
    if (act)
        for (int p0 = blk * ppb + pp; p0 < pend; p0 += 2 * pstep) {       // two pixels' loads in flight
            float xv[2][8], dv[2][8], rv[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = p0 + u * pstep;
                if (p < pend) {
                    const size_t off = ((size_t)b * HW + p) * C + ch0;
                    Chunk<T>::load(x + off, xv[u]); Chunk<T>::load(dA + off, dv[u]);
                    if (addend) Chunk<T>::load(addend + off, rv[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = p0 + u * pstep;
                if (p >= pend) break;
                float ov[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float da = dv[u][e];
                    if (silu) da *= dsilu<T>(fmaf(xv[u][e], a[e], c[e]));
                    const float xh = (xv[u][e] - mu[e]) * rs[e];
                    const float dx = a[e] * da - rs[e] * fmaf(xh, m2[e], m1[e]);
                    float o = dx * fs[e];
                    if (addend) o += rv[u][e];
                    ov[e] = o;
                    s1[e] += film ? dx : o; s2[e] = fmaf(dx, xv[u][e], s2[e]);
                }
                Chunk<T>::store(out + ((size_t)b * HW + p) * C + ch0, ov);
            }
        }
    if (!fpart) return;
    block_reduce16(sh, tid, sl, pp, nslb, pstep, s1, s2);
    if (pp == 0) {
        float2* o = fpart + ((size_t)b * nblk + blk) * C + ch0;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = make_float2(s1[e], s2[e]);
    }
}
hipError_t launch_gn_bwd_apply(int dtype, const void* x, const void* dA, const float2* ab, const float2* stats, const float2* gstat,
                               const void* addend, void* out, const float* film, int film_bstride, float2* fpart, int B, int HW, int C,
                               int cpg, int G, int silu, hipStream_t s)
{
    if (C % 8) return hipErrorInvalidValue;
    const GnBwdGeom g = gn_bwd_geom(dtype, B, HW, C);
    const dim3 grid(g.nblk, B, g.zblocks);
    if (dtype == 0) hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)dA, ab, stats, gstat, (const float*)addend, (float*)out, film, film_bstride, fpart, HW, C, cpg, G, silu, g.nslb, g.pstep, g.ppb, g.nblk);
    else hipLaunchKernelGGL(gn_bwd_apply_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)x, (const __bf16*)dA, ab, stats, gstat, (const __bf16*)addend, (__bf16*)out, film, film_bstride, fpart, HW, C, cpg, G, silu, g.nslb, g.pstep, g.ppb, g.nblk);
    return hipGetLastError();
}

// d scale[b][c] = sum dF * y1, y1 = (F - shift) / (1 + scale); d shift[b][c] = sum dF   (F = y1 (1 + scale) + shift is what the
// forward stored; a block whose 1 + scale is exactly zero has lost y1 and gets d scale = 0)
__global__ __launch_bounds__(256) void film_bwd_finalize_kernel(const float2* __restrict__ fpart, int nblk, const float* __restrict__ film,
                                                                 int film_bstride, float* __restrict__ dfilm, int C, float* __restrict__ dbias)
{
    __shared__ float red[16][16][2];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl, b = blockIdx.y;
    float p1 = 0.f, p2 = 0.f;
    if (c < C)
        for (int k = rl; k < nblk; k += 16) { const float2 v = fpart[((size_t)b * nblk + k) * C + c]; p1 += v.x; p2 += v.y; }
    red[rl][cl][0] = p1; red[rl][cl][1] = p2;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s1 += (double)red[q][cl][0]; s2 += (double)red[q][cl][1]; }
    const float sc = 1.0f + film[(size_t)b * film_bstride + c], sft = film[(size_t)b * film_bstride + C + c];
    dfilm[(size_t)b * film_bstride + c] = sc != 0.f ? (float)((s2 - (double)sft * s1) / (double)sc) : 0.f;
    dfilm[(size_t)b * film_bstride + C + c] = (float)s1;
    // the conv in front of the FiLM: d bias[c] = sum over pixels of dF (1 + scale)
    if (dbias) atomicAdd(dbias + c, (float)((double)sc * s1));
}
hipError_t launch_film_bwd_finalize(const float2* fpart, int nblk, const float* film, int film_bstride, float* dfilm, float* dbias, int B, int C,
                                    hipStream_t s)
{
    hipLaunchKernelGGL(film_bwd_finalize_kernel, dim3((C + 15) / 16, B), dim3(256), 0, s, fpart, nblk, film, film_bstride, dfilm, C, dbias);
    return hipGetLastError();
}

// ---- bias gradients -------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ scratch, int HW, int C, int nslb, int pstep,
                                                      int ppb, int nblk)
{
    __shared__ float sh[256 * 17];
    const int tid = threadIdx.x, b = blockIdx.y, blk = blockIdx.x;
    const int sl = tid % nslb, pp = tid / nslb;
    const int ch0 = (blockIdx.z * nslb + sl) * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const int pend = min(HW, (blk + 1) * ppb);
    if (pp < pstep)
        for (int p = blk * ppb + pp; p < pend; p += pstep) {
            float dv[8];
            Chunk<T>::load(dy + ((size_t)b * HW + p) * C + ch0, dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) s1[e] += dv[e];
        }
    block_reduce16(sh, tid, sl, pp, nslb, pstep, s1, s2);
    if (pp == 0) {
        float* o = scratch + ((size_t)b * nblk + blk) * C + ch0;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = s1[e];
    }
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ scratch, int rows, int C, float* __restrict__ db)
{
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < C)
        for (int k = blockIdx.y * 16 + rl; k < rows; k += 16 * gridDim.y) s += scratch[(size_t)k * C + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += (double)red[q][cl];
        atomicAdd(db + c, (float)t);
    }
}
// the same from the (sum, -) pairs the GroupNorm-backward apply pass leaves behind
__global__ __launch_bounds__(256) void colsum_pair_finalize_kernel(const float2* __restrict__ part, int rows, int C, float* __restrict__ db)
{
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < C)
        for (int k = blockIdx.y * 16 + rl; k < rows; k += 16 * gridDim.y) s += part[(size_t)k * C + c].x;
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += (double)red[q][cl];
        atomicAdd(db + c, (float)t);
    }
}
hipError_t launch_colsum_from_pairs(const float2* part, int rows, int C, float* db, hipStream_t s)
{
    const int chunks = rows >= 512 ? 8 : (rows >= 64 ? 4 : 1);
    hipLaunchKernelGGL(colsum_pair_finalize_kernel, dim3((C + 15) / 16, chunks), dim3(256), 0, s, part, rows, C, db);
    return hipGetLastError();
}
hipError_t launch_colsum(int dtype, const void* dy, float* scratch, float* db, int B, int HW, int C, hipStream_t s)
{
    if (C % 8) return hipErrorInvalidValue;
    const GnBwdGeom g = gn_bwd_geom(dtype, B, HW, C);
    const dim3 grid(g.nblk, B, g.zblocks);
    if (dtype == 0) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, scratch, HW, C, g.nslb, g.pstep, g.ppb, g.nblk);
    else hipLaunchKernelGGL(colsum_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)dy, scratch, HW, C, g.nslb, g.pstep, g.ppb, g.nblk);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 15) / 16, B * g.nblk >= 512 ? 8 : (B * g.nblk >= 64 ? 4 : 1)), dim3(256), 0, s, scratch, B * g.nblk, C, db);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void nchw_chansum_kernel(const float* __restrict__ x, float* __restrict__ db, int C, int64_t hw, int chunks)
{
    __shared__ double red[4];
    const int c = blockIdx.x, b = blockIdx.y / chunks, ck = blockIdx.y % chunks, tid = threadIdx.x;
    const int64_t per = (hw + chunks - 1) / chunks, i0 = ck * per, i1 = i0 + per < hw ? i0 + per : hw;
    const float* p = x + ((size_t)b * C + c) * hw;
    double s = 0.0;
    for (int64_t i = i0 + tid; i < i1; i += 256) s += (double)p[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) atomicAdd(db + c, (float)((red[0] + red[1]) + (red[2] + red[3])));
}
hipError_t launch_nchw_chansum(const float* x, float* db, int B, int C, int64_t hw, hipStream_t s)
{
    const int chunks = (int)((hw + 8191) / 8192);
    hipLaunchKernelGGL(nchw_chansum_kernel, dim3(C, B * chunks), dim3(256), 0, s, x, db, C, hw, chunks);
    return hipGetLastError();
}

// ---- weight gradients -------------------------------------------------------------------------------------------------------
// Workgroup = (Cout tile of 32*WN, Cin tile of 32*WK, parity, pixel split); 4 waves, each a 32 x 32 (co, ci) tile for all NTAPS
// taps (16 accumulator registers per tap).  Per pixel tile (4 rows x 32 pixels of the conv's M-space): stage dY [128][NT] and
// the activated input halo [ROWS][PITCH][KT] as fp32 in LDS, then for every pixel pair one LDS float per operand and one
// v_mfma_f32_32x32x2_f32 per tap (k = the two pixels).
template <int IS> struct WgGeom {
    static constexpr int ROWS = IS * 3 + 3, PITCH = IS == 1 ? 34 : 66;
};
template <typename T, int IS, int NTAPS, int WN, int WK, bool SILU = true>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgArgs a)
{
    constexpr int NT = 32 * WN, KT = 32 * WK, EPC = Vec16<T>::EPC;
    constexpr int ROWS = WgGeom<IS>::ROWS, PITCH = WgGeom<IS>::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const As = (float*)smem;                                   // [ROWS*PITCH][KT]
    float* const Ds = As + ROWS * PITCH * KT;                          // [128][NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wn = wave / WK, wk = wave % WK;
    const int n_kt = (a.Cin + KT - 1) / KT, n_nt = (a.Cout + NT - 1) / NT;
    int bid = blockIdx.x;
    const int kt = bid % n_kt; bid /= n_kt;
    const int nt = bid % n_nt; bid /= n_nt;
    const int par = bid % a.npar; const int split = bid / a.npar;
    const int k0 = kt * KT, n0 = nt * NT, py = par >> 1, px = par & 1, par_off = par * 4;

    int toff[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) toff[t] = ((a.tapinfo_dy(par_off + t) + 1) * PITCH + a.tapinfo_dx(par_off + t) + 1) * KT;
    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const int tiles = a.B * a.n_ty * a.n_tx;
    for (int tile = split; tile < tiles; tile += a.nsplit) {
        const int tx = tile % a.n_tx, ty = (tile / a.n_tx) % a.n_ty, b = tile / (a.n_tx * a.n_ty);
        const int my0 = ty * 4, mx0 = tx * 32;
        __syncthreads();                                            // previous tile's reads are done
        {   // activated input halo
            constexpr int CH = KT / EPC;                              // 16-byte chunks per pixel
            const int ck = tid % CH, cbase = k0 + ck * EPC;
            const bool cvalid = cbase < a.Cin;
            GnCoef<T> gk;
            gk.load(a.gn_ab + (size_t)b * a.Cin + (cvalid ? cbase : 0), a.gn_ab != nullptr && cvalid);
            const int iy0 = IS * my0 - 1, ix0 = IS * mx0 - 1;
            for (int pxl = tid / CH; pxl < ROWS * PITCH; pxl += 256 / CH) {
                const int hy = pxl / PITCH, hx = pxl - hy * PITCH, iy = iy0 + hy, ix = ix0 + hx;
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (cvalid && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
                    v = *(const u32x4*)((const T*)a.x + ((size_t)(b * a.Hin + iy) * a.Win + ix) * a.Cin + cbase);
                    if (a.gn_ab) v = gk.template apply<SILU>(v);
                }
                float f[EPC];
                Vec16<T>::unpack(v, f);
#pragma unroll
                for (int e = 0; e < EPC; e += 4) *(f32x4*)(As + pxl * KT + ck * EPC + e) = f32x4{f[e], f[e + 1], f[e + 2], f[e + 3]};
            }
        }
        {   // output gradient tile
            constexpr int CH = NT / EPC;
            const int ck = tid % CH, nbase = n0 + ck * EPC;
            for (int m = tid / CH; m < 128; m += 256 / CH) {
                const int my = my0 + (m >> 5), mx = mx0 + (m & 31);
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (nbase < a.Cout && my < a.MH && mx < a.MW)
                    v = *(const u32x4*)((const T*)a.dy + ((size_t)(b * a.Hout + my * a.OS + py) * a.Wout + mx * a.OS + px) * a.Cout + nbase);
                float f[EPC];
                Vec16<T>::unpack(v, f);
#pragma unroll
                for (int e = 0; e < EPC; e += 4) *(f32x4*)(Ds + m * NT + ck * EPC + e) = f32x4{f[e], f[e + 1], f[e + 2], f[e + 3]};
            }
        }
        __syncthreads();
        const float* const Aw = As + wk * 32 + r;
        const float* const Dw = Ds + wn * 32 + r;
        for (int i = 0; i < 4; ++i)
#pragma unroll 4
            for (int jp = 0; jp < 16; ++jp) {
                const int j = 2 * jp + h;
                const float dv = Dw[(i * 32 + j) * NT];
                const float* ap = Aw + (IS * i * PITCH + IS * j) * KT;
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv, ap[toff[t]], acc[t], 0, 0, 0);
            }
    }
    // D: lane r = column (ci), register q = row (co) (q & 3) + 8 (q >> 2) + 4 h
    const int k = k0 + wk * 32 + r;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int wt = a.tapinfo_w(par_off + t);
        float* const o = a.part + ((size_t)split * a.taps_w + wt) * a.Cout * a.Cin;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = n0 + wn * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (n < a.Cout && k < a.Cin) o[(size_t)n * a.Cin + k] = acc[t][q];
        }
    }
}

// bf16 mode: the same decomposition on v_mfma_f32_32x32x16_bf16.  The contraction index is the PIXEL, while the tiles in LDS are
// channel-contiguous, so both operands are fetched with the transposing LDS read of gfx950 (ds_read_b64_tr_b16: a 16-lane group
// reads 4 pixels x 16 channels and each lane receives 4 consecutive pixels of one channel).  LDS images are planes of
// [pixel][32 channels] (64-byte rows): the four pixel rows of a transposed read then cover 64 distinct banks (conflict-free)
// without padding; a wave works on one (Cout plane, Cin plane) pair of the workgroup's 64 x 64 tile.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
template <int IS, int NTAPS, bool SILU = true>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WgArgs a)
{
    typedef __bf16 T;
    constexpr int ROWS = WgGeom<IS>::ROWS, PITCH = WgGeom<IS>::PITCH, NPA = ROWS * PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;                                   // [2 planes][NPA pixels][32 ch] bf16
    unsigned char* const Ds = smem + 2 * NPA * 64;                     // [2 planes][128 pixels][32 ch] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int n_kt = (a.Cin + 63) / 64, n_nt = (a.Cout + 63) / 64;
    int bid = blockIdx.x;
    const int kt = bid % n_kt; bid /= n_kt;
    const int nt = bid % n_nt; bid /= n_nt;
    const int par = bid % a.npar; const int split = bid / a.npar;
    const int k0 = kt * 64, n0 = nt * 64, py = par >> 1, px = par & 1, par_off = par * 4;

    int toff[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) toff[t] = ((a.tapinfo_dy(par_off + t) + 1) * PITCH + a.tapinfo_dx(par_off + t) + 1) * 64;
    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    // transposed-read geometry of this lane: group g of 16 lanes -> pixels 8*(g>>1) + (0..3), channels 16*(g&1) + (0..15)
    const int g = lane >> 4, pix_l = 8 * (g >> 1) + ((lane & 15) >> 2), c_l = 16 * (g & 1) + 4 * (lane & 3);
    const unsigned char* const Dp = Ds + ((wn * 128 + pix_l) * 32 + c_l) * 2;
    const unsigned char* const Ap = As + ((wk * NPA + IS * pix_l) * 32 + c_l) * 2;

    const int tiles = a.B * a.n_ty * a.n_tx;
    for (int tile = split; tile < tiles; tile += a.nsplit) {
        const int tx = tile % a.n_tx, ty = (tile / a.n_tx) % a.n_ty, b = tile / (a.n_tx * a.n_ty);
        const int my0 = ty * 4, mx0 = tx * 32;
        __syncthreads();
        {   // activated input halo (8 chunks of 8 channels per pixel) and the output-gradient tile: every global load is issued
            // before the first use, then transformed and written to LDS
            constexpr int NA = (NPA + 31) / 32;
            const int ck = tid & 7, cbase = k0 + ck * 8, nbase = n0 + ck * 8;
            const bool cvalid = cbase < a.Cin;
            GnCoef<T> gk;
            gk.load(a.gn_ab + (size_t)b * a.Cin + (cvalid ? cbase : 0), a.gn_ab != nullptr && cvalid);
            const int iy0 = IS * my0 - 1, ix0 = IS * mx0 - 1;
            u32x4 rd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = (tid >> 3) + 32 * i, my = my0 + (m >> 5), mx = mx0 + (m & 31);
                rd[i] = u32x4{0u, 0u, 0u, 0u};
                if (nbase < a.Cout && my < a.MH && mx < a.MW)
                    rd[i] = *(const u32x4*)((const T*)a.dy + ((size_t)(b * a.Hout + my * a.OS + py) * a.Wout + mx * a.OS + px) * a.Cout + nbase);
            }
            unsigned char* const dstA = As + ((ck >> 2) * NPA * 32 + (ck & 3) * 8) * 2;
            constexpr int GRP = NA < 8 ? NA : 8;
#pragma unroll
            for (int g0 = 0; g0 < NA; g0 += GRP) {
                u32x4 ra[GRP]; unsigned okm = 0;
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const int pxl = (tid >> 3) + 32 * (g0 + u);
                    const int hy = pxl / PITCH, hx = pxl - hy * PITCH, iy = iy0 + hy, ix = ix0 + hx;
                    ra[u] = u32x4{0u, 0u, 0u, 0u};
                    if (g0 + u < NA && pxl < NPA && cvalid && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
                        okm |= 1u << u;
                        ra[u] = *(const u32x4*)((const T*)a.x + ((size_t)(b * a.Hin + iy) * a.Win + ix) * a.Cin + cbase);
                    }
                }
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const int pxl = (tid >> 3) + 32 * (g0 + u);
                    if (g0 + u < NA && pxl < NPA) {
                        u32x4 v = ra[u];
                        if (((okm >> u) & 1u) && a.gn_ab) v = gk.template apply<SILU>(v);
                        *(u32x4*)(dstA + pxl * 64) = v;
                    }
                }
            }
            unsigned char* const dstD = Ds + ((ck >> 2) * 128 * 32 + (ck & 3) * 8) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(dstD + ((tid >> 3) + 32 * i) * 64) = rd[i];
        }
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < 8; ++ks) {                              // 16 pixels per step: row ks >> 1, columns 16 (ks & 1) ..
            const s16x4 d0 = lds_tr16(Dp + ks * 1024), d1 = lds_tr16(Dp + ks * 1024 + 256);
            const s16x8 dv = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
            const unsigned char* const ap = Ap + (IS * (ks >> 1) * PITCH + IS * 16 * (ks & 1)) * 64;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const s16x4 a0 = lds_tr16(ap + toff[t]), a1 = lds_tr16(ap + toff[t] + 256 * IS);
                const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, dv), __builtin_bit_cast(bf16x8, av), acc[t], 0, 0, 0);
            }
        }
    }
    const int k = k0 + wk * 32 + r;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int wt = a.tapinfo_w(par_off + t);
        float* const o = a.part + ((size_t)split * a.taps_w + wt) * a.Cout * a.Cin;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = n0 + wn * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (n < a.Cout && k < a.Cin) o[(size_t)n * a.Cin + k] = acc[t][q];
        }
    }
}
// Warp-specialised form for the stride-1 kinds: waves 0-3 only run the transposed reads + MFMAs, waves 4-7 stage the NEXT pixel
// tile into the other LDS buffer (global loads for tile j+2 are issued into registers while tile j+1 is transformed and written), one
// workgroup barrier per tile.  The single-role kernel above spends ~60 % of a tile's time in its staging phase.
template <int NTAPS, bool SILU>
__global__ __launch_bounds__(512) void wgrad_bf16_ws_kernel(const WgArgs a)
{
    typedef __bf16 T;
    constexpr int ROWS = WgGeom<1>::ROWS, PITCH = WgGeom<1>::PITCH, NPA = ROWS * PITCH, NA = (NPA + 31) / 32;
    constexpr int A_BYTES = 2 * NPA * 64, D_BYTES = 2 * 128 * 64, BUF = A_BYTES + D_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_kt = (a.Cin + 63) / 64, n_nt = (a.Cout + 63) / 64;
    // XCD-aware mapping: workgroup i runs on XCD i % 8 (round-robin dispatch) and every XCD has its own L2; the (Cout, Cin, parity)
    // groups of one pixel split read the same pixels, so they are given to the same XCD (nsplit is a multiple of 8 or the
    // plain order is used)
    int bid = blockIdx.x;
    const int ngrp = n_kt * n_nt * a.npar;
    if ((a.nsplit & 7) == 0) { const int xcd = bid & 7, idx = bid >> 3; bid = (xcd + 8 * (idx / ngrp)) * ngrp + idx % ngrp; }
    const int kt = bid % n_kt; bid /= n_kt;
    const int nt = bid % n_nt; bid /= n_nt;
    const int par = bid % a.npar; const int split = bid / a.npar;
    const int k0 = kt * 64, n0 = nt * 64, py = par >> 1, px = par & 1, par_off = par * 4;
    const int tiles = a.B * a.n_ty * a.n_tx;
    const int ntile = split < tiles ? (tiles - split + a.nsplit - 1) / a.nsplit : 0;

    if (wave >= 4) {
        // ---- producers ----------------------------------------------------------------------------------------------------
        const int ptid = tid - 256, ck = ptid & 7, cbase = k0 + ck * 8, nbase = n0 + ck * 8;
        const bool cvalid = cbase < a.Cin;
        // two register sets: the loads of tile j+3 are issued while tile j+1 is written to LDS, so every load has two tile periods to
        // land (with one set the loop ran at one tile per memory latency: ~45 KB in flight per CU / ~4 us = 2.9 TB/s chip-wide)
        struct PSet { u32x4 ra[NA], rd[4]; unsigned okm; GnCoef<T> gk; };
        PSet s0, s1;
        auto issue = [&](PSet& ps, int j) __attribute__((always_inline)) {
            const int tile = split + j * a.nsplit;
            const int tx = tile % a.n_tx, ty = (tile / a.n_tx) % a.n_ty, b = tile / (a.n_tx * a.n_ty);
            const int my0 = ty * 4, mx0 = tx * 32, iy0 = my0 - 1, ix0 = mx0 - 1;
            ps.gk.load(a.gn_ab + (size_t)b * a.Cin + (cvalid ? cbase : 0), a.gn_ab != nullptr && cvalid);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = (ptid >> 3) + 32 * i, my = my0 + (m >> 5), mx = mx0 + (m & 31);
                ps.rd[i] = u32x4{0u, 0u, 0u, 0u};
                if (nbase < a.Cout && my < a.MH && mx < a.MW)
                    ps.rd[i] = *(const u32x4*)((const T*)a.dy + ((size_t)(b * a.Hout + my * a.OS + py) * a.Wout + mx * a.OS + px) * a.Cout + nbase);
            }
            ps.okm = 0;
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int pxl = (ptid >> 3) + 32 * u;
                const int hy = pxl / PITCH, hx = pxl - hy * PITCH, iy = iy0 + hy, ix = ix0 + hx;
                ps.ra[u] = u32x4{0u, 0u, 0u, 0u};
                if (pxl < NPA && cvalid && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
                    ps.okm |= 1u << u;
                    ps.ra[u] = *(const u32x4*)((const T*)a.x + ((size_t)(b * a.Hin + iy) * a.Win + ix) * a.Cin + cbase);
                }
            }
        };
        auto commit = [&](PSet& ps, int buf) __attribute__((always_inline)) {
            unsigned char* const As = smem + buf * BUF;
            unsigned char* const dstA = As + ((ck >> 2) * NPA * 32 + (ck & 3) * 8) * 2;
            unsigned char* const dstD = As + A_BYTES + ((ck >> 2) * 128 * 32 + (ck & 3) * 8) * 2;
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int pxl = (ptid >> 3) + 32 * u;
                if (pxl < NPA) {
                    u32x4 v = ps.ra[u];
                    if (((ps.okm >> u) & 1u) && a.gn_ab) v = ps.gk.template apply<SILU>(v);
                    *(u32x4*)(dstA + pxl * 64) = v;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(dstD + ((ptid >> 3) + 32 * i) * 64) = ps.rd[i];
        };
        // tile j lives in set j & 1 and goes to LDS buffer j & 1
        if (ntile > 0) issue(s0, 0);
        if (ntile > 1) issue(s1, 1);
        if (ntile > 0) { commit(s0, 0); if (ntile > 2) issue(s0, 2); }
        __syncthreads();
        const bool idle = CCN_DBG_BIT(a, 1);
        for (int j = 0; j < ntile; j += 2) {
            if (j + 1 < ntile && !idle) { commit(s1, 1); if (j + 3 < ntile) issue(s1, j + 3); }
            __syncthreads();
            if (j + 1 >= ntile) break;
            if (j + 2 < ntile && !idle) { commit(s0, 0); if (j + 4 < ntile) issue(s0, j + 4); }
            __syncthreads();
        }
        return;
    }
    // ---- consumers ------------------------------------------------------------------------------------------------------------
    const int r = lane & 31, h = lane >> 5, wn = wave >> 1, wk = wave & 1;
    int toff[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) toff[t] = ((a.tapinfo_dy(par_off + t) + 1) * PITCH + a.tapinfo_dx(par_off + t) + 1) * 64;
    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    const int g = lane >> 4, pix_l = 8 * (g >> 1) + ((lane & 15) >> 2), c_l = 16 * (g & 1) + 4 * (lane & 3);
    const int d_off = A_BYTES + ((wn * 128 + pix_l) * 32 + c_l) * 2, a_off = ((wk * NPA + pix_l) * 32 + c_l) * 2;
    __syncthreads();
    // The 8 k-steps x NTAPS MFMAs of a tile are one straight-line software pipeline: the transposed reads of step s + PF are issued
    // before the MFMA of step s (a ring of RING fragment pairs), so an MFMA never waits for the LDS latency of its own operands
    // (issuing read, wait, MFMA per step ran at 1/4 of the MFMA rate).
    constexpr int NSTEP = 8 * NTAPS, RING = 8, PF = 6;
    for (int j = 0; j < ntile; ++j) {
        if (CCN_DBG_BIT(a, 2)) { __syncthreads(); continue; }
        const unsigned char* const base = smem + (j & 1) * BUF;
        const unsigned char* const Dp = base + d_off;
        const unsigned char* apt[NTAPS];
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) apt[t] = base + a_off + toff[t];
        s16x4 dq[2][2], aq[RING][2];
        dq[0][0] = lds_tr16(Dp); dq[0][1] = lds_tr16(Dp + 256);
#pragma unroll
        for (int s0 = 0; s0 < PF; ++s0) {
            constexpr int dummy = 0; (void)dummy;
            const int ks = s0 / NTAPS, t = s0 % NTAPS;
            const int koff = ((ks >> 1) * PITCH + 16 * (ks & 1)) * 64;
            aq[s0 % RING][0] = lds_tr16(apt[t] + koff); aq[s0 % RING][1] = lds_tr16(apt[t] + koff + 256);
        }
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int ks = st / NTAPS, t = st % NTAPS;
            if (t == 0 && ks + 1 < 8) { dq[(ks + 1) & 1][0] = lds_tr16(Dp + (ks + 1) * 1024); dq[(ks + 1) & 1][1] = lds_tr16(Dp + (ks + 1) * 1024 + 256); }
            const int nx = st + PF;
            if (nx < NSTEP) {
                const int ks2 = nx / NTAPS, t2 = nx % NTAPS;
                const int koff = ((ks2 >> 1) * PITCH + 16 * (ks2 & 1)) * 64;
                aq[nx % RING][0] = lds_tr16(apt[t2] + koff); aq[nx % RING][1] = lds_tr16(apt[t2] + koff + 256);
            }
            const s16x4 d0 = dq[ks & 1][0], d1 = dq[ks & 1][1], a0 = aq[st % RING][0], a1 = aq[st % RING][1];
            const s16x8 dv = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
            const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, dv), __builtin_bit_cast(bf16x8, av), acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                    // keep the issue order: the scheduler otherwise sinks the reads next to their use
        }
        __syncthreads();
    }
    if (CCN_DBG_BIT(a, 4)) return;
    const int k = k0 + wk * 32 + r;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int wt = a.tapinfo_w(par_off + t);
        float* const o = a.part + ((size_t)split * a.taps_w + wt) * a.Cout * a.Cin;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = n0 + wn * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (n < a.Cout && k < a.Cin) o[(size_t)n * a.Cin + k] = acc[t][q];
        }
    }
}
static constexpr size_t wgrad_bf16_ws_lds() { return (size_t)2 * (2 * WgGeom<1>::ROWS * WgGeom<1>::PITCH + 2 * 128) * 64; }

template <int IS> static constexpr size_t wgrad_bf16_lds() { return (size_t)(2 * WgGeom<IS>::ROWS * WgGeom<IS>::PITCH + 2 * 128) * 64; }

template <int IS, int WN, int WK> static constexpr size_t wgrad_lds() { return (size_t)(WgGeom<IS>::ROWS * WgGeom<IS>::PITCH * 32 * WK + 128 * 32 * WN) * 4; }
typedef void (*wgrad_fn_t)(const WgArgs);
static wgrad_fn_t wgrad_pick(int dtype, int kind, size_t* lds, int* nt, int* kt, bool silu = true)
{
    if (dtype == 1) {
        *nt = 64; *kt = 64;
        static const bool no_ws = diag_env("CCN_WGRAD_NO_WS") != nullptr;         // A/B switch: single-role kernel everywhere
        if (!no_ws) {
            if (!silu && kind == KIND_C3S1) { *lds = wgrad_bf16_ws_lds(); return (wgrad_fn_t)wgrad_bf16_ws_kernel<9, false>; }
            if (kind == KIND_C3S1) { *lds = wgrad_bf16_ws_lds(); return (wgrad_fn_t)wgrad_bf16_ws_kernel<9, true>; }
            if (kind == KIND_CT4) { *lds = wgrad_bf16_ws_lds(); return (wgrad_fn_t)wgrad_bf16_ws_kernel<4, true>; }
        }
        if (!silu && kind == KIND_C3S1) { *lds = wgrad_bf16_lds<1>(); return (wgrad_fn_t)wgrad_bf16_kernel<1, 9, false>; }
        switch (kind) {
            case KIND_C3S2: *lds = wgrad_bf16_lds<2>(); return (wgrad_fn_t)wgrad_bf16_kernel<2, 9>;
            case KIND_CT4: *lds = wgrad_bf16_lds<1>(); return (wgrad_fn_t)wgrad_bf16_kernel<1, 4>;
            case KIND_STEM: *lds = wgrad_bf16_lds<1>(); return (wgrad_fn_t)wgrad_bf16_kernel<1, 1>;      // 1x1: im2col'ed stem
            default: *lds = wgrad_bf16_lds<1>(); return (wgrad_fn_t)wgrad_bf16_kernel<1, 9>;
        }
    }
    if (!silu && kind == KIND_C3S1) { *lds = wgrad_lds<1, 2, 2>(); *nt = 64; *kt = 64; return (wgrad_fn_t)wgrad_kernel<float, 1, 9, 2, 2, false>; }
    switch (kind) {
        case KIND_C3S2: *lds = wgrad_lds<2, 4, 1>(); *nt = 128; *kt = 32; return (wgrad_fn_t)wgrad_kernel<float, 2, 9, 4, 1>;
        case KIND_CT4: *lds = wgrad_lds<1, 2, 2>(); *nt = 64; *kt = 64; return (wgrad_fn_t)wgrad_kernel<float, 1, 4, 2, 2>;
        case KIND_STEM: *lds = wgrad_lds<1, 2, 2>(); *nt = 64; *kt = 64; return (wgrad_fn_t)wgrad_kernel<float, 1, 1, 2, 2>;
        default: *lds = wgrad_lds<1, 2, 2>(); *nt = 64; *kt = 64; return (wgrad_fn_t)wgrad_kernel<float, 1, 9, 2, 2>;
    }
}
hipError_t wgrad_prepare()
{
    static const int kinds[] = {KIND_C3S1, KIND_C3S2, KIND_CT4, KIND_STEM};
    for (int dt = 0; dt < 2; ++dt)
        for (int kind : kinds) {
            size_t lds; int nt, kt;
            for (int silu = 0; silu < 2; ++silu) {
                const wgrad_fn_t fn = wgrad_pick(dt, kind, &lds, &nt, &kt, silu != 0);
                hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
            }
        }
    return hipSuccess;
}
int wgrad_nsplit(int dtype, int kind, int B, int MH, int MW, int Cin, int Cout, bool concurrent)
{
    size_t lds; int nt, kt;
    (void)wgrad_pick(dtype, kind, &lds, &nt, &kt);
    const int npar = kind == KIND_CT4 ? 4 : 1;
    const int tiles = B * ((MH + 3) / 4) * ((MW + 31) / 32);
    const int groups = ((Cin + kt - 1) / kt) * ((Cout + nt - 1) / nt) * npar;
    static const int target_bf16 = diag_env("CCN_WGRAD_WGS") ? atoi(diag_env("CCN_WGRAD_WGS")) : 512;
    const bool ws = dtype == 1 && lds == wgrad_bf16_ws_lds();
    // one workgroup per CU when the kernel has the GPU to itself; when it runs on the side stream next to the data-gradient chain,
    // ~100 workgroups: measured 6.46 ms per step at 96, 6.51 at 128, 7.08 at 256 (the main-stream kernels get the other CUs)
    static const int env_ws = diag_env("CCN_WGRAD_WS_WGS") ? atoi(diag_env("CCN_WGRAD_WS_WGS")) : 0;
    const int target_ws = env_ws > 0 ? env_ws : (concurrent ? 96 : 256);
    int ns = ((ws ? target_ws : (dtype == 1 ? target_bf16 : 768)) + groups - 1) / groups;     // 1 (warp-specialised) or 2-3 workgroups per CU
    if (ns > tiles) ns = tiles;
    if (ns >= 8) ns &= ~7;                                        // whole XCD rounds (see the kernels' block mapping)
    return ns < 1 ? 1 : ns;
}
hipError_t launch_wgrad(int dtype, int kind, const WgArgs& a, hipStream_t s)
{
    size_t lds; int nt, kt;
    if (a.gn_ab && !a.silu && kind != KIND_C3S1) return hipErrorInvalidValue;       // GroupNorm without SiLU only exists in front of the head
    const wgrad_fn_t fn = wgrad_pick(dtype, kind, &lds, &nt, &kt, !a.gn_ab || a.silu != 0);
    const int epc = dtype == 0 ? 4 : 8;
    if (a.Cin % epc || a.Cout % epc) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(((a.Cin + kt - 1) / kt) * ((a.Cout + nt - 1) / nt) * a.npar * a.nsplit);
    const bool ws = dtype == 1 && lds == wgrad_bf16_ws_lds();
    static const int dbg = diag_env("CCN_WG_DBG") ? atoi(diag_env("CCN_WG_DBG")) : 0;
    WgArgs d = a; d.dbg = dbg;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(ws ? 512 : 256), lds, s, d);
    return hipGetLastError();
}

// small layers (few (o, i) pairs, many splits): thread -> one element, eight splits in flight
__global__ void wgrad_reduce_elem_kernel(const float* __restrict__ part, int nsplit, int taps, int O, int I, int Ov, int Iv, int transposed,
                                    float* __restrict__ grad)
{
    const size_t per = (size_t)taps * O * I;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % I), o = (int)((idx / I) % O), t = (int)(idx / ((size_t)I * O));
        if (i >= Iv || o >= Ov) continue;
        // fixed order, eight loads in flight
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= nsplit; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * per + idx];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; k < nsplit; ++k) s += part[(size_t)k * per + idx];
        const size_t dst = transposed ? ((size_t)i * Ov + o) * taps + t : ((size_t)o * Iv + i) * taps + t;
        grad[dst] += s;
    }
}
// thread -> one (o, i) pair and all its taps: the partial reads are coalesced per tap plane, the gradient writes are the pair's
// contiguous taps (one thread = 36 or 64 bytes, neighbours adjacent) -- per-tap threads wrote 4 bytes at a 36-byte stride
template <int TAPS>
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, int O, int I, int Ov, int Iv, int transposed,
                                    float* __restrict__ grad)
{
    const size_t plane = (size_t)O * I, per = plane * TAPS;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < plane; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % I), o = (int)(idx / I);
        if (i >= Iv || o >= Ov) continue;
        float s[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) s[t] = 0.f;
        for (int k = 0; k < nsplit; ++k) {                          // fixed order; TAPS loads in flight
            const float* p = part + (size_t)k * per + idx;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) s[t] += p[(size_t)t * plane];
        }
        float* dst = grad + (transposed ? ((size_t)i * Ov + o) : ((size_t)o * Iv + i)) * TAPS;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) dst[t] += s[t];
    }
}
hipError_t launch_wgrad_reduce(const float* part, int nsplit, int taps, int O, int I, int Ov, int Iv, int transposed, float* grad, hipStream_t s)
{
    const size_t plane = (size_t)O * I;
    if (plane < (size_t)200000) {                                  // not enough pairs to fill the GPU: one thread per element
        const size_t per = plane * taps;
        const unsigned ge = (unsigned)((per + 255) / 256 < 16384 ? (per + 255) / 256 : 16384);
        hipLaunchKernelGGL(wgrad_reduce_elem_kernel, dim3(ge ? ge : 1), dim3(256), 0, s, part, nsplit, taps, O, I, Ov, Iv, transposed, grad);
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)((plane + 255) / 256 < 16384 ? (plane + 255) / 256 : 16384);
    switch (taps) {
        case 9: hipLaunchKernelGGL(wgrad_reduce_kernel<9>, dim3(grid ? grid : 1), dim3(256), 0, s, part, nsplit, O, I, Ov, Iv, transposed, grad); break;
        case 16: hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(grid ? grid : 1), dim3(256), 0, s, part, nsplit, O, I, Ov, Iv, transposed, grad); break;
        case 1: hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(grid ? grid : 1), dim3(256), 0, s, part, nsplit, O, I, Ov, Iv, transposed, grad); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// im2col of the NCHW fp32 image: dst[b][y][x][k], k = ci*9 + (dy+1)*3 + (dx+1) for k < C*9, zero up to 32 columns
template <typename T>
__global__ void im2col27_kernel(const float* __restrict__ x, T* __restrict__ dst, int C, int H, int W)
{
    const int b = blockIdx.y;
    const size_t n = (size_t)H * W * 32;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx & 31); const size_t p = idx >> 5;
        const int xx = (int)(p % W), yy = (int)(p / W);
        float v = 0.f;
        if (k < C * 9) {
            const int ci = k / 9, t = k - ci * 9, iy = yy + t / 3 - 1, ix = xx + t % 3 - 1;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((size_t)(b * C + ci) * H + iy) * W + ix];
        }
        dst[(size_t)b * n + idx] = to_elem<T>(v);
    }
}
hipError_t launch_im2col27(int dtype, const float* x, void* dst, int B, int C, int H, int W, hipStream_t s)
{
    if (C * 9 > 32) return hipErrorInvalidValue;
    const size_t n = (size_t)H * W * 32;
    const dim3 grid((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), B);
    if (dtype == 0) hipLaunchKernelGGL(im2col27_kernel<float>, grid, dim3(256), 0, s, x, (float*)dst, C, H, W);
    else hipLaunchKernelGGL(im2col27_kernel<__bf16>, grid, dim3(256), 0, s, x, (__bf16*)dst, C, H, W);
    return hipGetLastError();
}
template <typename T>
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, T* __restrict__ dst, int C, int Cp, int64_t hw)
{
    const int b = blockIdx.y;
    const int64_t n = hw * Cp;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cp); const int64_t p = idx / Cp;
        dst[(size_t)b * n + idx] = to_elem<T>(c < C ? x[((size_t)b * C + c) * hw + p] : 0.f);
    }
}
hipError_t launch_nchw_to_nhwc_pad(int dtype, const float* x, void* dst, int B, int C, int Cp, int64_t hw, hipStream_t s)
{
    const int64_t n = hw * Cp;
    const dim3 grid((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), B);
    if (dtype == 0) hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<float>, grid, dim3(256), 0, s, x, (float*)dst, C, Cp, hw);
    else hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<__bf16>, grid, dim3(256), 0, s, x, (__bf16*)dst, C, Cp, hw);
    return hipGetLastError();
}

// ---- small fp32 linears ---------------------------------------------------------------------------------------------------
// y[r][n] = act(u), u = sum_k x[r][k] W[n][k] + b[n]; one wave per n.  The weight row is read ONCE for up to eight rows r, sixteen bytes per
// lane and load, the loads of a row independent of each other (one scalar load in flight per row pass left time_proj.2 -- K = 2048, 512
// waves -- at 57 us; conditioning is the first thing of the training step and nothing overlaps it)
__global__ __launch_bounds__(256) void tlinear_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ y, int ldy, float* __restrict__ u, int R, int K, int N, int silu)
{
    constexpr int RM = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const float* wrow = W + (size_t)n * K;
    const bool vec = (K & 3) == 0 && (ldx & 3) == 0 && (((size_t)x | (size_t)W) & 15) == 0;
    for (int r0 = 0; r0 < R; r0 += RM) {
        float acc[RM];
#pragma unroll
        for (int j = 0; j < RM; ++j) acc[j] = 0.f;
        if (vec) {
            for (int k = lane * 4; k < K; k += 256) {
                const f32x4 w4 = *(const f32x4*)(wrow + k);
#pragma unroll
                for (int j = 0; j < RM; ++j) {
                    if (r0 + j >= R) break;
                    const f32x4 x4 = *(const f32x4*)(x + (size_t)(r0 + j) * ldx + k);
                    acc[j] = fmaf(x4[0], w4[0], fmaf(x4[1], w4[1], fmaf(x4[2], w4[2], fmaf(x4[3], w4[3], acc[j]))));
                }
            }
        } else {
            for (int k = lane; k < K; k += 64) {
                const float w1 = wrow[k];
#pragma unroll
                for (int j = 0; j < RM; ++j) { if (r0 + j >= R) break; acc[j] = fmaf(x[(size_t)(r0 + j) * ldx + k], w1, acc[j]); }
            }
        }
#pragma unroll
        for (int j = 0; j < RM; ++j) {
            if (r0 + j >= R) break;
            float a = acc[j];
#pragma unroll
            for (int s2 = 32; s2 >= 1; s2 >>= 1) a += __shfl_xor(a, s2);
            if (lane == 0) {
                float v = a + (bias ? bias[n] : 0.f);
                if (u) u[(size_t)(r0 + j) * ldy + n] = v;
                if (silu) v = v / (1.0f + expf(-v));
                y[(size_t)(r0 + j) * ldy + n] = v;
            }
        }
    }
}
hipError_t launch_tlinear_fwd(const float* x, int ldx, const float* W, const float* b, float* y, int ldy, float* u, int R, int K, int N, int silu,
                              hipStream_t s)
{
    hipLaunchKernelGGL(tlinear_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, ldx, W, b, y, ldy, u, R, K, N, silu);
    return hipGetLastError();
}
__global__ void tlinear_dw_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx, float* __restrict__ dW,
                                  float* __restrict__ db, int R, int K, int N)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * K) return;
    const int k = (int)(idx % K), n = (int)(idx / K);
    float acc = 0.f, sb = 0.f;
    for (int r = 0; r < R; ++r) { const float d = dy[(size_t)r * lddy + n]; acc = fmaf(d, x[(size_t)r * ldx + k], acc); sb += d; }
    dW[idx] += acc;
    if (k == 0 && db) db[n] += sb;
}
hipError_t launch_tlinear_dw(const float* dy, int lddy, const float* x, int ldx, float* dW, float* db, int R, int K, int N, hipStream_t s)
{
    hipLaunchKernelGGL(tlinear_dw_kernel, dim3((unsigned)(((size_t)N * K + 255) / 256)), dim3(256), 0, s, dy, lddy, x, ldx, dW, db, R, K, N);
    return hipGetLastError();
}
// dx[r][k] = sum_n dy[r][n] W[n][k]: block = 64 k x 4 waves over n, 4 rows r per block share every weight load
__global__ __launch_bounds__(256) void tlinear_dx_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ W, float* __restrict__ dx,
                                                          int lddx, int R, int K, int N, int accumulate)
{
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane, r0 = blockIdx.y * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int kk = k < K ? k : K - 1;
#pragma unroll 4
    for (int n = wave; n < N; n += 4) {
        const float w = W[(size_t)n * K + kk];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int r = r0 + j < R ? r0 + j : R - 1; acc[j] = fmaf(dy[(size_t)r * lddy + n], w, acc[j]); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    if (wave == 0 && k < K)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + j < R) {
                const float v = (red[0][j][lane] + red[1][j][lane]) + (red[2][j][lane] + red[3][j][lane]);
                float* o = dx + (size_t)(r0 + j) * lddx + k;
                *o = accumulate ? *o + v : v;
            }
}
hipError_t launch_tlinear_dx(const float* dy, int lddy, const float* W, float* dx, int lddx, int R, int K, int N, int accumulate, hipStream_t s)
{
    hipLaunchKernelGGL(tlinear_dx_kernel, dim3((K + 63) / 64, (R + 3) / 4), dim3(256), 0, s, dy, lddy, W, dx, lddx, R, K, N, accumulate);
    return hipGetLastError();
}
__global__ void silu_bwd_kernel(float* __restrict__ du, const float* __restrict__ dy, const float* __restrict__ u, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) du[i] = dy[i] * dsilu<float>(u[i]);
}
hipError_t launch_silu_bwd(float* du, const float* dy, const float* u, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, du, dy, u, n);
    return hipGetLastError();
}
__global__ void add2_kernel(float* __restrict__ y, const float* __restrict__ a, const float* __restrict__ b, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}
hipError_t launch_add2(float* y, const float* a, const float* b, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, a, b, n);
    return hipGetLastError();
}

// ---- loss and optimiser ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ eps, const float* __restrict__ target, int64_t n, float inv_n,
                                                           float* __restrict__ d_eps, float* __restrict__ scratch)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = eps[i] - target[i];
        s += (double)d * (double)d;
        if (d_eps) d_eps[i] = 2.0f * d * inv_n;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) scratch[blockIdx.x] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ void mse_final_kernel(const float* __restrict__ scratch, int nb, double inv_n, float* __restrict__ loss)
{
    // one wave, fixed order: lane l sums partials l, l + 64, ... in fp64, then a shuffle tree (a single thread walking the 1024 partials
    // took 46 us between the forward and the backward of the step)
    if (blockIdx.x) return;
    double s = 0.0;
    for (int i = (int)threadIdx.x; i < nb; i += 64) s += (double)scratch[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (threadIdx.x == 0) *loss = (float)(s * inv_n);
}
hipError_t launch_mse_loss_grad(const float* eps, const float* target, int64_t n, float* loss, float* d_eps, float* scratch, hipStream_t s)
{
    const int nb = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(256), 0, s, eps, target, n, (float)(1.0 / (double)n), d_eps, scratch);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, s, scratch, nb, 1.0 / (double)n, loss);
    return hipGetLastError();
}

template <bool ZERO>
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                             float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        if (ZERO) g[i] = 0.f;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
    }
}
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, int step,
                        hipStream_t s, bool zero_grad)
{
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (zero_grad) hipLaunchKernelGGL(adamw_kernel<true>, dim3(grid ? grid : 1), dim3(256), 0, s, p, (float*)g, m, v, n, lr, b1, b2, eps, wd, bc1, sqrtf(bc2));
    else hipLaunchKernelGGL(adamw_kernel<false>, dim3(grid ? grid : 1), dim3(256), 0, s, p, (float*)g, m, v, n, lr, b1, b2, eps, wd, bc1, sqrtf(bc2));
    return hipGetLastError();
}

}  // namespace ccn
