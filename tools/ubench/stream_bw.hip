// Read bandwidth of a plain streaming reduction on MI355X as a function of grid size and loads in flight per thread:
// the yardstick for the HBM-bound passes (GroupNorm backward, pre-pass, stem / head).  hipcc --offload-arch=gfx950 -O3 stream_bw.hip -o stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ p, size_t n, unsigned* out)
{
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n; i += stride) { const u32x4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// contiguous chunk per block (like the GroupNorm passes): block b reads [b*per, (b+1)*per)
template <int U>
__global__ __launch_bounds__(256) void rd_chunk(const u32x4* __restrict__ p, size_t n, unsigned* out)
{
    unsigned acc = 0;
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < hi; i += 256) { const u32x4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// The GroupNorm-backward reduce pass's own mapping (thread = 8 channels of a pixel, 16 slices x 16 pixel lanes per block, 256 pixels
// per block, C = 128 bf16) with three amounts of math per element: 0 = xor only, 1 = unpack + two fmas, 2 = + SiLU derivative
__device__ __forceinline__ float blo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
template <int MATH, int U>
__global__ __launch_bounds__(256) void gn_like(const u32x4* __restrict__ x, const u32x4* __restrict__ d, int HW, float* out)
{
    const int tid = threadIdx.x, sl = tid & 15, pp = tid >> 4, b = blockIdx.y, blk = blockIdx.x;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const int p_lo = blk * 256, p_hi = p_lo + 256 < HW ? p_lo + 256 : HW;
    for (int p0 = p_lo + pp; p0 < p_hi; p0 += U * 16) {
        u32x4 xr[U], dr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int p = p0 + u * 16; if (p < p_hi) { const size_t off = ((size_t)b * HW + p) * 16 + sl; xr[u] = x[off]; dr[u] = d[off]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p0 + u * 16 >= p_hi) break;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MATH == 0) { s1[2 * q] += __uint_as_float(xr[u][q] ^ dr[u][q]); continue; }
                float xv[2] = {blo(xr[u][q]), bhi(xr[u][q])}, dv[2] = {blo(dr[u][q]), bhi(dr[u][q])};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float da = dv[k];
                    if (MATH == 2) { const float y = fmaf(xv[k], 1.01f, 0.02f), sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y)); da *= sg * (1.0f + y * (1.0f - sg)); }
                    s1[2 * q + k] += da; s2[2 * q + k] = fmaf(da, (xv[k] - 0.1f) * 0.9f, s2[2 * q + k]);
                }
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) t += s1[e] + s2[e];
    if (t == 1.2345f) out[0] = t;
}
template <int MATH, int U> float time_gn(const u32x4* x, const u32x4* d, int HW, int B, float* out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(HW / 256, B);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gn_like<MATH, U>), grid, dim3(256), 0, 0, x, d, HW, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gn_like<MATH, U>), grid, dim3(256), 0, 0, x, d, HW, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}
template <typename K> float time_it(K k, int grid, const u32x4* p, size_t n, unsigned* out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, p, n, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, p, n, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}
int main()
{
    unsigned* out; hipMalloc(&out, 4);
    for (size_t mb : {8, 67, 536}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        u32x4* p; hipMalloc(&p, bytes); hipMemset(p, 1, bytes);
        for (int grid : {512, 1024, 2048, 4096, 16384}) {
            const float a1 = time_it(rd<1>, grid, p, n, out), a4 = time_it(rd<4>, grid, p, n, out), a8 = time_it(rd<8>, grid, p, n, out);
            const float c4 = time_it(rd_chunk<4>, grid, p, n, out), c8 = time_it(rd_chunk<8>, grid, p, n, out);
            printf("%4zu MB grid %5d: strided U1 %.2f U4 %.2f U8 %.2f TB/s | chunked U4 %.2f U8 %.2f TB/s  (U4 %.1f us)\n", mb, grid,
                   bytes / a1 / 1e9, bytes / a4 / 1e9, bytes / a8 / 1e9, bytes / c4 / 1e9, bytes / c8 / 1e9, a4 * 1e3);
        }
        hipFree(p);
    }
    {
        const int HW = 65536, B = 4; const size_t bytes = (size_t)B * HW * 128 * 2;
        u32x4 *x, *d; hipMalloc(&x, bytes); hipMalloc(&d, bytes); hipMemset(x, 0x3c, bytes); hipMemset(d, 0x3c, bytes);
        printf("gn-like reduce, 2 x %zu MB, us per launch: math0 U4 %.1f U8 %.1f | math1 U4 %.1f U8 %.1f | math2 U2 %.1f U4 %.1f U8 %.1f\n", bytes >> 20,
               time_gn<0, 4>(x, d, HW, B, (float*)out), time_gn<0, 8>(x, d, HW, B, (float*)out), time_gn<1, 4>(x, d, HW, B, (float*)out), time_gn<1, 8>(x, d, HW, B, (float*)out),
               time_gn<2, 2>(x, d, HW, B, (float*)out), time_gn<2, 4>(x, d, HW, B, (float*)out), time_gn<2, 8>(x, d, HW, B, (float*)out));
    }
    return 0;
}
