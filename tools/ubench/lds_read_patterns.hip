// Micro-benchmark: cost of ds_read_b128 under the address patterns of the persistent conv kernel's A-fragment reads.
// Lane (r = lane & 31, h = lane >> 5) reads 16 bytes of halo pixel (r + dx): LDS rows of 128 B per pixel, slice index
// (2 * kslice + h) XOR-swizzled by a function of the pixel column.  Four waves per workgroup (one per SIMD) read concurrently.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 lds_read_patterns.hip -o lds_read_patterns && ./lds_read_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__device__ __forceinline__ int addr_of(int lane, int dx, int ks)
{
    const int r = lane & 31, h = lane >> 5, hx = r + dx;
    if (PAT == 0) return lane * 16;                                                   // contiguous: the reference
    if (PAT == 1) return hx * 128 + ((((hx >> 1) & 6) << 4) + (((h ^ (hx >> 1)) & 1) << 4) ^ (ks << 5));   // the kernel's (round 2)
    if (PAT == 2) return hx * 128 + ((((2 * ks + h) ^ (hx & 7)) & 7) << 4);           // swizzle by the column itself
    if (PAT == 3) return hx * 128 + ((2 * ks + h) << 4);                              // no swizzle
    if (PAT == 4) return hx * 128 + ((((2 * ks + h) ^ ((hx >> 1) & 7)) & 7) << 4);    // swizzle by column pairs
    if (PAT == 5) return hx * 128 + ((((2 * ks + h) ^ ((hx >> 2) & 7)) & 7) << 4);    // swizzle by column quads
    return 0;
}

template <int PAT>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters)
{
    __shared__ u32x4 lds[4096];                                   // 64 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const unsigned char* base = (const unsigned char*)lds;
    int a[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a[dx][ks] = addr_of<PAT>(lane, dx, ks);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4 v[8];
                const int ad = a[dx][ks];
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4352\n ds_read_b128 %2, %8 offset:8704\n ds_read_b128 %3, %8 offset:13056\n"
                             "ds_read_b128 %4, %8 offset:17408\n ds_read_b128 %5, %8 offset:21760\n ds_read_b128 %6, %8 offset:26112\n ds_read_b128 %7, %8 offset:30464\n"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(ad) : "memory");
                acc ^= v[0] ^ v[7];
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

template <int PAT> static void run(const char* name, unsigned long long* out, float* sink)
{
    const int iters = 256;
    hipLaunchKernelGGL((k<PAT>), dim3(1), dim3(256), 0, 0, out, sink, iters); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<PAT>), dim3(1), dim3(256), 0, 0, out, sink, iters); (void)hipDeviceSynchronize();
    unsigned long long h[4]; (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-36s %.1f cycles per ds_read_b128 per wave (4 waves reading: %.1f B/clk per CU)\n", name, (double)h[0] / (iters * 96.0), 4.0 * 1024.0 / ((double)h[0] / (iters * 96.0)));
}

int main()
{
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 256 * 4);
    run<0>("contiguous (lane * 16)", out, sink);
    run<1>("kernel layout, round 2", out, sink);
    run<2>("swizzle by column", out, sink);
    run<3>("no swizzle", out, sink);
    run<4>("swizzle by column pair", out, sink);
    run<5>("swizzle by column quad", out, sink);
    return 0;
}
