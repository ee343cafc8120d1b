// Micro-benchmark: plain streaming WRITE bandwidth (16-byte stores, 1 KiB contiguous per wave instruction), the roof of the stem
// kernel (134 MB NHWC bf16 output at C2 / batch 8).   hipcc -O3 --offload-arch=gfx950 write_bw.hip -o write_bw && ./write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill(u32x4* __restrict__ p, size_t n16, unsigned v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p[i] = u32x4{v, v + 1, v + 2, (unsigned)i};
}
__global__ __launch_bounds__(256) void fill_nt(u32x4* __restrict__ p, size_t n16, unsigned v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(u32x4{v, v + 1, v + 2, (unsigned)i}, p + i);
}
int main()
{
    const size_t bytes = (size_t)134217728;                       // 8 x 256 x 256 x 128 x 2
    u32x4* p; (void)hipMalloc(&p, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
        for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
            for (int w = 0; w < 2; ++w) { if (nt) fill_nt<<<grid, 256>>>(p, bytes / 16, 7u); else fill<<<grid, 256>>>(p, bytes / 16, 7u); }
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int r = 0; r < 10; ++r) { if (nt) fill_nt<<<grid, 256>>>(p, bytes / 16, 7u + r); else fill<<<grid, 256>>>(p, bytes / 16, 7u + r); }
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s grid %5d: %.1f us per 134 MB = %.2f TB/s\n", nt ? "nontemporal" : "plain      ", grid, ms * 100.0, bytes / (ms * 1e-4) / 1e12);
        }
    return 0;
}
