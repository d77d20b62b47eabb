// What does it cost to write PART of a 64-byte line on gfx950?  N records at a 64-byte stride (1.2 GB, far
// beyond L2 + MALL), each thread writes the first K bytes of one record (lane-contiguous 16-byte stores
// for the 64-byte case; the other cases leave holes).  Reports time per pass and useful GB/s.
// build: hipcc --offload-arch=gfx950 -O2 -o partial_writes partial_writes.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WORDS16, int FIRST>   // 16-byte words written per record, starting at word FIRST
__global__ void __launch_bounds__(256) k_write(float4 *rec, size_t n, float v) {
    // 4 consecutive lanes own one record (lane&3 = word) -> full-record case is one contiguous run per wave
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t r = t >> 2;
    const int w = (int)(t & 3);
    if (r < n && w >= FIRST && w < FIRST + WORDS16) rec[r * 4 + w] = make_float4(v, v, v, v);
}
__global__ void __launch_bounds__(256) k_write8(float2 *rec, size_t n, float v) {   // 8 bytes per 64-byte record
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) rec[r * 8] = make_float2(v, v);
}

template <typename F>
static void timeit(const char *name, double useful_bytes, F launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); launch();
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    printf("%-44s %8.3f ms   %7.1f GB/s useful\n", name, ms, useful_bytes / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t n = (size_t)20 << 20;   // 20 M records x 64 B = 1.34 GB
    float4 *d;
    if (hipMalloc(&d, n * 64) != hipSuccess) return 1;
    (void)hipMemset(d, 0, n * 64);
    const unsigned blocks4 = (unsigned)((n * 4 + 255) / 256), blocks1 = (unsigned)((n + 255) / 256);
    timeit("64 of 64 bytes (words 0-3)", 64.0 * n, [&] { hipLaunchKernelGGL((k_write<4, 0>), dim3(blocks4), dim3(256), 0, 0, d, n, 1.f); });
    timeit("48 of 64 bytes (words 0-2)", 48.0 * n, [&] { hipLaunchKernelGGL((k_write<3, 0>), dim3(blocks4), dim3(256), 0, 0, d, n, 2.f); });
    timeit("32 of 64 bytes (words 0-1)", 32.0 * n, [&] { hipLaunchKernelGGL((k_write<2, 0>), dim3(blocks4), dim3(256), 0, 0, d, n, 3.f); });
    timeit("32 of 64 bytes (words 2-3)", 32.0 * n, [&] { hipLaunchKernelGGL((k_write<2, 2>), dim3(blocks4), dim3(256), 0, 0, d, n, 4.f); });
    timeit("16 of 64 bytes (word 0)", 16.0 * n, [&] { hipLaunchKernelGGL((k_write<1, 0>), dim3(blocks4), dim3(256), 0, 0, d, n, 5.f); });
    timeit(" 8 of 64 bytes", 8.0 * n, [&] { hipLaunchKernelGGL(k_write8, dim3(blocks1), dim3(256), 0, 0, (float2 *)d, n, 6.f); });
    return 0;
}
