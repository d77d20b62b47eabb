// gather_fetch.hip — calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ on the compositing kernels' access
// pattern: lanes gather 64-byte records by index from an array far larger than L2 + Infinity
// Cache, with a known byte count, next to the pattern the microarchitecture guide calibrated (16 B per lane,
// coalesced streaming).  Run each counter set in its own pass:
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o p -- ./gather_fetch
// and compare the per-kernel counter with the "expected" bytes this program prints (tools/microbench/README in DESIGN §5).
//   k_stream16      : N16 x 16 B, lane-contiguous                      expected N16 * 16
//   k_gather64      : NG records, each lane reads all 64 B of ITS record as four 16-byte loads (the staging loads of
//                     k_render_fwd / k_render_bwd for payloads of 5..8 channels)          expected NG * 64
//   k_gather48      : the same with three loads (48 of the 64 bytes: <= 4 payload channels, the bench shape)
//                                                                     expected NG * 64 (whole lines) or NG * 48
//   k_gather64_hot  : the same gather from a 16 MB array (L2 / Infinity Cache resident)   expected ~0 from HBM
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(float4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
}
__global__ void k_stream16(const float4 *__restrict__ p, size_t n, float *out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int LOADS>
__global__ void k_gather(const float4 *__restrict__ rec, const uint32_t *__restrict__ idx, size_t n, float *out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 *R = rec + (size_t)idx[i] * 4;
#pragma unroll
        for (int k = 0; k < LOADS; ++k) { const float4 v = R[k]; acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t REC = (size_t)32 << 20;          // 32 Mi records x 64 B = 2 GiB
    const size_t HOT = (size_t)256 << 10;         // 256 Ki records = 16 MiB
    const size_t NG = (size_t)8 << 20;            // gathers per launch
    const size_t N16 = (size_t)64 << 20;          // 1 GiB streamed
    float4 *rec; uint32_t *idx, *idx_hot; float *out;
    CK(hipMalloc((void **)&rec, REC * 64));
    CK(hipMalloc((void **)&idx, NG * 4));
    CK(hipMalloc((void **)&idx_hot, NG * 4));
    CK(hipMalloc((void **)&out, 256));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, rec, REC * 4);
    // distinct records: a multiplicative permutation of [0, REC) (REC is a power of two, the multiplier is odd)
    uint32_t *h = (uint32_t *)malloc(NG * 4), *hh = (uint32_t *)malloc(NG * 4);
    for (size_t i = 0; i < NG; ++i) { h[i] = (uint32_t)((i * 2654435761ull + 12345ull) & (REC - 1)); hh[i] = (uint32_t)((i * 2654435761ull + 12345ull) & (HOT - 1)); }
    CK(hipMemcpy(idx, h, NG * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(idx_hot, hh, NG * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double expect, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-16s expected_bytes %.0f  best_ms %.4f  -> %.2f TB/s\n", name, expect, best, expect / (best * 1e-3) / 1e12);
    };
    const dim3 grid(256 * 8), block(256);
    timed("k_stream16", (double)N16 * 16, [&] { hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, rec, N16, out); });
    timed("k_gather<4>", (double)NG * 64, [&] { hipLaunchKernelGGL((k_gather<4>), grid, block, 0, 0, rec, idx, NG, out); });
    timed("k_gather<3>", (double)NG * 64, [&] { hipLaunchKernelGGL((k_gather<3>), grid, block, 0, 0, rec, idx, NG, out); });
    timed("k_gather<2>", (double)NG * 64, [&] { hipLaunchKernelGGL((k_gather<2>), grid, block, 0, 0, rec, idx, NG, out); });
    timed("k_gather<4>hot", 0.0, [&] { hipLaunchKernelGGL((k_gather<4>), grid, block, 0, 0, rec, idx_hot, NG, out); });
    printf("index reads per gather launch: %.0f bytes (streamed, 4 B per lane)\n", (double)NG * 4);
    return 0;
}
