// What does a wave64 VALU instruction cost on a gfx950 SIMD, by instruction kind and by the number of waves
// that share the SIMD?  Each wave runs REPS x 32 independent instructions of one kind (inline asm, 16
// independent destination registers, so no dependency stalls); the kernel is timed with s_memtime inside
// and hipEvents outside.  Reports shader cycles per wave-instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ void __launch_bounds__(1024) k_rate(float *out, unsigned long long *cyc, int reps, float seed) {
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = seed * 0.5f + i; }
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { p[i] = f2{a[2 * i], a[2 * i + 1]}; q[i] = f2{b[2 * i], b[2 * i + 1]}; }
    unsigned long long mask = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (KIND == 0) {          // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                R16(X)
#undef X
            } else if (KIND == 1) {   // v_pk_fma_f32 (8 per group -> count 8)
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(q[i & 7]));
                R16(X)
#undef X
            } else if (KIND == 2) {   // v_exp_f32
#define X(i) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                R16(X)
#undef X
            } else if (KIND == 3) {   // v_cmp_le_f32 -> SGPR pair
#define X(i) { unsigned long long m; asm volatile("v_cmp_le_f32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b[i])); mask ^= m; }
                R16(X)
#undef X
            } else if (KIND == 4) {   // v_cndmask_b32 with an SGPR mask
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(0x5555555555555555ull));
                R16(X)
#undef X
            } else if (KIND == 5) {   // v_add_f32 DPP
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
                R16(X)
#undef X
            } else if (KIND == 6) {   // v_mul_f32
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                R16(X)
#undef X
            } else if (KIND == 7) {   // v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(q[i & 7]));
                R16(X)
#undef X
            } else if (KIND == 8) {   // dependent chain of v_fma_f32 (latency)
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[0]) : "v"(b[i]));
                R16(X)
#undef X
            } else if (KIND == 9) {   // v_min_f32
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                R16(X)
#undef X
            } else if (KIND == 10) {  // v_rcp_f32
#define X(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                R16(X)
#undef X
            } else if (KIND == 11) {  // s_and_b64 (SALU)
#define X(i) asm volatile("s_and_b64 %0, %0, %1" : "+s"(mask) : "s"(0x5555555555555555ull) : "scc");
                R16(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(mask & 1);
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int cus) {
    const int reps = 2048;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)cus * 2048 * 4);
    (void)hipMalloc(&cyc, (size_t)cus * 32 * 8);
    for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: one workgroup of 4*wps waves per CU (two of 16 for 8)
        const int threads = wps == 8 ? 1024 : 256 * wps, blocks = wps == 8 ? 2 * cus : cus;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 16, 1.0f);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, reps, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * threads / 64);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto c : h) mean += (double)c;
        mean /= (double)h.size();
        const double insts = (double)reps * 32.0;
        printf("%-14s waves/SIMD %d: %6.2f shader cycles per instruction per wave, %5.2f per SIMD; wall %.3f ms -> %5.2f cycles per SIMD-instruction at 2.4 GHz\n",
               name, wps, mean / insts, mean / insts / wps, ms, ms * 1e-3 * 2.4e9 / (insts * wps));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    run<0>("v_fma_f32", cus);
    run<6>("v_mul_f32", cus);
    run<1>("v_pk_fma_f32", cus);
    run<7>("v_pk_mul_f32", cus);
    run<2>("v_exp_f32", cus);
    run<10>("v_rcp_f32", cus);
    run<9>("v_min_f32", cus);
    run<3>("v_cmp_le_f32", cus);
    run<4>("v_cndmask", cus);
    run<5>("v_add_dpp", cus);
    run<8>("fma chain", cus);
    run<11>("s_and_b64", cus);
    return 0;
}
