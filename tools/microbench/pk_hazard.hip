// Does a packed-f32 result need a wait state before the NEXT instruction reads it on gfx950?
// The compiler puts `s_nop 0` between v_pk_mul_f32 / v_pk_fma_f32 and an immediately following VALU reader
// (its "dst_sel forwarding" hazard check keys on a modifier bit that every packed-f32 instruction has set by
// default: op_sel_hi of source 0).  This test issues the dependent instructions back to back from inline asm
// (which the hazard recognizer does not look into) on registers preloaded with junk, and compares with the same
// arithmetic done by the compiler.  Any stale read shows up as a mismatch.
// build: hipcc --offload-arch=gfx950 -O2 -o pk_hazard pk_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_test(const float *in, float *out_asm, float *out_ref, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    float r0, r1, r2, r3;
    asm volatile(
        "v_mov_b32 v20, %4\n\t v_mov_b32 v21, %5\n\t v_mov_b32 v22, %6\n\t v_mov_b32 v23, %7\n\t"
        "v_mov_b32 v24, 0x7fc00000\n\t v_mov_b32 v25, 0x7fc00000\n\t v_mov_b32 v26, 0x7fc00000\n\t v_mov_b32 v27, 0x7fc00000\n\t"
        "v_mov_b32 v28, 0x7fc00000\n\t v_mov_b32 v29, 0x7fc00000\n\t v_mov_b32 v30, 0x7fc00000\n\t v_mov_b32 v31, 0x7fc00000\n\t"
        "s_nop 7\n\t"
        "v_pk_mul_f32 v[24:25], v[20:21], v[22:23]\n\t"                      // (a c, b d)
        "v_pk_fma_f32 v[26:27], v[24:25], v[22:23], v[20:21]\n\t"            // reads the product as src0, next instruction
        "v_pk_fma_f32 v[28:29], v[22:23], v[26:27], v[26:27]\n\t"            // reads that as src1 and src2, next instruction
        "v_add_f32 v30, v28, v29\n\t"                                        // scalar reader of a packed result, next instruction
        "v_pk_mul_f32 v[26:27], v[28:29], v[20:21] op_sel_hi:[0,1]\n\t"      // broadcast reader
        "v_cmp_le_f32 vcc, v26, v27\n\t"                                     // compare reader, next instruction
        "v_cndmask_b32 v31, v26, v27, vcc\n\t"
        "s_nop 7\n\t"
        "v_mov_b32 %0, v30\n\t v_mov_b32 %1, v31\n\t v_mov_b32 %2, v28\n\t v_mov_b32 %3, v29"
        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
        : "v"(a), "v"(b), "v"(c), "v"(d)
        : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "vcc");
    out_asm[4 * i] = r0; out_asm[4 * i + 1] = r1; out_asm[4 * i + 2] = r2; out_asm[4 * i + 3] = r3;
    // the same arithmetic, compiler-scheduled (with its wait states)
    const float m0 = a * c, m1 = b * d;
    const float f0 = __builtin_fmaf(m0, c, a), f1 = __builtin_fmaf(m1, d, b);
    const float g0 = __builtin_fmaf(c, f0, f0), g1 = __builtin_fmaf(d, f1, f1);
    const float s = g0 + g1;
    const float h0 = g0 * a, h1 = g0 * b;
    out_ref[4 * i] = s; out_ref[4 * i + 1] = h0 <= h1 ? h1 : h0; out_ref[4 * i + 2] = g0; out_ref[4 * i + 3] = g1;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> h(4 * (size_t)n);
    unsigned s = 12345u;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 20)); }
    float *in, *oa, *orf;
    (void)hipMalloc(&in, h.size() * 4); (void)hipMalloc(&oa, h.size() * 4); (void)hipMalloc(&orf, h.size() * 4);
    (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    size_t bad = 0;
    for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(k_test, dim3(n / 256), dim3(256), 0, 0, in, oa, orf, n);
        std::vector<float> a(h.size()), r(h.size());
        (void)hipMemcpy(a.data(), oa, h.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(r.data(), orf, h.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < a.size(); ++i) bad += std::memcmp(&a[i], &r[i], 4) != 0;
    }
    printf("packed-f32 result read by the next instruction, no wait state: %zu mismatching values of %zu\n", bad, (size_t)20 * h.size());
    return bad != 0;
}
