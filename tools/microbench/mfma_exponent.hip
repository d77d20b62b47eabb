// "prototype the MFMA-exponent forward at C = 4 and MEASURE it".
//
// The compositing kernels spend their time in one loop: per list entry and pixel the exponent
//     e' = a2 dx^2 + b2 dx dy + c2 dy^2 + log2(255 o)            (lsr_blend.h)
// then exp2 / min / two tests / the transmittance step / 4 + 2 accumulator updates.  The exponent is a rank-6 bilinear form
// between the monomials [1, u, v, u^2, uv, v^2] of a pixel's offset (u, v) from its 4x4 sub-block's centre and six
// coefficients per (entry, sub-block) — i.e. a 16-pixel x 16-element x K = 8 matrix product, two v_mfma_f32_16x16x4_f32.
// Because the monomial matrix is the same for every sub-block, the 16 rows of a product may be 4 consecutive entries of
// each of 4 DIFFERENT sub-blocks' lists: lane l then receives e' of four consecutive entries of ITS pixel (sub-block l >> 4,
// pixel l & 15) and walks them serially — no cross-lane scan for the transmittance.
//
// This file holds the INNER LOOPS of both formulations over synthetic, LDS-resident batches (64 staged entries per wave,
// every entry listed for every sub-block: no staging, no global memory in the loop — the part DESIGN.md's costing was about)
// with the product kernel's occupancy (24 waves per CU):
//   k_valu : the loop of render_forward.hip k_render_fwd<4, 12> (two pixels per lane, 8-lane groups, packed f32),
//            128 evaluations per iteration;
//   k_mfma : one pixel per lane, four sub-blocks per wave; per step two MFMAs (operands fetched from per-(entry, sub-block)
//            coefficient records) give 4 entries x 64 pixels = 256 evaluations, then the four serial blend steps with the
//            accumulators packed across channels.
// Both produce the same image up to rounding (checked), so the comparison is like for like; printed: shader cycles per
// 256 evaluations per SIMD and the ratio.  The MFMA variant's EXTRA costs outside the loop (six coefficients per (entry,
// sub-block) instead of one record per entry: ~2.6 x the staging writes; an exact replay at the decision boundary for
// bit-identical keep / stop decisions) are NOT in here: the measured ratio is an upper bound of what a full kernel gets.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_exponent mfma_exponent.hip ; run: ./mfma_exponent
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kWaves = 8, kEntries = 32;     // 3 workgroups of 8 waves per CU = the product kernel's 24 waves per CU; batches of 32 staged entries
                                             // (the MFMA variant's coefficient records — 32 bytes per (entry, SUB-BLOCK) instead of 48 per entry —
                                             // would not fit 24 waves per CU with 64)
constexpr float kInv255 = 1.0f / 255.0f, kAlphaMax255 = 0.99f * 255.0f, kTEps = 1e-4f;

struct Entry { float x, y, a2, c2, b2, l2o, z, pad; float pay[4]; };   // 48 bytes, as staged by k_render_fwd

__device__ __forceinline__ void pk_fma_lo(f2 &acc, f2 src, f2 ww) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(src), "v"(ww)); }
__device__ __forceinline__ void pk_fma_hi(f2 &acc, f2 src, f2 ww) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(src), "v"(ww)); }

// ---- today's loop: one wave = a 16x8 half tile, lane = two horizontally adjacent pixels, 8-lane group = one sub-block ----
__global__ void __launch_bounds__(64 * kWaves) k_valu(const Entry *ents, float *out, unsigned long long *cyc, int reps) {
    __shared__ float4 s_ent[kWaves][kEntries][3];
    __shared__ uint32_t s_list[kWaves][8][kEntries + 1];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Entry e = ents[lane & (kEntries - 1)];
    const char *ent_base = (const char *)&s_ent[0][0][0];
    if (lane < kEntries) {
        s_ent[wid][lane][0] = make_float4(e.x, e.y, e.a2, e.c2);
        s_ent[wid][lane][1] = make_float4(e.b2, e.l2o, e.z * kInv255, -kInv255);
        s_ent[wid][lane][2] = make_float4(e.pay[0] * kInv255, e.pay[1] * kInv255, e.pay[2] * kInv255, e.pay[3] * kInv255);
#ifdef DISTINCT   // every lane group at a DIFFERENT staged entry per iteration, as in the product kernel (8 distinct records per LDS read)
        for (int b = 0; b < 8; ++b) s_list[wid][b][lane] = (uint32_t)((wid * kEntries + ((lane * 5 + b * 7) & (kEntries - 1))) * 48);
#else
        for (int b = 0; b < 8; ++b) s_list[wid][b][lane] = (uint32_t)((wid * kEntries + lane) * 48);
#endif
    }
    __syncthreads();
    const int grp = lane >> 3, gcol = grp & 3, grow = grp >> 2, lx = 2 * (lane & 1), ly = (lane >> 1) & 3;
    const float px = 4.0f * gcol + lx, py = 4.0f * grow + ly;
    f2 acc[4], T2, D2;
    float kmax = kAlphaMax255;
    asm volatile("" : "+v"(kmax));
    float total = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        const f2 pxx = f2{px, px + 1.0f};
        T2 = f2{1.0f, 1.0f}; D2 = f2{0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f2{0.0f, 0.0f};
        const uint32_t *lp = &s_list[wid][grp][0];
#pragma unroll 1
        for (uint32_t i = 0; i < (uint32_t)kEntries; ++i) {
            const uint32_t off = lp[i];
            const float4 *E = (const float4 *)(ent_base + off);
            const float4 a = E[0], b = E[1], t4 = E[2];
            const f2 pay0 = f2{t4.x, t4.y}, pay1 = f2{t4.z, t4.w}, zk = f2{b.z, b.w};
            const f2 d2 = f2{a.x, a.x} - pxx;
            const float dy = a.y - py;
            const float t = b.x * dy;
            const float s = __builtin_fmaf(a.w * dy, dy, b.y);
            const f2 p1 = __builtin_elementwise_fma(f2{a.z, a.z}, d2, f2{t, t});
            const f2 ex = __builtin_elementwise_fma(p1, d2, f2{s, s});
            const f2 al = f2{fminf(kmax, __builtin_amdgcn_exp2f(ex.x)), fminf(kmax, __builtin_amdgcn_exp2f(ex.y))};
            const uint32_t lim = __float_as_uint(b.y);
            const uint64_t ok0 = __ballot(__float_as_uint(ex.x) <= lim), ok1 = __ballot(__float_as_uint(ex.y) <= lim);
            f2 aT, tT;
            asm("v_pk_mul_f32 %0, %2, %3\n\tv_pk_fma_f32 %1, %0, %4, %3 op_sel:[0,1,0]" : "=&v"(aT), "=v"(tT) : "v"(al), "v"(T2), "v"(zk));
            const uint64_t room0 = __ballot(tT.x >= kTEps), room1 = __ballot(tT.y >= kTEps);
            const float w0 = __builtin_amdgcn_inverse_ballot_w64(ok0 & room0) ? aT.x : 0.0f;
            const float w1 = __builtin_amdgcn_inverse_ballot_w64(ok1 & room1) ? aT.y : 0.0f;
            const f2 ww = f2{w0, w1};
            pk_fma_lo(acc[0], pay0, ww); pk_fma_hi(acc[1], pay0, ww); pk_fma_lo(acc[2], pay1, ww); pk_fma_hi(acc[3], pay1, ww);
            pk_fma_lo(D2, zk, ww); pk_fma_hi(T2, zk, ww);
            if ((ok0 & ~room0) | (ok1 & ~room1)) { total += 1.0f; }   // (the stop bookkeeping of the real kernel: rare)
        }
        total += acc[0].x + acc[0].y + acc[1].x + acc[1].y + acc[2].x + acc[2].y + acc[3].x + acc[3].y + D2.x + D2.y + T2.x + T2.y;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = total;
    if (lane == 0) cyc[blockIdx.x * kWaves + wid] = t1 - t0;
}

// ---- the MFMA formulation: lane = ONE pixel, wave = four sub-blocks (a 16x4 pixel row of the half tile) ----
// coefficient record of (entry, sub-block), 8 floats in MFMA operand order: (c0 c4) (c1 c5) (c2 0) (c3 0): lane quarter k
// reads the pair (k, k + 4) with one 8-byte LDS read.  Monomials: m0 = 1, m1 = u, m2 = v, m3 = u^2 | m4 = uv, m5 = v^2.
__global__ void __launch_bounds__(64 * kWaves) k_mfma(const Entry *ents, float *out, unsigned long long *cyc, int reps) {
    __shared__ float2 s_coef[kWaves][4][kEntries][4];        // [sub-block][entry][k]: (c_k, c_{k+4})
    __shared__ float4 s_pay[kWaves][kEntries];               // payload / 255
    __shared__ float2 s_lz[kWaves][kEntries];                // (log2(255 o), z / 255)
    __shared__ uint32_t s_list[kWaves][4][kEntries + 4];     // entry index per list position
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Entry e = ents[lane & (kEntries - 1)];
    if (lane < kEntries) {
    for (int g = 0; g < 4; ++g) {       // expansion of e' around the centre of sub-block g: pixel = centre + (u, v), d = mean - pixel
        const float cx = 4.0f * g + 1.5f, cy = 1.5f;
        const float dx0 = e.x - cx, dy0 = e.y - cy;
        const float c0 = e.a2 * dx0 * dx0 + e.b2 * dx0 * dy0 + e.c2 * dy0 * dy0 + e.l2o;
        const float c1 = -2.0f * e.a2 * dx0 - e.b2 * dy0, c2 = -e.b2 * dx0 - 2.0f * e.c2 * dy0;
        s_coef[wid][g][lane][0] = make_float2(c0, e.b2);      // m0 = 1,  m4 = uv
        s_coef[wid][g][lane][1] = make_float2(c1, e.c2);      // m1 = u,  m5 = v^2
        s_coef[wid][g][lane][2] = make_float2(c2, 0.0f);      // m2 = v
        s_coef[wid][g][lane][3] = make_float2(e.a2, 0.0f);    // m3 = u^2
#ifdef DISTINCT
        s_list[wid][g][lane] = (lane * 5 + g * 7) & (kEntries - 1);
#else
        s_list[wid][g][lane] = lane;
#endif
    }
    s_pay[wid][lane] = make_float4(e.pay[0] * kInv255, e.pay[1] * kInv255, e.pay[2] * kInv255, e.pay[3] * kInv255);
    s_lz[wid][lane] = make_float2(e.l2o, e.z * kInv255);
    }
    __syncthreads();
    const int g = lane >> 4, n = lane & 15;                   // my sub-block, my pixel of it (also: my K quarter, my operand row)
    const float u = (float)(n & 3) - 1.5f, v = (float)(n >> 2) - 1.5f;
    const int k = lane >> 4;
    const float b0 = k == 0 ? 1.0f : (k == 1 ? u : (k == 2 ? v : u * u));      // B[k][n]: monomials 1, u, v, u^2
    const float b1 = k == 0 ? u * v : (k == 1 ? v * v : 0.0f);                 //          uv, v^2, 0, 0
    const int rg = (lane & 15) >> 2, rslot = lane & 3;        // operand row m = n: entry `rslot` of sub-block `rg`'s current four
    f2 acc01, acc23, DT;
    float kmax = kAlphaMax255;
    asm volatile("" : "+v"(kmax));
    f2 kz = f2{0.0f, -kInv255};
    float total = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        acc01 = acc23 = f2{0.0f, 0.0f};
        DT = f2{0.0f, 1.0f};                                  // (sum alpha T z, T)
#pragma unroll 1
        for (uint32_t i = 0; i < (uint32_t)kEntries; i += 4) {
            // A operands: the coefficient pairs of row (rg, rslot)
            const uint32_t ea = s_list[wid][rg][i + rslot];
            const float2 ca = s_coef[wid][rg][ea][k];
            f4 ex = f4{0.0f, 0.0f, 0.0f, 0.0f};
            ex = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.x, b0, ex, 0, 0, 0);
            ex = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.y, b1, ex, 0, 0, 0);
            // my four entries
            const uint4 mine = *(const uint4 *)&s_list[wid][g][i];
            const uint32_t idx[4] = {mine.x, mine.y, mine.z, mine.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 lz = s_lz[wid][idx[j]];
                const float4 pay = s_pay[wid][idx[j]];
                const float al = fminf(kmax, __builtin_amdgcn_exp2f(ex[j]));
                const uint64_t ok = __ballot(__float_as_uint(ex[j]) <= __float_as_uint(lz.x));
                const float aT = al * DT.y;
                const float tT = __builtin_fmaf(aT, -kInv255, DT.y);
                const uint64_t room = __ballot(tT >= kTEps);
                const float w = __builtin_amdgcn_inverse_ballot_w64(ok & room) ? aT : 0.0f;
                const f2 ww = f2{w, w};
                acc01 = __builtin_elementwise_fma(f2{pay.x, pay.y}, ww, acc01);
                acc23 = __builtin_elementwise_fma(f2{pay.z, pay.w}, ww, acc23);
                kz.x = lz.y;
                DT = __builtin_elementwise_fma(kz, ww, DT);
                if (ok & ~room) { total += 1.0f; }
            }
        }
        total += acc01.x + acc01.y + acc23.x + acc23.y + DT.x + DT.y;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = total;
    if (lane == 0) cyc[blockIdx.x * kWaves + wid] = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 3, reps = 800;
    std::vector<Entry> h(kEntries);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (auto &e : h) {     // splats of 1 - 5 px around a 16 x 8 half tile: about 40 % of the evaluations pass the alpha test
        const float sx = 1.0f + 4.0f * rnd(), sy = sx * (0.4f + 0.6f * rnd()), rho = 0.8f * (rnd() - 0.5f);
        const float A = 1.0f / (sx * sx * (1 - rho * rho)), C = 1.0f / (sy * sy * (1 - rho * rho)), B = -rho / (sx * sy * (1 - rho * rho));
        const float o = 0.02f + 0.3f * rnd();
        e.x = -2.0f + 20.0f * rnd(); e.y = -2.0f + 12.0f * rnd();
        e.a2 = -0.5f * 1.4426950408889634f * A; e.b2 = -1.4426950408889634f * B; e.c2 = -0.5f * 1.4426950408889634f * C;
        e.l2o = log2f(255.0f * o); e.z = 1.0f + 9.0f * rnd(); e.pad = 0.0f;
        for (float &p : e.pay) p = rnd();
    }
    Entry *d_e; float *d_out; unsigned long long *d_cyc;
    (void)hipMalloc((void **)&d_e, sizeof(Entry) * kEntries);
    (void)hipMalloc((void **)&d_out, sizeof(float) * blocks * 64 * kWaves);
    (void)hipMalloc((void **)&d_cyc, 8 * blocks * kWaves);
    (void)hipMemcpy(d_e, h.data(), sizeof(Entry) * kEntries, hipMemcpyHostToDevice);
    std::vector<float> o1(blocks * 64 * kWaves), o2(o1.size());
    std::vector<unsigned long long> c(blocks * kWaves);
    double per256[2] = {0, 0};
    for (int variant = 0; variant < 2; ++variant) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 0.0f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (variant == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(64 * kWaves), 0, 0, d_e, d_out, d_cyc, reps);
            else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(64 * kWaves), 0, 0, d_e, d_out, d_cyc, reps);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("  kernel %.3f ms (hipEvents)\n", ms);
        (void)hipMemcpy(variant ? o2.data() : o1.data(), d_out, sizeof(float) * o1.size(), hipMemcpyDeviceToHost);
        (void)hipMemcpy(c.data(), d_cyc, 8 * c.size(), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto x : c) mean += (double)x;
        mean /= c.size();
        // a wave performs (variant 0: kEntries iterations x 128, variant 1: kEntries / 4 steps x 256) evaluations per rep; 6 waves share a SIMD
        const double evals = (variant == 0 ? kEntries * 128.0 : (kEntries / 4) * 256.0) * reps;
        // (a wave's lifetime against the kernel's: all 24 waves of a CU resident together <=> the two agree)
        printf("  mean wave lifetime %.3f ms at 2.42 GHz\n", mean / 2.42e6);
        per256[variant] = mean / evals * 256.0 / 6.0;
        printf("%s: %.0f cycles per wave for %d batches -> %.1f shader cycles per 256 evaluations per SIMD (6 waves per SIMD)\n",
               variant ? "k_mfma (1 pixel / lane, 2 MFMA per 256 evaluations)" : "k_valu (today's loop)", mean, reps, per256[variant]);
    }
    // same image? wave 0 of block 0: the valu variant's lane (group, l8) holds pixels (4 gcol + lx + {0, 1}, 4 grow + ly); compare the
    // totals over the top 16 x 4 row, which both variants cover (mfma: wave = sub-blocks 0..3 = rows 0..3)
    double a = 0, b = 0;
    for (int l = 0; l < 64; ++l) { if ((l >> 3) < 4) a += o1[l]; b += o2[l]; }
    printf("sum over the 16 x 4 pixel row: valu %.6f  mfma %.6f  (relative difference %.2e)\n", a, b, fabs(a - b) / fabs(a));
    printf("ratio valu / mfma per evaluation: %.3f\n", per256[0] / per256[1]);
    return 0;
}
