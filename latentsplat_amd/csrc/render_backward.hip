// render_backward.hip — stage K7: per-pixel gradient of the compositing, with the same sub-block
// decomposition, culling and work items as the forward (render_forward.hip, lsr_blend.h).
//
// The lists are walked FRONT TO BACK like the forward (the published kernel goes back to front and
// keeps, per pixel and channel, the colour accumulated behind the current entry: four operations
// per channel per evaluation).  Here the only per-pixel state is the transmittance T and ONE scalar
//   R_i = sum_{j>i} w_j (g . c_j)  [+ T_final (g . bg - g_mask)]  =  (g . C_rendered) - prefix_i,
// seeded from the rendered images of the matching forward; with d_i = g . c_i
//   dL/dalpha_i = T_i d_i - R_i / (1 - alpha_i),      R_i = R_{i-1} - w_i d_i,   T_{i+1} = T_i (1 - alpha_i)
// i.e. one dot product and one multiply (payload gradient) per channel.  Cancellation in R only
// matters when the remaining contribution is already ~1e-7 of the pixel's total — far below the
// 1e-4 tolerance.
//
// Per (entry, pixel) the kernel recomputes alpha with the forward's exact arithmetic and emits the
// MOMENTS of u = exp2(e) * dL/dalpha (= opacity * G * dL/dalpha) over the entry's pixels:
//   m0 = sum u,   m1 = sum u (dx, dy),   m2 = sum u (dx^2, dx dy, dy^2),
// plus w * g per payload channel and w * g_depth.  The conic / opacity factors that turn moments
// into dL/d(x, y), dL/d(A, B, C), dL/d opacity are applied ONCE per (view, Gaussian) by
// k_preprocess_bwd (they are constant over the pixels), not per evaluation.
//
// Execution shape: the forward's work items (one 8x8 quadrant of a tile per wave, walking that quadrant's
// render list); one lane = one pixel, one 16-lane group = one 4x4 sub-block walking its own
// compacted list (as in the forward), so a wave instruction serves up to four different entries.
// The contributions of a group's 16 pixels are summed by a transposed butterfly INSIDE the DPP row
// (row_ror / row_half_mirror / quad_perm — full-rate, no permlane, no LDS): afterwards lane s of
// the group holds gradient slot s, and ONE ds_add_f32 instruction adds the four groups' records
// into a per-batch LDS table (row = staged entry).  When the batch is done the rows are flushed with
// coalesced global atomics, one 64-byte record per 16 lanes: one global record-add per entry of a
// quadrant list (1.2 per (Gaussian, tile) pair) instead of one per (entry, wave) reduction of 64
// lanes + atomic in round 1 of this kernel (that reduction was ~40 % of its time).
// Spec: SURVEY.md Appendix A.6.
#include "lsr_blend.h"

namespace lsr {

__device__ __forceinline__ void wave_lds_fence_bwd() {
    // compiler-only barrier: the LDS slice is private to the wave and its LDS operations execute in
    // order (a memory fence would also drain the prefetching global loads and the flush atomics)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

typedef float float2_b __attribute__((ext_vector_type(2)));
typedef float float4_b __attribute__((ext_vector_type(4)));

struct RenderBwdParams {
    int H, W, gx, T, G, C, has_color;
    int num_cus;                  // workgroups of 4*WPS waves (one per compute unit)
    const uint32_t *items;        // work items of the forward (view*T + tile | quadrant << 28), costliest first
    const uint32_t *header;       // geometry-workspace header (item count)
    uint32_t *queue;              // work-queue head (zeroed with the gradient workspace)
    const float *views;
    const float4 *geo;            // [V*G][rec_f4] screen-space records (lsr_internal.h)
    int rec_f4;
    const uint32_t *tile_start, *quad_list;
    const float *final_T;
    const uint32_t *n_contrib;
    const float *g_color, *g_feat, *g_mask, *g_depth;  // dL/d outputs (any may be NULL)
    const float *f_color, *f_feat, *f_depth;           // images rendered by the matching forward
    float *rec;         // [V*G][rec_floats] packed gradient records (zeroed by the caller)
    long long *rec_fixed;   // deterministic mode: the same records as 2^30 fixed point (zeroed by the caller), else NULL
    int rec_floats;
};

// ---- transposed reduction inside a DPP row ------------------------------------------------
// Sums 16 per-lane values over the 16 lanes of a row; every butterfly step halves the number of
// values a lane still carries and lane l ends up with the row total of value l.
//   step A  row_ror:8          lane bit 3 selects value i / i+8
//   step B  row_half_mirror    lane bit 2 selects i / i+4   (partner 7-l: same bit 3, other bit 2)
//   step C  quad_perm [2,3,0,1] lane bit 1 selects i / i+2
//   step D  quad_perm [1,0,3,2] lane bit 0 selects i / i+1
// LIVE = compile-time mask of values that can be non-zero: a pair with one live member costs one
// DPP add, a dead pair nothing (lanes that end up with a dead value hold junk and do not write).
// (Values are passed and kept as individual scalars on purpose: as an array the optimiser turns the
// lane-bit selects into dynamic element extraction — a chain of compares and selects per value.)
template <int CTRL>
__device__ __forceinline__ float row_add(float x) {   // x + partner(x)
    return x + f_from_u((unsigned)__builtin_amdgcn_update_dpp(0, (int)u_from_f(x), CTRL, 0xf, 0xf, false));
}
// one butterfly output: lanes with `bit` keep the hi value, the others the lo value; each adds
// what its partner sends for the kept value
template <int CTRL, bool LO, bool HI>
__device__ __forceinline__ float row_step(bool bit, float lo, float hi) {
    if (LO && HI) {
        const float keep = bit ? hi : lo, send = bit ? lo : hi;
        return keep + f_from_u((unsigned)__builtin_amdgcn_update_dpp(0, (int)u_from_f(send), CTRL, 0xf, 0xf, false));
    }
    if (LO) return row_add<CTRL>(lo);
    if (HI) return row_add<CTRL>(hi);
    return 0.0f;
}
template <uint32_t LIVE>
__device__ __forceinline__ float row_reduce16_transposed(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                                         float v8, float v9, float v10, float v11, float v12, float v13, float v14, float v15,
                                                         int l16) {
    const bool b3 = l16 & 8, b2 = l16 & 4, b1 = l16 & 2, b0 = l16 & 1;
#define LV(i) ((LIVE >> (i)) & 1u)
    const float r0 = row_step<0x128, LV(0), LV(8)>(b3, v0, v8), r1 = row_step<0x128, LV(1), LV(9)>(b3, v1, v9);
    const float r2 = row_step<0x128, LV(2), LV(10)>(b3, v2, v10), r3 = row_step<0x128, LV(3), LV(11)>(b3, v3, v11);
    const float r4 = row_step<0x128, LV(4), LV(12)>(b3, v4, v12), r5 = row_step<0x128, LV(5), LV(13)>(b3, v5, v13);
    const float r6 = row_step<0x128, LV(6), LV(14)>(b3, v6, v14), r7 = row_step<0x128, LV(7), LV(15)>(b3, v7, v15);
    constexpr uint32_t LA = (LIVE | (LIVE >> 8)) & 0xFFu;
#define LVA(i) ((LA >> (i)) & 1u)
    const float s0 = row_step<0x141, LVA(0), LVA(4)>(b2, r0, r4), s1 = row_step<0x141, LVA(1), LVA(5)>(b2, r1, r5);
    const float s2 = row_step<0x141, LVA(2), LVA(6)>(b2, r2, r6), s3 = row_step<0x141, LVA(3), LVA(7)>(b2, r3, r7);
    constexpr uint32_t LB = (LA | (LA >> 4)) & 0xFu;
#define LVB(i) ((LB >> (i)) & 1u)
    const float t0 = row_step<0x4E, LVB(0), LVB(2)>(b1, s0, s2), t1 = row_step<0x4E, LVB(1), LVB(3)>(b1, s1, s3);
    constexpr uint32_t LC = (LB | (LB >> 2)) & 0x3u;
    return row_step<0xB1, (LC & 1u) != 0, (LC & 2u) != 0>(b0, t0, t1);
#undef LV
#undef LVA
#undef LVB
}

// Hand-scheduled versions of the same reduction for the two hot payload widths (4 and 8 channels, no
// depth gradient).  Steps A and B select "which half a lane keeps" with the DPP BANK mask (a bank =
// 4 lanes, exactly lane bits 2-3): the second add of a pair simply overwrites the lanes of the
// selected banks, so a full pair costs two DPP adds instead of two selects + one DPP add.  Steps C
// and D select inside a quad, which no DPP mask can express: v_cndmask with constant lane masks.
// 25 VALU for 10 values (36 as compiled from row_reduce16_transposed), 31 for 14.
// DPP reads need two wait states after a VALU write of the same register and inline asm is opaque
// to the hazard recogniser: the instruction order below keeps >= 2 instructions between every
// write and its DPP read, the two places where that is impossible carry an s_nop.
#define LSR_DPP_A "row_ror:8 row_mask:0xf "
#define LSR_DPP_B "row_half_mirror row_mask:0xf "
#define LSR_DPP_C "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
#define LSR_DPP_D "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
__device__ __forceinline__ float row_reduce_4ch(float a0, float a1, float a2, float a3, float a4, float a5,
                                                float p0, float p1, float p2, float p3) {
    // lane l ends with: slots 0..5 = a0..a5, slots 8..11 = p0..p3 (other lanes: junk)
    float r0, r1, r2, r3, r4, r5, s0, s1, s2, s3, k0, n0, k1, n1, t0, t1, k, n, out;
    const uint64_t m1 = 0xCCCCCCCCCCCCCCCCull, m0 = 0xAAAAAAAAAAAAAAAAull;   // lanes with bit 1 / bit 0 set
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %[r0], %[a0], %[a0] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r1], %[a1], %[a1] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r2], %[a2], %[a2] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r3], %[a3], %[a3] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r4], %[a4], %[a4] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r5], %[a5], %[a5] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r0], %[p0], %[p0] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r1], %[p1], %[p1] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r2], %[p2], %[p2] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r3], %[p3], %[p3] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[s0], %[r0], %[r0] " LSR_DPP_B "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s1], %[r1], %[r1] " LSR_DPP_B "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s2], %[r2], %[r2] " LSR_DPP_B "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s3], %[r3], %[r3] " LSR_DPP_B "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s0], %[r4], %[r4] " LSR_DPP_B "bank_mask:0xa\n\t"
        "v_add_f32_dpp %[s1], %[r5], %[r5] " LSR_DPP_B "bank_mask:0xa\n\t"
        "v_cndmask_b32_e64 %[k0], %[s0], %[s2], %[m1]\n\t"
        "v_cndmask_b32_e64 %[n0], %[s2], %[s0], %[m1]\n\t"
        "v_cndmask_b32_e64 %[n1], %[s3], %[s1], %[m1]\n\t"
        "v_cndmask_b32_e64 %[k1], %[s1], %[s3], %[m1]\n\t"
        "v_add_f32_dpp %[t0], %[n0], %[k0] " LSR_DPP_C "\n\t"
        "v_add_f32_dpp %[t1], %[n1], %[k1] " LSR_DPP_C "\n\t"
        "v_cndmask_b32_e64 %[k], %[t0], %[t1], %[m0]\n\t"
        "v_cndmask_b32_e64 %[n], %[t1], %[t0], %[m0]\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[out], %[n], %[k] " LSR_DPP_D
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5),
          [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [s3] "=&v"(s3), [k0] "=&v"(k0), [n0] "=&v"(n0),
          [k1] "=&v"(k1), [n1] "=&v"(n1), [t0] "=&v"(t0), [t1] "=&v"(t1), [k] "=&v"(k), [n] "=&v"(n), [out] "=&v"(out)
        : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [a4] "v"(a4), [a5] "v"(a5),
          [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [m1] "s"(m1), [m0] "s"(m0));
    return out;
}
__device__ __forceinline__ float row_reduce_8ch(float a0, float a1, float a2, float a3, float a4, float a5,
                                                float p0, float p1, float p2, float p3, float p4, float p5, float p6, float p7) {
    // slots 0..5 = a0..a5, slots 8..15 = p0..p7
    float r0, r1, r2, r3, r4, r5, r6, r7, s0, s1, s2, s3, k0, n0, k1, n1, t0, t1, k, n, out;
    const uint64_t m1 = 0xCCCCCCCCCCCCCCCCull, m0 = 0xAAAAAAAAAAAAAAAAull;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %[r0], %[a0], %[a0] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r1], %[a1], %[a1] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r2], %[a2], %[a2] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r3], %[a3], %[a3] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r4], %[a4], %[a4] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r5], %[a5], %[a5] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r6], %[p6], %[p6] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r7], %[p7], %[p7] " LSR_DPP_A "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[r0], %[p0], %[p0] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r1], %[p1], %[p1] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r2], %[p2], %[p2] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r3], %[p3], %[p3] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r4], %[p4], %[p4] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[r5], %[p5], %[p5] " LSR_DPP_A "bank_mask:0xc\n\t"
        "v_add_f32_dpp %[s0], %[r0], %[r0] " LSR_DPP_B "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s1], %[r1], %[r1] " LSR_DPP_B "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s2], %[r2], %[r2] " LSR_DPP_B "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s3], %[r3], %[r3] " LSR_DPP_B "bank_mask:0xf\n\t"
        "v_add_f32_dpp %[s0], %[r4], %[r4] " LSR_DPP_B "bank_mask:0xa\n\t"
        "v_add_f32_dpp %[s1], %[r5], %[r5] " LSR_DPP_B "bank_mask:0xa\n\t"
        "v_add_f32_dpp %[s2], %[r6], %[r6] " LSR_DPP_B "bank_mask:0xa\n\t"
        "v_add_f32_dpp %[s3], %[r7], %[r7] " LSR_DPP_B "bank_mask:0xa\n\t"
        "v_cndmask_b32_e64 %[k0], %[s0], %[s2], %[m1]\n\t"
        "v_cndmask_b32_e64 %[n0], %[s2], %[s0], %[m1]\n\t"
        "v_cndmask_b32_e64 %[n1], %[s3], %[s1], %[m1]\n\t"
        "v_cndmask_b32_e64 %[k1], %[s1], %[s3], %[m1]\n\t"
        "v_add_f32_dpp %[t0], %[n0], %[k0] " LSR_DPP_C "\n\t"
        "v_add_f32_dpp %[t1], %[n1], %[k1] " LSR_DPP_C "\n\t"
        "v_cndmask_b32_e64 %[k], %[t0], %[t1], %[m0]\n\t"
        "v_cndmask_b32_e64 %[n], %[t1], %[t0], %[m0]\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[out], %[n], %[k] " LSR_DPP_D
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5), [r6] "=&v"(r6), [r7] "=&v"(r7),
          [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [s3] "=&v"(s3), [k0] "=&v"(k0), [n0] "=&v"(n0),
          [k1] "=&v"(k1), [n1] "=&v"(n1), [t0] "=&v"(t0), [t1] "=&v"(t1), [k] "=&v"(k), [n] "=&v"(n), [out] "=&v"(out)
        : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [a4] "v"(a4), [a5] "v"(a5),
          [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4), [p5] "v"(p5), [p6] "v"(p6), [p7] "v"(p7),
          [m1] "s"(m1), [m0] "s"(m0));
    return out;
}
#undef LSR_DPP_A
#undef LSR_DPP_B
#undef LSR_DPP_C
#undef LSR_DPP_D

__device__ __forceinline__ void atomic_add_f32(float *addr, float v) { unsafeAtomicAdd(addr, v); }

// WPS = resident waves per SIMD (workgroup = 4*WPS waves = one compute unit's worth): 4 where the
// per-wave LDS slice allows it, fewer for the wide payloads.
template <int NCHP, bool DEPTH_GRAD, int WPS>
__global__ void __launch_bounds__(LSR_WAVE * 4 * WPS)
k_render_bwd(RenderBwdParams p) {
    constexpr int WPB = 4 * WPS;
    // float4 per staged entry: odd (conflict-free staging stores) except for the 8-channel payload, whose
    // 16 floats are stored at a 64-byte stride (2-way conflicts in the staging stores only) so that the
    // slice fits 16 waves per CU — with 12 (3 per SIMD) this VALU-bound kernel ran 32 % slower (measured
    // on the 4-channel variant: 0.92 -> 1.21 ms)
    constexpr int kEnt = NCHP == 8 ? 4 : ((2 + NCHP / 4) | 1);
    constexpr int RF = NCHP <= 8 ? 16 : (NCHP <= 12 ? 32 : 64);       // == rec_floats (lsr_internal.h)
    constexpr int NGRP = RF / 16;                                     // 16-value reduction passes per evaluation
    // words per row of the LDS gradient table: the record without its two never-written slots (6 unless the
    // depth gradient is on, 7) for the 8-channel payload (same LDS budget), the full record otherwise
    constexpr bool kPackRow = NCHP == 8 && !DEPTH_GRAD;
    constexpr int RT = kPackRow ? 14 : RF;
    constexpr int kListRow = LSR_WAVE + 2;    // u16 per list row: 33 words, so the four lane groups' reads of list[b][i] hit four banks
    struct Lds {
        float4 ent[WPB][LSR_WAVE + 1][kEnt];   // (x, y, a2, c2) (b2, log2 o, z, list position) payload...; slot 64 = null record
        float acc[WPB][LSR_WAVE + 1][RT];      // this batch's gradient records, row = staged entry (row 64: dump row of the null record)
        uint32_t gid[WPB][LSR_WAVE];           // Gaussian index of the staged entry
        uint16_t list[WPB][4][kListRow];       // per sub-block: staging slots of the entries that can reach it, in list order
    };
    __shared__ Lds s_lds;

    const int lane = threadIdx.x & (LSR_WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / LSR_WAVE);
    float4 (*s_ent)[kEnt] = s_lds.ent[wid];
    float (*s_acc)[RT] = s_lds.acc[wid];
    uint32_t *s_gid = s_lds.gid[wid];
    uint16_t (*s_list)[kListRow] = s_lds.list[wid];
    {   // null record + cleared gradient table (rows are re-zeroed by the flush)
        if (lane == 0) {
            s_ent[LSR_WAVE][0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            s_ent[LSR_WAVE][1] = make_float4(0.0f, -INFINITY, 0.0f, __uint_as_float(0xFFFFFFFFu));  // log2(opacity) = -inf, position beyond every list
#pragma unroll
            for (int c4 = 0; c4 < NCHP / 4; ++c4) s_ent[LSR_WAVE][2 + c4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (lane < 8) s_list[lane >> 1][LSR_WAVE + (lane & 1)] = (uint16_t)LSR_WAVE;   // the rows' pad words
        float *A = &s_acc[0][0];
        for (int i = lane; i < (LSR_WAVE + 1) * RT; i += LSR_WAVE) A[i] = 0.0f;
    }
    const uint32_t num_items = p.header[kHdrNumItems];
    const int coff = p.has_color ? 3 : 0;
    const size_t HW = (size_t)p.H * p.W;
    const int grp = lane >> 4, gcol = grp & 1, grow = grp >> 1, l16 = lane & 15;
    const int tcol = kPackRow ? (l16 < 8 ? l16 : l16 - 2) : l16;   // this lane's record slot -> column of the table row (slots 6, 7 unused when packed)
    const int lx = lane & 3, ly = (lane >> 2) & 3;

    const uint32_t simd_bins = (uint32_t)p.num_cus * 4u, slots = simd_bins * (uint32_t)WPS;
    const uint32_t vwave = (uint32_t)wid;                     // 0..WPB-1; waves w, w+4, ... share a SIMD
    const uint32_t bin = blockIdx.x * 4u + (vwave & 3u);
    const uint32_t j0 = vwave >> 2;
    bool first = true;
    for (;;) {
        uint32_t qi;
        if (first) {
            qi = (j0 & 1u) ? (j0 + 1u) * simd_bins - 1u - bin : j0 * simd_bins + bin;
            first = false;
            if (qi >= num_items) continue;
        } else {
            if (num_items <= slots) break;
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(p.queue, 1u);
            qi = slots + __builtin_amdgcn_readfirstlane(t);
            if (qi >= num_items) break;
        }
        qi = __builtin_amdgcn_readfirstlane(qi);
        const uint32_t item = p.items[qi];
        const uint32_t vt = item & kItemTileMask, quad = item >> kItemQuadShift;
        const int tile = (int)(vt % (uint32_t)p.T), v = (int)(vt / (uint32_t)p.T);
        const int tx0 = (tile % p.gx) * LSR_TILE + 8 * (int)(quad & 1u), ty0 = (tile / p.gx) * LSR_TILE + 8 * (int)(quad >> 1);
        const size_t vG = (size_t)v * p.G;
        const uint32_t tstart = p.tile_start[vt], tn = p.tile_start[vt + 1] - tstart;
        const uint32_t *qlist = p.quad_list + 4 * (size_t)tstart + (size_t)quad * tn;

        float2_b pxy;
        float Tr, Rr, ddep;
        float2_b dpix[NCHP / 2];
        uint32_t last;
        uint32_t maxlast;
        {
            const int px = tx0 + 4 * gcol + lx, py = ty0 + 4 * grow + ly;
            pxy = float2_b{(float)px, (float)py};
            const bool inside = px < p.W && py < p.H;
            // every per-pixel load is issued UNCONDITIONALLY at a clamped pixel (all of a row's loads in
            // flight together) and masked afterwards: loads under the per-lane `inside` test were waited
            // for one by one — ~30 serial round trips at the head of every item
            const int pxc = min(px, p.W - 1), pyc = min(py, p.H - 1);
            const size_t pix = (size_t)pyc * p.W + pxc, vp = (size_t)v * HW + pix;
            const float Tfin_l = p.final_T[vp];
            const uint32_t nc_l = p.n_contrib[vp];
            float gl[NCHP], fl[NCHP];
#pragma unroll
            for (int c = 0; c < NCHP; ++c) gl[c] = fl[c] = 0.0f;
            if (p.has_color && p.g_color) {          // kernel-argument uniform
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    gl[c] = p.g_color[((size_t)v * 3 + c) * HW + pix];
                    fl[c] = p.f_color[((size_t)v * 3 + c) * HW + pix];
                }
            }
            if (p.g_feat) {
#pragma unroll
                for (int c = 0; c < NCHP; ++c)
                    if (c >= coff && c - coff < p.C) {   // uniform
                        gl[c] = p.g_feat[((size_t)v * p.C + (c - coff)) * HW + pix];
                        fl[c] = p.f_feat[((size_t)v * p.C + (c - coff)) * HW + pix];
                    }
            }
            const float gmask_l = p.g_mask ? p.g_mask[vp] : 0.0f;
            const float gdep_l = DEPTH_GRAD ? p.g_depth[vp] : 0.0f, fdep_l = DEPTH_GRAD ? p.f_depth[vp] : 0.0f;
            const float Tfin = inside ? Tfin_l : 1.0f;
            Tr = 1.0f;
            last = inside ? nc_l : 0u;
            maxlast = last;
            // R_0 = g . (rendered - T_final * bg)  +  T_final * (g . bg - g_mask)  =  g . rendered - T_final * g_mask
            float r0 = 0.0f;
            float dp[NCHP];
#pragma unroll
            for (int c = 0; c < NCHP; ++c) {
                dp[c] = inside ? gl[c] : 0.0f;
                r0 = __builtin_fmaf(fl[c], dp[c], r0);     // channels that are not rendered hold zeros
            }
            r0 = __builtin_fmaf(-Tfin, inside ? gmask_l : 0.0f, r0);  // mask = 1 - T_final
#pragma unroll
            for (int c = 0; c < NCHP / 2; ++c) dpix[c] = float2_b{dp[2 * c], dp[2 * c + 1]};
            ddep = (DEPTH_GRAD && inside) ? gdep_l : 0.0f;
            if (DEPTH_GRAD) r0 = __builtin_fmaf(fdep_l, ddep, r0);
            Rr = r0;
        }
        // wave-uniform upper bound of the list entries any pixel has to consider
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxlast = max(maxlast, (uint32_t)__shfl_xor((int)maxlast, off));
        maxlast = __builtin_amdgcn_readfirstlane(maxlast);

        // software-pipelined staging as in the forward: records of batch b+1 and list entries of
        // batch b+2 are in flight while batch b is processed
        // (unconditional loads at clamped positions, see the forward: a load under a per-lane condition is
        // waited for at the join)
        struct StageRec { float4 a, b, pay[NCHP / 4]; uint32_t w; };
        const uint32_t lastrel = maxlast - 1u;   // only used when maxlast > 0
        auto load_ent = [&](uint32_t rel) -> uint32_t { return qlist[min(rel, lastrel)]; };
        auto load_rec = [&](uint32_t w) {
            StageRec r;
            r.w = w;
            const float4 *R = p.geo + (vG + (w & kQuadIndexMask)) * (size_t)p.rec_f4;
            r.a = R[0]; r.b = R[1];
#pragma unroll
            for (int c4 = 0; c4 < NCHP / 4; ++c4) r.pay[c4] = R[2 + c4];
            return r;
        };
        uint32_t w_ahead = 0;
        StageRec nxt;
        nxt.w = 0;
        nxt.a = nxt.b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int c4 = 0; c4 < NCHP / 4; ++c4) nxt.pay[c4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (maxlast > 0) {   // wave-uniform
            w_ahead = load_ent(lane);
            nxt = load_rec(w_ahead);
            w_ahead = load_ent(LSR_WAVE + lane);
        }

        for (uint32_t cbase = 0; cbase < maxlast; cbase += LSR_WAVE) {
            const StageRec cur = nxt;
            nxt = load_rec(w_ahead);
            w_ahead = load_ent(cbase + 2 * LSR_WAVE + lane);
            // ---- stage up to 64 list entries (one per lane) ----
#pragma unroll
            for (int b = 0; b < 4; ++b) s_list[b][lane] = (uint16_t)LSR_WAVE;   // every list slot starts as the null record's slot
            const uint32_t rel = cbase + lane;  // 0-based position in the quadrant's list
            const uint32_t m = rel < maxlast ? (cur.w >> kQuadBitsShift) : 0u;
            if (m) {
                const float4 a = cur.a, b = cur.b;
                const FoldedConic f = fold_conic(a.z, a.w, b.x, b.y);   // same folding as the forward
                s_ent[lane][0] = make_float4(a.x, a.y, f.a2, f.c2);
                s_ent[lane][1] = make_float4(f.b2, f.l2o, b.z, __uint_as_float(rel + 1u));
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4) s_ent[lane][2 + c4] = cur.pay[c4];
                s_gid[lane] = cur.w & kQuadIndexMask;
            }
            const uint64_t staged = __ballot(m != 0);
            // compaction: per sub-block, the staged entries that can reach it, in list order.  An entry
            // that sits at the SAME position in the lists of two sub-blocks will be processed by two lane
            // groups in the same iteration; their updates of the entry's table row are ordered by a rank
            // (number of lower sub-blocks holding the entry at that position), computed here once per
            // staged entry and stored with the list element:   list element = staging slot | rank << 8.
            uint32_t nk = 0, at[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                at[b] = 0xFFFFu;           // 0xFFFF: not in this list
                const uint64_t bal = __ballot((m >> b) & 1u);
                nk = max(nk, (uint32_t)__builtin_popcountll(bal));
                if (__builtin_amdgcn_inverse_ballot_w64(bal))
                    at[b] = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            }
            nk = __builtin_amdgcn_readfirstlane(nk);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (at[j] == 0xFFFFu) continue;
                uint32_t rank = 0;
#pragma unroll
                for (int jj = 0; jj < j; ++jj) rank += (uint32_t)(at[jj] == at[j]);
                s_list[j][at[j]] = (uint16_t)((uint32_t)lane | (rank << 8));
            }
            wave_lds_fence_bwd();

            const uint16_t *lp = &s_list[grp][0];
            for (uint32_t i = 0; i < nk; ++i) {
                const uint32_t lel = lp[i];
                const uint32_t slot = lel & 0xFFu, rank = lel >> 8;
                const float4_b *E = (const float4_b *)&s_ent[slot][0];
                float *row = &s_acc[slot][tcol];
                float acc_old[NGRP];                             // this lane's word(s) of the entry's table row, read early
#pragma unroll
                for (int gi = 0; gi < NGRP; ++gi) acc_old[gi] = row[16 * gi];
                float4_b a = E[0], b = E[1], t4[NCHP / 4];
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4) t4[c4] = E[2 + c4];
                // Two or more lane groups can be at the SAME staged entry in this iteration (44 % of the
                // iterations on the bench scene): their read-modify-writes of the entry's table row are
                // ordered by the rank stored with the list element.  Rank 0 updates with the early-read
                // row; rank r > 0 re-reads the row after rank r-1 has written (a wave's LDS operations
                // execute in order).
                auto accumulate = [&](float *dst, float old, float tot, bool live) {
                    if (live && rank == 0u) *dst = old + tot;
                    uint64_t later = __ballot(rank != 0u);
                    for (uint32_t r = 1; later; ++r) {      // wave-uniform, usually no or one round
                        wave_lds_fence_bwd();
                        if (live && rank == r) *dst = *dst + tot;
                        later &= ~__ballot(rank == r);
                    }
                };
                // keep the record reads whole 16-byte LDS loads into aligned register tuples (left alone
                // the compiler splits them by use and re-pairs the packed operands with moves)
                asm volatile("" : "+v"(a), "+v"(b));
                float2_b pay[NCHP / 2];
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4) {
                    asm volatile("" : "+v"(t4[c4]));
                    pay[2 * c4] = float2_b{t4[c4].x, t4[c4].y}; pay[2 * c4 + 1] = float2_b{t4[c4].z, t4[c4].w};
                }
                // same operations as the forward's exponent (two of them packed)
                const float2_b d = float2_b{a.x, a.y} - pxy;
                const float2_b q = float2_b{a.z, a.w} * d;              // (a2 dx, c2 dy)
                const float p1 = __builtin_fmaf(b.x, d.y, q.x);
                const float p2 = __builtin_fmaf(q.y, d.y, b.y);
                const float ex = __builtin_fmaf(p1, d.x, p2);
                const float araw = fast_exp2(ex);
                const float aclamp = fminf(LSR_ALPHA_MAX, araw);
                const bool valid = (__float_as_uint(b.w) <= last) & (ex <= b.y) & (aclamp >= LSR_ALPHA_MIN);
                const float alpha = valid ? aclamp : 0.0f;
                const float av = valid ? araw : 0.0f;           // opacity * exp(power)
                const float om = 1.0f - alpha;
                const float rcp1m = __builtin_amdgcn_rcpf(om);
                const float Tk = Tr;                             // transmittance in front of this entry
                const float w = alpha * Tk;
                const float2_b ww = float2_b{w, w};
                float2_b ds2 = pay[0] * dpix[0];                 // g . c_i, two channels at a time
#pragma unroll
                for (int c = 1; c < NCHP / 2; ++c) ds2 = __builtin_elementwise_fma(pay[c], dpix[c], ds2);
                float dsum = ds2.x + ds2.y;
                float2_b gp[NCHP / 2];                           // dL/d payload channel pairs
#pragma unroll
                for (int c = 0; c < NCHP / 2; ++c) gp[c] = dpix[c] * ww;
                float gz = 0.0f;
                if (DEPTH_GRAD) {
                    dsum = __builtin_fmaf(b.z, ddep, dsum);
                    gz = w * ddep;
                }
                Rr = __builtin_fmaf(-w, dsum, Rr);               // what is left behind this entry
                Tr = Tk * om;
                const float dL_dalpha = __builtin_fmaf(Tk, dsum, -Rr * rcp1m);
                const float u = av * dL_dalpha;                  // straight through the 0.99 clamp (A.6)
                const float2_b t1 = float2_b{u, u} * d;          // u (dx, dy)
                const float2_b t2 = t1 * d;                      // u (dx^2, dy^2)
                const float mxy = t1.x * d.y;
                // ---- sum over the sub-block's 16 pixels; lane s of the group ends up with slot s ----
                {
                    constexpr uint32_t LIVE = 0x3Fu | (DEPTH_GRAD ? 0x40u : 0u) | (((1u << (NCHP < 8 ? NCHP : 8)) - 1u) << 8);
                    float tot;
                    if (!DEPTH_GRAD && NCHP == 4) tot = row_reduce_4ch(t1.x, t1.y, t2.x, mxy, t2.y, u, gp[0].x, gp[0].y, gp[1].x, gp[1].y);
                    else if (!DEPTH_GRAD && NCHP == 8)
                        tot = row_reduce_8ch(t1.x, t1.y, t2.x, mxy, t2.y, u, gp[0].x, gp[0].y, gp[1].x, gp[1].y,
                                             gp[2 % (NCHP / 2)].x, gp[2 % (NCHP / 2)].y, gp[3 % (NCHP / 2)].x, gp[3 % (NCHP / 2)].y);
                    else tot = row_reduce16_transposed<LIVE>(
                        t1.x, t1.y, t2.x, mxy, t2.y, u, gz, 0.0f,
                        gp[0].x, gp[0].y, gp[1].x, gp[1].y,
                        NCHP > 4 ? gp[2 % (NCHP / 2)].x : 0.0f, NCHP > 4 ? gp[2 % (NCHP / 2)].y : 0.0f,
                        NCHP > 4 ? gp[3 % (NCHP / 2)].x : 0.0f, NCHP > 4 ? gp[3 % (NCHP / 2)].y : 0.0f, l16);
                    // plain read-modify-write: only this wave touches its table and a wave's LDS operations
                    // execute in order (measured: ds_add_f32 costs ~120 LDS cycles per wave instruction; with
                    // it on every iteration the kernel was LDS bound at 1.44 ms)
                    accumulate(row, acc_old[0], tot, LIVE >> l16 & 1u);
                }
#pragma unroll
                for (int gi = 1; gi < NGRP; ++gi) {   // payload channels 16 gi - 8 .. 16 gi + 7
#define GPC(j) ((16 * gi - 8 + (j)) < NCHP ? gp[((16 * gi - 8 + (j)) / 2) % (NCHP / 2)][(j) & 1] : 0.0f)
                    const float tot = row_reduce16_transposed<0xFFFFu>(GPC(0), GPC(1), GPC(2), GPC(3), GPC(4), GPC(5), GPC(6), GPC(7),
                                                                        GPC(8), GPC(9), GPC(10), GPC(11), GPC(12), GPC(13), GPC(14), GPC(15), l16);
#undef GPC
                    accumulate(row + 16 * gi, acc_old[gi], tot, true);
                }
            }
            wave_lds_fence_bwd();
            // ---- flush: one global record-add per staged entry (every list entry reaches one of the quadrant's sub-blocks) ----
#pragma unroll 1
            for (int e0 = 0; e0 < LSR_WAVE; e0 += 4) {
                if (!((staged >> e0) & 0xFull)) continue;   // wave-uniform
                const int e = e0 + grp;
                const bool hit = (staged >> e) & 1ull;
                const uint32_t g = s_gid[e];
#pragma unroll
                for (int gi = 0; gi < NGRP; ++gi) {
                    const bool mine = !kPackRow || ((l16 & 14) != 6);   // slots 6 and 7 have no column in a packed row
                    const float val = mine ? s_acc[e][16 * gi + tcol] : 0.0f;
                    if (mine) s_acc[e][16 * gi + tcol] = 0.0f;
                    if (hit && val != 0.0f) {
                        const size_t at = (vG + g) * (size_t)RF + 16 * gi + l16;
                        if (p.rec_fixed)   // order-independent integer sum (LSR_DETERMINISTIC)
                            atomicAdd((unsigned long long *)p.rec_fixed + at, (unsigned long long)__double2ll_rn((double)val * kFixedPointScale));
                        else atomic_add_f32(p.rec + at, val);
                    }
                }
            }
            wave_lds_fence_bwd();
        }
    }  // item loop
}

__global__ void __launch_bounds__(256) k_fixed_to_float(const long long *__restrict__ in, float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (float)((double)in[i] * (1.0 / kFixedPointScale));
}

template <int NCHP, bool DG, int WPS>
static void launch_variant(const RenderBwdParams &p, hipStream_t s) {
    hipLaunchKernelGGL((k_render_bwd<NCHP, DG, WPS>), dim3(p.num_cus), dim3(LSR_WAVE * 4 * WPS), 0, s, p);
}

hipError_t launch_render_backward(const lsr_dims &d, const lsr_inputs &in, const char *geom,
                                  const char *bin, int64_t num_pairs, const char *img,
                                  const lsr_outputs &fwd, const lsr_out_grads &gout, char *grad,
                                  const lsr_in_grads &gin, hipStream_t s) {
    const GeomLayout L = geom_layout(d);
    const ImgLayout I = img_layout(d);
    const BinLayout B = bin_layout(d, num_pairs, 0);
    const GradLayout R = grad_layout(d);
    RenderBwdParams p;
    p.H = d.height; p.W = d.width; p.gx = tiles_x(d); p.T = (int)num_tiles(d); p.G = d.num_gaussians;
    p.C = d.feat_channels; p.has_color = d.color_mode != LSR_COLOR_NONE;
    p.num_cus = device_cus();
    p.items = (const uint32_t *)(geom + L.tile_order);
    p.header = (const uint32_t *)(geom + L.header);
    p.views = in.views;
    p.geo = (const float4 *)(geom + L.rec); p.rec_f4 = L.rec_floats / 4;
    p.tile_start = (const uint32_t *)(geom + L.tile_start);
    p.quad_list = (const uint32_t *)(bin + B.quad_list);
    p.final_T = (const float *)(img + I.final_T); p.n_contrib = (const uint32_t *)(img + I.n_contrib);
    p.g_color = gout.color; p.g_feat = gout.feature; p.g_mask = gout.mask; p.g_depth = gout.depth;
    p.f_color = fwd.color; p.f_feat = fwd.feature; p.f_depth = fwd.depth;
    p.rec = (float *)(grad + R.rec); p.rec_floats = R.rec_floats;
    const bool det = deterministic_backward();
    p.rec_fixed = det ? (long long *)(grad + R.fixed) : nullptr;
    p.queue = (uint32_t *)(grad + R.fixed - 256);   // the zeroed slack behind the float records
    (void)gin;
    const int nch = (p.has_color ? 3 : 0) + d.feat_channels;
    const int nchp = nch <= 4 ? 4 : (nch <= 8 ? 8 : (nch <= 12 ? 12 : 36));
    const bool dg = gout.depth != nullptr;
    prof_begin(kStRenderBwd, s);
#define LSR_RB(N, W)                                             \
    do {                                                         \
        if (dg) launch_variant<N, true, W>(p, s);                \
        else launch_variant<N, false, W>(p, s);                  \
    } while (0)
    if (nchp == 4) LSR_RB(4, 4);
    else if (nchp == 8 && !dg) launch_variant<8, false, 4>(p, s);   // packed table rows: 16 waves per CU fit
    else if (nchp == 8) launch_variant<8, true, 3>(p, s);
    else if (nchp == 12) LSR_RB(12, 2);
    else LSR_RB(36, 1);
#undef LSR_RB
    if (det) {   // fixed-point sums -> the float records the next stages read
        const size_t n = (size_t)d.num_views * (size_t)d.num_gaussians * (size_t)R.rec_floats;
        const size_t blocks = (n + 255) / 256;
        hipLaunchKernelGGL(k_fixed_to_float, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s,
                           (const long long *)(grad + R.fixed), p.rec, n);
    }
    prof_end(kStRenderBwd, s);
    return hipGetLastError();
}

}  // namespace lsr
