// lsr_blend.h — per-(pixel, Gaussian) blending arithmetic shared by the forward and backward
// compositing kernels, so both make bit-identical alpha / skip decisions (the backward rebuilds
// the transmittance by dividing out exactly the alphas the forward multiplied in).
//
// Work decomposition (CDNA4, wave64): a 16x16 tile is four 8x8 quadrants.  One wave owns PXL of
// them (1, 2 or 4) with lane l at position (l & 7, l >> 3) inside each owned quadrant, i.e. PXL
// pixels per lane.  Waves never synchronise with each other: each stages 64 list entries at a
// time into its own LDS slice and then walks only the entries whose exact alpha >= 1/255
// footprint can reach one of its quadrants (per-entry 4-bit quadrant mask, wave-uniform
// branches).  Skipping is lossless: a culled (entry, quadrant) would fail the alpha test at every
// pixel of the quadrant.
#pragma once
#include "lsr_internal.h"

namespace lsr {

constexpr float kLog2e = 1.4426950408889634f;

// alpha_raw = opacity * exp(power), power = -0.5 (A dx^2 + C dy^2) - B dx dy, evaluated as
// exp2(e) with e = a2 dx^2 + b2 dx dy + c2 dy^2 + log2(opacity), (a2,b2,c2) = log2(e)*(-A/2,-B,-C/2).
__device__ __forceinline__ float blend_exponent(float dx, float dy, float a2, float b2, float c2, float l2o) {
    float p1 = a2 * dx;
    p1 = __builtin_fmaf(b2, dy, p1);
    float p2 = c2 * dy;
    p2 = __builtin_fmaf(p2, dy, l2o);
    return __builtin_fmaf(p1, dx, p2);
}
__device__ __forceinline__ float fast_exp2(float e) { return __builtin_amdgcn_exp2f(e); }

struct FoldedConic { float a2, b2, c2, l2o; };
__device__ __forceinline__ FoldedConic fold_conic(float A, float B, float C, float o) {
    FoldedConic f;
    f.a2 = (-0.5f * kLog2e) * A;
    f.b2 = (-kLog2e) * B;
    f.c2 = (-0.5f * kLog2e) * C;
    f.l2o = __log2f(o);
    return f;
}

// Conservative 4-bit mask of the tile's 8x8 quadrants that the Gaussian can touch with
// alpha >= 1/255: axis-aligned bounding box of { d : 1/2 d^T Q d <= ln(255 o) } (+ slack).
// bit q <-> quadrant with origin (8*(q&1), 8*(q>>1)).
__device__ __forceinline__ uint32_t quadrant_mask(float x, float y, float A, float B, float C, float o,
                                                  float tile_x0, float tile_y0) {
    if (!(o >= LSR_ALPHA_MIN)) return 0u;  // alpha = min(.99, o*G) <= o < 1/255 everywhere (also NaN)
    const float det = A * C - B * B;
    // Fall back to "all quadrants" for non-positive-definite or badly conditioned conics, where
    // the box computed from (A,B,C) would not be trustworthy.
    if (!(det > 0.0f) || !(A * C < 1000.0f * det)) return 0xFu;
    const float tau = __logf(255.0f * o) * 1.0001f + 1e-4f;
    const float s = 2.0f * tau / det;
    const float ex = __fsqrt_rn(s * C) * 1.001f + 0.05f;
    const float ey = __fsqrt_rn(s * A) * 1.001f + 0.05f;
    const float x0 = x - ex - tile_x0, x1 = x + ex - tile_x0;
    const float y0 = y - ey - tile_y0, y1 = y + ey - tile_y0;
    if (!(x0 == x0) || !(x1 == x1) || !(y0 == y0) || !(y1 == y1)) return 0xFu;
    const bool xl = x0 <= 7.0f && x1 >= 0.0f;    // columns 0..7
    const bool xr = x0 <= 15.0f && x1 >= 8.0f;   // columns 8..15
    const bool yt = y0 <= 7.0f && y1 >= 0.0f;    // rows 0..7
    const bool yb = y0 <= 15.0f && y1 >= 8.0f;   // rows 8..15
    return (uint32_t)(xl && yt) | ((uint32_t)(xr && yt) << 1) | ((uint32_t)(xl && yb) << 2) |
           ((uint32_t)(xr && yb) << 3);
}

// Finer version of the same test: 16-bit mask of the tile's 4x4-pixel sub-blocks the footprint box
// can reach; bit (4*row + col) <-> sub-block with origin (4*col, 4*row).  The compositing kernels
// give every 16-lane group of a wave its own sub-block (one pixel per lane), so a staged entry is
// only evaluated by the lane groups whose sub-block it can touch: on the bench scene 39 pixel
// evaluations per (Gaussian, tile) pair instead of 77 with 8x8 quadrants (15 of them pass the
// alpha test).  Same conservative box as quadrant_mask, hence lossless.
__device__ __forceinline__ uint32_t span_mask4(float lo, float hi) {
    // cells c = 0..3 cover pixel centres 4c .. 4c+3; cell c is reached iff lo <= 4c+3 && hi >= 4c
    const int c0 = max(0, (int)__builtin_ceilf((lo - 3.0f) * 0.25f));
    const int c1 = min(3, (int)__builtin_floorf(hi * 0.25f));
    return c0 <= c1 ? ((2u << c1) - (1u << c0)) : 0u;
}
__device__ __forceinline__ uint32_t subblock_mask(float x, float y, float A, float B, float C, float o,
                                                  float tile_x0, float tile_y0) {
    if (!(o >= LSR_ALPHA_MIN)) return 0u;
    const float det = A * C - B * B;
    if (!(det > 0.0f) || !(A * C < 1000.0f * det)) return 0xFFFFu;
    // hardware log2 / rcp / sqrt (1 ulp): the box carries 0.1 % + 0.05 px of slack
    const float tau = __builtin_amdgcn_logf(255.0f * o) * (0.6931471806f * 1.0001f) + 1e-4f;
    const float s = 2.0f * tau * __builtin_amdgcn_rcpf(det);
    const float ex = __builtin_amdgcn_sqrtf(s * C) * 1.001f + 0.05f;
    const float ey = __builtin_amdgcn_sqrtf(s * A) * 1.001f + 0.05f;
    const float x0 = x - ex - tile_x0, x1 = x + ex - tile_x0;
    const float y0 = y - ey - tile_y0, y1 = y + ey - tile_y0;
    if (!(x0 == x0) || !(x1 == x1) || !(y0 == y0) || !(y1 == y1)) return 0xFFFFu;
    if (!(x0 <= 15.0f && x1 >= 0.0f && y0 <= 15.0f && y1 >= 0.0f)) return 0u;
    // clamp before the float -> int conversions (huge footprints)
    const uint32_t cm = span_mask4(fmaxf(x0, -8.0f), fminf(x1, 24.0f));
    const uint32_t rm = span_mask4(fmaxf(y0, -8.0f), fminf(y1, 24.0f));
    return ((rm & 1u) ? cm : 0u) | ((rm & 2u) ? cm << 4 : 0u) | ((rm & 4u) ? cm << 8 : 0u) | ((rm & 8u) ? cm << 12 : 0u);
}
// sub-blocks of quadrant q (origin (8*(q&1), 8*(q>>1))): bits {r*4+c : r in 2*(q>>1)+{0,1}, c in 2*(q&1)+{0,1}}
__host__ __device__ constexpr uint32_t quadrant_subblocks(int q) { return 0x33u << (8 * (q >> 1) + 2 * (q & 1)); }
__device__ __forceinline__ uint32_t own_subblocks(uint32_t own) {
    return ((own & 1u) ? quadrant_subblocks(0) : 0u) | ((own & 2u) ? quadrant_subblocks(1) : 0u) |
           ((own & 4u) ? quadrant_subblocks(2) : 0u) | ((own & 8u) ? quadrant_subblocks(3) : 0u);
}

// Quadrants owned by wave `part` of a tile for a given pixels-per-lane setting.
template <int PXL>
__device__ __forceinline__ int owned_quadrant(int part, int k) {
    return PXL == 4 ? k : (PXL == 2 ? 2 * part + k : part);
}
template <int PXL>
__device__ __forceinline__ uint32_t owned_mask(int part) {
    return PXL == 4 ? 0xFu : (PXL == 2 ? (0x3u << (2 * part)) : (1u << part));
}

// Wave-wide sum (wave64) with DPP; the total ends up in lanes 48..63.
#define LSR_DPP_ADD(v, ctrl, rmask)                                                              \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, \
                                                                rmask, 0xf, false))
__device__ __forceinline__ float wave_sum_to_row3(float v) {
    LSR_DPP_ADD(v, 0xB1, 0xf);   // quad_perm [1,0,3,2]
    LSR_DPP_ADD(v, 0x4E, 0xf);   // quad_perm [2,3,0,1]
    LSR_DPP_ADD(v, 0x141, 0xf);  // row_half_mirror
    LSR_DPP_ADD(v, 0x140, 0xf);  // row_mirror
    LSR_DPP_ADD(v, 0x142, 0xa);  // row_bcast:15 into rows 1,3
    LSR_DPP_ADD(v, 0x143, 0xc);  // row_bcast:31 into rows 2,3
    return v;
}

// ---- transposed wave reduction ------------------------------------------------------------
// Sums 16 per-lane values across the 64 lanes of a wave in ~2 instructions per value instead of
// 6: every butterfly step halves the number of values a lane still carries.  Afterwards lane l
// holds the wave-wide total of slot (l >> 2) (replicated over its quad), so ONE atomic
// instruction with 16 active lanes can add a whole 64-byte gradient record.
//   step A  v_permlane32_swap : lanes 0-31 keep slot i, lanes 32-63 keep slot i+8
//   step B  v_permlane16_swap : even rows keep slot i, odd rows slot i+4
//   step C  DPP row_ror:8     : lane bit 3 selects slot i / i+2
//   step D  DPP row_half_mirror: lane bit 2 selects slot i / i+1
//   then the two quad_perm steps finish the sum inside each quad.
// LIVE is a compile-time bitmask of slots that can be non-zero; dead pairs cost nothing.
__device__ __forceinline__ float f_from_u(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned u_from_f(float f) { return __builtin_bit_cast(unsigned, f); }

template <uint32_t LIVE>
__device__ __forceinline__ float wave_reduce16_transposed(const float (&v)[16], int lane) {
    float r[8], s[4], t[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if ((LIVE >> i & 1u) || (LIVE >> (i + 8) & 1u)) {
            const auto w = __builtin_amdgcn_permlane32_swap(u_from_f(v[i]), u_from_f(v[i + 8]), false, false);
            r[i] = f_from_u(w[0]) + f_from_u(w[1]);
        } else r[i] = 0.0f;
    }
    constexpr uint32_t LA = (LIVE | (LIVE >> 8)) & 0xFFu;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((LA >> i & 1u) || (LA >> (i + 4) & 1u)) {
            const auto w = __builtin_amdgcn_permlane16_swap(u_from_f(r[i]), u_from_f(r[i + 4]), false, false);
            s[i] = f_from_u(w[0]) + f_from_u(w[1]);
        } else s[i] = 0.0f;
    }
    constexpr uint32_t LB = (LA | (LA >> 4)) & 0xFu;
    const bool b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if ((LB >> i & 1u) || (LB >> (i + 2) & 1u)) {
            const float keep = b3 ? s[i + 2] : s[i], send = b3 ? s[i] : s[i + 2];
            t[i] = keep + f_from_u(__builtin_amdgcn_update_dpp(0, u_from_f(send), 0x128, 0xf, 0xf, false));  // row_ror:8
        } else t[i] = 0.0f;
    }
    const float keep = b2 ? t[1] : t[0], send = b2 ? t[0] : t[1];
    float u = keep + f_from_u(__builtin_amdgcn_update_dpp(0, u_from_f(send), 0x141, 0xf, 0xf, false));  // row_half_mirror
    LSR_DPP_ADD(u, 0xB1, 0xf);  // quad_perm [1,0,3,2]
    LSR_DPP_ADD(u, 0x4E, 0xf);  // quad_perm [2,3,0,1]
    return u;
}

}  // namespace lsr
