// lsr_blend.h — per-(pixel, Gaussian) blending arithmetic shared by the forward and backward
// compositing kernels, so both make bit-identical alpha / skip decisions (the backward rebuilds
// the transmittance by dividing out exactly the alphas the forward multiplied in), and the
// footprint-span / sub-block-code helpers shared by k_preprocess (writes the span), k_scatter (per-pair
// code in the sort key) and k_sort_tiles (turns the codes into the two half-tile render lists of a tile).
//
// Work decomposition (CDNA4, wave64): a 16x16 tile is two 16x8 halves, a half eight 4x4-pixel
// SUB-BLOCKS.  One wave renders one half with TWO horizontally adjacent pixels per lane: the 8-lane
// group g = lane >> 3 owns sub-block (g & 3, g >> 2) of the half, lane l of the group the pixels
// (2 (l & 1) + {0, 1}, l >> 1) of the sub-block.  A list entry is evaluated only by the lane groups whose
// sub-block its alpha >= 1/255 footprint box can reach.  Skipping is lossless: a culled (entry,
// sub-block) would fail the alpha test at every pixel of the sub-block.
#pragma once
#include "lsr_internal.h"

namespace lsr {

constexpr float kLog2e = 1.4426950408889634f;

// ---- per-(pixel, Gaussian) blending arithmetic --------------------------------------------------------
// alpha_raw = opacity * exp(power), power = -0.5 (A dx^2 + C dy^2) - B dx dy.  The kernels work in the
// exponent domain, shifted so that the alpha >= 1/255 threshold sits at zero:
//   e' = a2 dx^2 + b2 dx dy + c2 dy^2 + l2o',   (a2,b2,c2) = log2(e) * (-A/2, -B, -C/2),   l2o' = log2(255 opacity)
//   255 alpha_raw = exp2(e')
// and the two skip tests of the published algorithm become ONE unsigned integer comparison of float bits
//   keep  <=>  bits(e') <= bits(l2o')      (0 <= e': alpha >= 1/255;  e' <= l2o': power <= 0;  NaN and negative e' have larger bits)
// (l2o' >= 0 for every list entry: Gaussians with opacity < 1/255 reach no pixel and are never listed).
// Everything downstream stays in units of 255 alpha: staged payloads and depths carry the factor 1/255, the
// transmittance update is T -= w' / 255.  Evaluated per lane for two horizontally adjacent pixels:
//   e'(dx) = dx (a2 dx + b2 dy) + (c2 dy^2 + l2o'),   dx = dx0, dx0 - 1
// with the row terms (dy, b2 dy, c2 dy^2 + l2o') shared.  The forward and the backward perform the identical
// operation sequence: both must make bit-identical keep / skip decisions.
constexpr float kLog2_255 = 7.994353436858858f;
constexpr float kInv255 = 1.0f / 255.0f;
constexpr float kAlphaMax255 = LSR_ALPHA_MAX * 255.0f;

__device__ __forceinline__ float fast_exp2(float e) { return __builtin_amdgcn_exp2f(e); }

struct FoldedConic { float a2, b2, c2, l2o; };
__device__ __forceinline__ FoldedConic fold_conic(float A, float B, float C, float o) {
    FoldedConic f;
    f.a2 = (-0.5f * kLog2e) * A;
    f.b2 = (-kLog2e) * B;
    f.c2 = (-0.5f * kLog2e) * C;
    f.l2o = __log2f(o) + kLog2_255;      // log2(255 o)
    return f;
}

// Conservative footprint of a screen-space Gaussian: axis-aligned bounding box of
// { d : 1/2 d^T Q d <= ln(255 o) } around (x, y) in pixels, returned as the packed cell span of
// lsr_internal.h relative to cell (4 * rminx, 4 * rminy), the first cell of the tile rectangle.
// Hardware log2 / rcp / sqrt (1 ulp): the box carries 0.1 % + 0.05 px of slack, so it contains every
// pixel centre with alpha >= 1/255.  Cell c is reached iff lo <= 4c+3 && hi >= 4c.
__device__ __forceinline__ uint32_t footprint_cells(float x, float y, float A, float B, float C, float o, int rminx, int rminy) {
    const float det = A * C - B * B;
    const float tau = __builtin_amdgcn_logf(255.0f * o) * (0.6931471806f * 1.0001f) + 1e-4f;
    const float s = 2.0f * tau * __builtin_amdgcn_rcpf(det);
    const float ex = __builtin_amdgcn_sqrtf(s * C) * 1.001f + 0.05f;
    const float ey = __builtin_amdgcn_sqrtf(s * A) * 1.001f + 0.05f;
    // first / last reached cell, relative; clamped in float so the conversions are safe for huge footprints
    const float ox = 4.0f * (float)rminx, oy = 4.0f * (float)rminy;
    const float fx0 = __builtin_ceilf((x - ex - 3.0f) * 0.25f) - ox, fx1 = __builtin_floorf((x + ex) * 0.25f) - ox;
    const float fy0 = __builtin_ceilf((y - ey - 3.0f) * 0.25f) - oy, fy1 = __builtin_floorf((y + ey) * 0.25f) - oy;
    // alpha = min(.99, o*G) <= o < 1/255 everywhere (also NaN opacity), or entirely left of / above the rectangle
    const bool none = !(o >= LSR_ALPHA_MIN) || fx1 < 0.0f || fy1 < 0.0f;
    // non-positive-definite or badly conditioned conics (the box would not be trustworthy), NaN anywhere
    const bool all = !(det > 0.0f) || !(A * C < 1000.0f * det) || !(fx0 == fx0) || !(fx1 == fx1) || !(fy0 == fy0) || !(fy1 == fy1);
    const uint32_t x0 = (uint32_t)fminf(fmaxf(fx0, 0.0f), 255.0f), x1 = (uint32_t)fminf(fmaxf(fx1, 0.0f), 255.0f);
    const uint32_t y0 = (uint32_t)fminf(fmaxf(fy0, 0.0f), 255.0f), y1 = (uint32_t)fminf(fmaxf(fy1, 0.0f), 255.0f);
    const uint32_t span = x0 | (x1 << 8) | (y0 << 16) | (y1 << 24);
    return none ? kSpanNone : (all ? kSpanAll : span);
}
// 8-bit sub-block code (lsr_internal.h) of the pair (span, tile at offset (dx, dy) tiles from the rectangle's first tile)
__device__ __forceinline__ uint32_t span_code(uint32_t span, int dx, int dy) {
    const int x0 = (int)(span & 0xFFu), x1 = (int)((span >> 8) & 0xFFu), y0 = (int)((span >> 16) & 0xFFu), y1 = (int)(span >> 24);
    const int c0 = max(0, x0 - 4 * dx), c1 = x1 == 255 ? 3 : min(3, x1 - 4 * dx);
    const int r0 = max(0, y0 - 4 * dy), r1 = y1 == 255 ? 3 : min(3, y1 - 4 * dy);
    const bool some = x0 <= x1 && c0 <= c1 && r0 <= r1;      // (c0 can exceed 3, c1 / r1 can be negative: all "none")
    return some ? (uint32_t)(c0 | (c1 << 2) | (r0 << 4) | (r1 << 6)) : kCodeNone;
}
// LSR_FWD_REACHED_ONLY: the tiles of the rectangle [x0, x1) x [y0, y1) the footprint span reaches form a sub-rectangle (the span is
// an axis-aligned box of 4-pixel cells, four cells to a tile): tile offset dx is reached iff X0 >> 2 <= dx <= X1 >> 2 — exactly
// the pairs whose span_code is not kCodeNone.  Shrinks the rectangle in place (empty when the span is kSpanNone).
__device__ __forceinline__ void reached_rect(uint32_t span, int &x0, int &y0, int &x1, int &y1) {
    const int sx0 = (int)(span & 0xFFu), sx1 = (int)((span >> 8) & 0xFFu), sy0 = (int)((span >> 16) & 0xFFu), sy1 = (int)(span >> 24);
    const int nx1 = sx1 == 255 ? x1 : min(x1, x0 + (sx1 >> 2) + 1), ny1 = sy1 == 255 ? y1 : min(y1, y0 + (sy1 >> 2) + 1);
    x0 += sx0 >> 2; y0 += sy0 >> 2;
    x1 = sx0 <= sx1 ? nx1 : x0;       // (first cell beyond the last: a footprint between two pixel centres reaches nothing)
    y1 = sy0 <= sy1 ? ny1 : y0;
}
// 16-bit mask of the tile's 4x4-pixel sub-blocks a code stands for; bit (4*row + col) <-> sub-block with origin (4*col, 4*row)
__device__ __forceinline__ uint32_t code_mask(uint32_t code) {
    const uint32_t c0 = code & 3u, c1 = (code >> 2) & 3u, r0 = (code >> 4) & 3u, r1 = (code >> 6) & 3u;
    const uint32_t cm = c0 <= c1 ? ((2u << c1) - (1u << c0)) : 0u;                 // columns c0..c1
    const uint32_t rows = r0 <= r1 ? ((2u << r1) - (1u << r0)) : 0u;
    return cm * (((rows & 1u) ? 0x1u : 0u) | ((rows & 2u) ? 0x10u : 0u) | ((rows & 4u) ? 0x100u : 0u) | ((rows & 8u) ? 0x1000u : 0u));
}
// the eight sub-blocks of half h (0: pixel rows 0-7, 1: rows 8-15) as an 8-bit mask: bit (4*r + c) <-> sub-block (c, r) of the half
__device__ __forceinline__ uint32_t half_bits(uint32_t m16, int h) { return (m16 >> (8 * h)) & 0xFFu; }

__device__ __forceinline__ float f_from_u(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned u_from_f(float f) { return __builtin_bit_cast(unsigned, f); }

}  // namespace lsr
