// lsr_project.h — the per-(view, Gaussian) projection of stage K1: view transform,
// near cull, clip-space projection, EWA covariance (clamped Jacobian), low-pass, conic, radius, tile rectangle.
// Shared by k_preprocess (thread = Gaussian, loop over the block's views) and the fused projection + SH payload kernel
// (thread = (view, Gaussian), sh.hip); both are compiled with -ffp-contract=off: radius / rectangle / depth bits feed
// the bit-exact tile lists, so every float operation is the single IEEE operation written here — or, with FMA = true,
// the contracted form described below — and the CPU oracle performs the identical sequence.
#pragma once
#include "lsr_internal.h"

namespace lsr {

__device__ __forceinline__ float fmin_sel(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float fmax_sel(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ int imin_sel(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_sel(int a, int b) { return a > b ? a : b; }

// ---- arithmetic convention of the projection (lsr_set_projection_contraction) ----
// FMA = false: the published source with every float operation a separate IEEE operation — the convention of the
// oracle and of the bit-exact index tests.
// FMA = true : the same source as a compiler with contraction ON builds it (nvcc's default -fmad=true): a product
// that feeds a sum is fused into it.  The rule applied is LLVM's default combine on the published expression trees
// (glm's mat3 products of computeCov2D INCLUDING their terms with a literal zero factor, which decide which product
// of a sum stays rounded):  x*y + z -> fma(x, y, z);  z + x*y -> fma(x, y, z);  x*y - z -> fma(x, y, -z);  sums
// associate left to right, so  p1 + p2 + p3 -> fma(a3, b3, fma(a1, b1, a2*b2)).
// What nvcc / ptxas really emit for the fork cannot be known here; the switch exists to measure how much of the
// "bit-exact" index contract depends on the answer, and to flip the default in one commit once fork vectors say so.
template <bool FMA> __device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return FMA ? __builtin_fmaf(a2, b2, __builtin_fmaf(a0, b0, a1 * b1)) : a0 * b0 + a1 * b1 + a2 * b2;
}
// glm's  w0*j0 + w1*0 + w2*j2  (a column of J with one zero): contracted, the zero term swallows the fusion of the
// first product (fma(w0, j0, w1*0) = round(w0*j0)) and the LAST product is the fused one
template <bool FMA> __device__ __forceinline__ float dot2z(float j0, float w0, float j2, float w2) {
    return FMA ? __builtin_fmaf(w2, j2, w0 * j0) : j0 * w0 + j2 * w2;
}
template <bool FMA> __device__ __forceinline__ float ndc2pix(float v, int S) {
    return FMA ? (float)(__builtin_fma(v + 1.0, (double)S, -1.0) * 0.5) : (float)(((v + 1.0) * S - 1.0) * 0.5);
}

struct Projected {
    bool ok;                              // survives the near cull, has a finite conic and touches at least one tile
    float px, py;                         // pixel-space mean
    float conic_a, conic_b, conic_c;
    float tz;                             // view-space depth (the sort key's bits)
    float radius;                         // ceil(3 sigma_max), pixels
    int rminx, rminy, rmaxx, rmaxy;       // tile rectangle [min, max)
};

// vm / pm: the view and projection matrices of the camera table (memory = transposed matrices); (p0, p1, p2): the mean
// already multiplied by the scene scale; s0..s5: the packed covariance already multiplied by its square.
// ONE visibility predicate instead of nested early exits (each exit level made the compiler re-materialise the zeroed
// outputs): culled lanes run the arithmetic on whatever they hold (IEEE special values are harmless here) and are
// masked where results leave the thread.
template <bool FMA>
__device__ __forceinline__ Projected project_gaussian(const float (&vm)[16], const float (&pm)[16], float limx, float limy,
                                                      float focal_x, float focal_y, float p0, float p1, float p2,
                                                      float s0, float s1, float s2, float s3, float s4, float s5,
                                                      int width, int height, int gx, int gy, bool in_range) {
    Projected r;
    const float t0 = dot3<FMA>(vm[0], p0, vm[4], p1, vm[8], p2) + vm[12];
    const float t1 = dot3<FMA>(vm[1], p0, vm[5], p1, vm[9], p2) + vm[13];
    const float t2 = dot3<FMA>(vm[2], p0, vm[6], p1, vm[10], p2) + vm[14];
    bool ok = in_range && !(t2 <= LSR_NEAR_CULL);
    const float h0 = dot3<FMA>(pm[0], p0, pm[4], p1, pm[8], p2) + pm[12];
    const float h1 = dot3<FMA>(pm[1], p0, pm[5], p1, pm[9], p2) + pm[13];
    const float h3 = dot3<FMA>(pm[3], p0, pm[7], p1, pm[11], p2) + pm[15];
    const float p_w = 1.0f / (h3 + 0.0000001f);
    const float ndc_x = h0 * p_w, ndc_y = h1 * p_w;

    const float txtz = t0 / t2, tytz = t1 / t2;
    const float tx = fmin_sel(limx, fmax_sel(-limx, txtz)) * t2;
    const float ty = fmin_sel(limy, fmax_sel(-limy, tytz)) * t2;
    const float tz = t2;
    const float j00 = focal_x / tz, j02 = -(focal_x * tx) / (tz * tz);
    const float j11 = focal_y / tz, j12 = -(focal_y * ty) / (tz * tz);
    const float m00 = dot2z<FMA>(j00, vm[0], j02, vm[2]);
    const float m01 = dot2z<FMA>(j00, vm[4], j02, vm[6]);
    const float m02 = dot2z<FMA>(j00, vm[8], j02, vm[10]);
    const float m10 = dot2z<FMA>(j11, vm[1], j12, vm[2]);
    const float m11 = dot2z<FMA>(j11, vm[5], j12, vm[6]);
    const float m12 = dot2z<FMA>(j11, vm[9], j12, vm[10]);
    const float v00 = dot3<FMA>(s0, m00, s1, m01, s2, m02);
    const float v01 = dot3<FMA>(s1, m00, s3, m01, s4, m02);
    const float v02 = dot3<FMA>(s2, m00, s4, m01, s5, m02);
    const float v10 = dot3<FMA>(s0, m10, s1, m11, s2, m12);
    const float v11 = dot3<FMA>(s1, m10, s3, m11, s4, m12);
    const float v12 = dot3<FMA>(s2, m10, s4, m11, s5, m12);
    const float ca = dot3<FMA>(m00, v00, m01, v01, m02, v02) + LSR_LOWPASS;
    const float cb = dot3<FMA>(m00, v10, m01, v11, m02, v12);
    const float cc = dot3<FMA>(m10, v10, m11, v11, m12, v12) + LSR_LOWPASS;
    const float det = FMA ? __builtin_fmaf(ca, cc, -(cb * cb)) : ca * cc - cb * cb;
    ok = ok && !(det == 0.0f);
    const float det_inv = 1.0f / det;
    r.conic_a = cc * det_inv; r.conic_b = -cb * det_inv; r.conic_c = ca * det_inv;
    const float mid = 0.5f * (ca + cc);
    const float disc = sqrtf(fmax_sel(0.1f, FMA ? __builtin_fmaf(mid, mid, -det) : mid * mid - det));
    const float lambda1 = mid + disc, lambda2 = mid - disc;
    const float my_radius = ceilf(3.0f * sqrtf(fmax_sel(lambda1, lambda2)));
    const float px = ndc2pix<FMA>(ndc_x, width), py = ndc2pix<FMA>(ndc_y, height);
    r.rminx = imin_sel(gx, imax_sel(0, (int)((px - my_radius) / LSR_TILE)));
    r.rminy = imin_sel(gy, imax_sel(0, (int)((py - my_radius) / LSR_TILE)));
    // ((p + r) + 16) - 1, in THIS order: the published expression `p.x + max_radius + BLOCK_X - 1` is evaluated left
    // to right in float, and (p + r) + 15 rounds differently when p sits just below a pixel centre
    // (x = 233.99998, r = 7: 256.99998 rounds up to 257 -> tile 16; 255.99998 is exact -> tile 15)
    r.rmaxx = imin_sel(gx, imax_sel(0, (int)((((px + my_radius) + (float)LSR_TILE) - 1.0f) / LSR_TILE)));
    r.rmaxy = imin_sel(gy, imax_sel(0, (int)((((py + my_radius) + (float)LSR_TILE) - 1.0f) / LSR_TILE)));
    ok = ok && (r.rmaxx - r.rminx) * (r.rmaxy - r.rminy) != 0;
    r.ok = ok; r.px = px; r.py = py; r.tz = tz; r.radius = my_radius;
    return r;
}

}  // namespace lsr
