// render_forward.hip — stage K6: front-to-back alpha compositing of the depth-sorted tile lists.
// Outputs colour (+ bg), features (bg 0), mask = 1 - T, depth = sum alpha T z, and keeps
// final_T / n_contrib for the backward pass.  Spec: SURVEY.md Appendix A.3 step 7 + A.4.
// Work decomposition and the lossless quadrant culling are described in lsr_blend.h.
#include "lsr_blend.h"

namespace lsr {

struct RenderFwdParams {
    int H, W, gx, T, G, C, has_color;
    int64_t vs_feat;
    const float *views;
    const float4 *q0, *q1, *rgb;
    const float *features;
    const uint32_t *tile_start, *point_list;
    float *out_color, *out_feat, *out_mask, *out_depth;
    float *final_T;
    uint32_t *n_contrib;
};

template <int NCHP, int PXL>
__global__ void __launch_bounds__(LSR_WAVE)
k_render_fwd(RenderFwdParams p) {
    constexpr int NW = 4 / PXL;  // waves (= workgroups) per tile
    __shared__ float4 s_q0[LSR_WAVE];
    __shared__ float4 s_q1[LSR_WAVE];
    __shared__ float4 s_pay[LSR_WAVE][NCHP / 4];

    const int lane = threadIdx.x;
    const int tile = blockIdx.x / NW, part = blockIdx.x % NW;
    const int v = blockIdx.y;
    const int tx0 = (tile % p.gx) * LSR_TILE, ty0 = (tile / p.gx) * LSR_TILE;
    const size_t vG = (size_t)v * p.G;
    const uint32_t start = p.tile_start[(size_t)v * p.T + tile];
    const uint32_t end = p.tile_start[(size_t)v * p.T + tile + 1];
    const uint32_t own = owned_mask<PXL>(part);
    const int coff = p.has_color ? 3 : 0;

    float pxf[PXL], pyf[PXL], Tr[PXL], accd[PXL];
    float acc[PXL][NCHP];
    uint32_t last[PXL];
    bool done[PXL], inside[PXL];
#pragma unroll
    for (int k = 0; k < PXL; ++k) {
        const int q = owned_quadrant<PXL>(part, k);
        const int px = tx0 + 8 * (q & 1) + (lane & 7), py = ty0 + 8 * (q >> 1) + (lane >> 3);
        pxf[k] = (float)px; pyf[k] = (float)py;
        inside[k] = px < p.W && py < p.H;
        done[k] = !inside[k];
        Tr[k] = 1.0f; accd[k] = 0.0f; last[k] = 0;
#pragma unroll
        for (int c = 0; c < NCHP; ++c) acc[k][c] = 0.0f;
    }

    for (uint32_t base = start; base < end; base += LSR_WAVE) {
        bool all_done = true;
#pragma unroll
        for (int k = 0; k < PXL; ++k) all_done = all_done && done[k];
        if (__all(all_done)) break;

        // ---- stage up to 64 list entries (one per lane) ----
        const uint32_t e = base + lane;
        uint32_t m = 0;
        if (e < end) {
            const uint32_t g = p.point_list[e];
            const float4 a = p.q0[vG + g], b = p.q1[vG + g];  // (x,y,A,B) (C,o,z,-)
            m = quadrant_mask(a.x, a.y, a.z, a.w, b.x, b.y, (float)tx0, (float)ty0) & own;
            if (m) {
                const FoldedConic f = fold_conic(a.z, a.w, b.x, b.y);
                s_q0[lane] = make_float4(a.x, a.y, f.a2, f.b2);
                s_q1[lane] = make_float4(f.c2, f.l2o, b.z, __uint_as_float(m));
                float pay[NCHP];
#pragma unroll
                for (int c = 0; c < NCHP; ++c) pay[c] = 0.0f;
                if (p.has_color) {
                    const float4 col = p.rgb[vG + g];
                    pay[0] = col.x; pay[1] = col.y; pay[2] = col.z;
                }
                const float *fp = p.features + (size_t)v * p.vs_feat + (size_t)g * p.C;
#pragma unroll
                for (int c = 0; c < NCHP; ++c)
                    if (c >= coff && c - coff < p.C) pay[c] = fp[c - coff];
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4)
                    s_pay[lane][c4] = make_float4(pay[4 * c4], pay[4 * c4 + 1], pay[4 * c4 + 2], pay[4 * c4 + 3]);
            }
        }
        uint64_t todo = __ballot(m != 0);
        __syncthreads();  // single-wave workgroup: orders the LDS writes above before the reads below

        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 a = s_q0[j], b = s_q1[j];
            const uint32_t mj = __builtin_amdgcn_readfirstlane(__float_as_uint(b.w));
            const uint32_t pos = base - start + (uint32_t)j + 1u;
            float pay[NCHP];
#pragma unroll
            for (int c4 = 0; c4 < NCHP / 4; ++c4) {
                const float4 t = s_pay[j][c4];
                pay[4 * c4] = t.x; pay[4 * c4 + 1] = t.y; pay[4 * c4 + 2] = t.z; pay[4 * c4 + 3] = t.w;
            }
#pragma unroll
            for (int k = 0; k < PXL; ++k) {
                if (!(mj & (1u << owned_quadrant<PXL>(part, k)))) continue;  // wave-uniform
                const float dx = a.x - pxf[k], dy = a.y - pyf[k];
                const float ex = blend_exponent(dx, dy, a.z, a.w, b.x, b.y);
                const float alpha = fminf(LSR_ALPHA_MAX, fast_exp2(ex));
                const bool live = !done[k] && (ex <= b.y) && (alpha >= LSR_ALPHA_MIN);
                const float test_T = __builtin_fmaf(-Tr[k], alpha, Tr[k]);
                const bool stop = live && (test_T < LSR_T_EPS);
                const bool blend = live && !stop;
                const float w = blend ? alpha * Tr[k] : 0.0f;
#pragma unroll
                for (int c = 0; c < NCHP; ++c) acc[k][c] = __builtin_fmaf(pay[c], w, acc[k][c]);
                accd[k] = __builtin_fmaf(b.z, w, accd[k]);
                Tr[k] = blend ? test_T : Tr[k];
                last[k] = blend ? pos : last[k];
                done[k] = done[k] || stop;
            }
        }
        __syncthreads();  // WAR on the LDS slice before the next batch is staged
    }

    const float *vw = p.views + (size_t)v * LSR_VIEW_FLOATS;
    const size_t HW = (size_t)p.H * p.W;
#pragma unroll
    for (int k = 0; k < PXL; ++k) {
        if (!inside[k]) continue;
        const size_t pix = (size_t)pyf[k] * p.W + (size_t)pxf[k];
        const size_t vp = (size_t)v * HW + pix;
        p.final_T[vp] = Tr[k];
        p.n_contrib[vp] = last[k];
        p.out_mask[vp] = 1.0f - Tr[k];
        p.out_depth[vp] = accd[k];
        if (p.has_color) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                p.out_color[((size_t)v * 3 + c) * HW + pix] = __builtin_fmaf(Tr[k], vw[37 + c], acc[k][c]);
        }
#pragma unroll
        for (int c = 0; c < NCHP; ++c)
            if (c >= coff && c - coff < p.C)
                p.out_feat[((size_t)v * p.C + (c - coff)) * HW + pix] = acc[k][c];
    }
}

static int pick_pxl(int nchp, int64_t tiles_total) {
    if (const char *e = getenv("LSR_PXL")) {
        const int x = atoi(e);
        if (x == 1 || x == 2 || x == 4) return (nchp > 12 && x == 4) ? 2 : x;
    }
    // enough independent waves to put >= 2 on every SIMD of the 256 CUs (1024 SIMDs)
    int pxl = tiles_total >= 2048 ? 4 : (tiles_total >= 1024 ? 2 : 1);
    if (nchp > 12 && pxl == 4) pxl = 2;  // keep accumulators within the VGPR budget
    return pxl;
}

hipError_t launch_render_forward(const lsr_dims &d, const lsr_inputs &in, const char *geom,
                                 const char *bin, int64_t num_pairs, char *img, const lsr_outputs &out,
                                 hipStream_t s) {
    const GeomLayout L = geom_layout(d);
    const ImgLayout I = img_layout(d);
    const BinLayout B = bin_layout(d, num_pairs, 0);
    RenderFwdParams p;
    p.H = d.height; p.W = d.width; p.gx = tiles_x(d); p.T = (int)num_tiles(d); p.G = d.num_gaussians;
    p.C = d.feat_channels; p.has_color = d.color_mode != LSR_COLOR_NONE; p.vs_feat = d.vs_feat;
    p.views = in.views;
    p.q0 = (const float4 *)(geom + L.q0); p.q1 = (const float4 *)(geom + L.q1);
    p.rgb = (const float4 *)(geom + L.rgb);
    p.features = in.features;
    p.tile_start = (const uint32_t *)(geom + L.tile_start);
    p.point_list = (const uint32_t *)(bin + B.point_list);
    p.out_color = out.color; p.out_feat = out.feature; p.out_mask = out.mask; p.out_depth = out.depth;
    p.final_T = (float *)(img + I.final_T); p.n_contrib = (uint32_t *)(img + I.n_contrib);
    const int nch = (p.has_color ? 3 : 0) + d.feat_channels;
    const int nchp = nch <= 4 ? 4 : (nch <= 8 ? 8 : (nch <= 12 ? 12 : 36));
    const int pxl = pick_pxl(nchp, (int64_t)p.T * d.num_views);
    dim3 grid(p.T * (4 / pxl), d.num_views);
#define LSR_RF(N, X) hipLaunchKernelGGL((k_render_fwd<N, X>), grid, dim3(LSR_WAVE), 0, s, p)
#define LSR_RF_N(N)                                  \
    do {                                             \
        if (pxl == 4) LSR_RF(N, 4);                  \
        else if (pxl == 2) LSR_RF(N, 2);             \
        else LSR_RF(N, 1);                           \
    } while (0)
    prof_begin(kStRenderFwd, s);
    if (nchp == 4) LSR_RF_N(4);
    else if (nchp == 8) LSR_RF_N(8);
    else if (nchp == 12) LSR_RF_N(12);
    else { if (pxl == 2) LSR_RF(36, 2); else LSR_RF(36, 1); }
#undef LSR_RF_N
#undef LSR_RF
    prof_end(kStRenderFwd, s);
    return hipGetLastError();
}

}  // namespace lsr
