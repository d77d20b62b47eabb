// render_backward.hip — stage K7: per-pixel gradient of the compositing, with the same quadrant
// decomposition and culling as the forward (lsr_blend.h).
//
// The lists are walked FRONT TO BACK like the forward (the published kernel goes back to front and
// keeps, per pixel and channel, the colour accumulated behind the current entry: four operations
// per channel per evaluation).  Here the only per-pixel state is the transmittance T and ONE scalar
//   R_i = sum_{j>i} w_j (g . c_j)  [+ T_final (g . bg - g_mask)]  =  (g . C_rendered) - prefix_i,
// seeded from the rendered images of the matching forward; with d_i = g . c_i
//   dL/dalpha_i = T_i d_i - R_i / (1 - alpha_i),      R_i = R_{i-1} - w_i d_i,   T_{i+1} = T_i (1 - alpha_i)
// i.e. one dot product and one FMA (payload gradient) per channel.  Half the per-pixel registers,
// so 8-channel payloads also run 4 pixels per lane.  Cancellation in R only matters when the
// remaining contribution is already ~1e-7 of the pixel's total — far below the 1e-4 tolerance.
// Per (entry, pixel): recompute alpha with the forward's exact arithmetic and emit
//   dL/d(x,y)_pixel, dL/d(A,B,C) conic, dL/d opacity, dL/d payload (rgb / features), dL/d z.
// The per-lane arithmetic is branch-free: an invalid (pixel, entry) simply has alpha = G = 0, which
// leaves T, R and every gradient sum unchanged.
// Per (entry, wave): contributions of the wave's 64*PXL pixels are summed with the transposed
// butterfly of lsr_blend.h (permlane swaps + DPP, no LDS traffic), which leaves the total of
// gradient slot s in lane 4s; ONE atomic instruction then adds the whole 64-byte gradient record
// of the (view, Gaussian) (lsr_internal.h GradLayout).
//
// Scheduling as in the forward: one 16-wave workgroup per CU; a unit of work is (tile, part) where
// `part` selects the wave's PXL of the tile's 4 quadrants; units are ordered by list length, the
// first unit of every wave is assigned statically (folded over the sorted list so the 4 waves of a
// SIMD get a balanced total), the rest comes from a global queue.
// Spec: SURVEY.md Appendix A.6.
#include "lsr_blend.h"

namespace lsr {


__device__ __forceinline__ void wave_lds_fence_bwd() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct RenderBwdParams {
    int H, W, gx, T, G, C, has_color;
    int num_cus;                  // workgroups of 16 waves (one per compute unit)
    uint32_t num_units;           // V * T * (4 / PXL)
    const uint32_t *tile_lpt;     // (view*T + tile), costliest first
    uint32_t *queue;              // work-queue head (zeroed with the gradient workspace)
    const float *views;
    const float4 *geo;            // [V*G][rec_f4] screen-space records (lsr_internal.h)
    int rec_f4;
    const uint32_t *tile_start, *point_list;
    const float *final_T;
    const uint32_t *n_contrib;
    const float *g_color, *g_feat, *g_mask, *g_depth;  // dL/d outputs (any may be NULL)
    const float *f_color, *f_feat, *f_depth;           // images rendered by the matching forward
    float *rec;         // [V*G][rec_floats] packed gradient records (zeroed by the caller)
    int rec_floats;
};

__device__ __forceinline__ void atomic_add_f32(float *addr, float v) { unsafeAtomicAdd(addr, v); }

template <int NCHP, int PXL, bool DEPTH_GRAD, int WPB>
__global__ void __launch_bounds__(LSR_WAVE * WPB)
k_render_bwd(RenderBwdParams p) {
    constexpr int NW = 4 / PXL;
    extern __shared__ float4 s_dyn[];
    const int lane = threadIdx.x & (LSR_WAVE - 1);
    const int wid = threadIdx.x / LSR_WAVE;
    // per-wave LDS slice: [64] (x,y,A,B) | [64] (C,o,z,gid) | [64][NCHP/4] payload
    constexpr int SLICE = LSR_WAVE * (2 + NCHP / 4);
    float4 *s_q0 = s_dyn + (size_t)wid * SLICE, *s_q1 = s_q0 + LSR_WAVE;
    float4 (*s_pay)[NCHP / 4] = (float4 (*)[NCHP / 4])(s_q1 + LSR_WAVE);
    const int coff = p.has_color ? 3 : 0;
    const size_t HW = (size_t)p.H * p.W;

    const uint32_t simd_bins = (uint32_t)p.num_cus * 4u, slots = simd_bins * 4u;
    const uint32_t vwave = (uint32_t)wid + (uint32_t)WPB * (blockIdx.x / (uint32_t)p.num_cus);   // 0..15
    const uint32_t bin = (blockIdx.x % (uint32_t)p.num_cus) * 4u + (vwave & 3u);
    const uint32_t j0 = vwave >> 2;
    bool first = true;
    for (;;) {
        uint32_t ui;
        if (first) {
            ui = (j0 & 1u) ? (j0 + 1u) * simd_bins - 1u - bin : j0 * simd_bins + bin;
            first = false;
            if (ui >= p.num_units) continue;
        } else {
            if (p.num_units <= slots) break;
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(p.queue, 1u);
            ui = slots + __builtin_amdgcn_readfirstlane(t);
            if (ui >= p.num_units) break;
        }
        const uint32_t vt = p.tile_lpt[ui / NW];
        const int part = (int)(ui % NW);
        const int tile = (int)(vt % (uint32_t)p.T), v = (int)(vt / (uint32_t)p.T);
        const int tx0 = (tile % p.gx) * LSR_TILE, ty0 = (tile / p.gx) * LSR_TILE;
        const size_t vG = (size_t)v * p.G;
        const uint32_t start = p.tile_start[vt];
        const uint32_t own = owned_mask<PXL>(part);

        float pxf[PXL], pyf[PXL], Tr[PXL], Rr[PXL], ddep[PXL];
        float dpix[PXL][NCHP];
        uint32_t last[PXL];
        uint32_t maxlast = 0;
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
            const int q = owned_quadrant<PXL>(part, k);
            const int px = tx0 + 8 * (q & 1) + (lane & 7), py = ty0 + 8 * (q >> 1) + (lane >> 3);
            pxf[k] = (float)px; pyf[k] = (float)py;
            const bool inside = px < p.W && py < p.H;
            const size_t pix = (size_t)py * p.W + px, vp = (size_t)v * HW + pix;
            const float Tfin = inside ? p.final_T[vp] : 1.0f;
            Tr[k] = 1.0f;
            last[k] = inside ? p.n_contrib[vp] : 0u;
            maxlast = max(maxlast, last[k]);
            // R_0 = g . (rendered - T_final * bg)  +  T_final * (g . bg - g_mask)  =  g . rendered - T_final * g_mask
            float r0 = 0.0f;
#pragma unroll
            for (int c = 0; c < NCHP; ++c) dpix[k][c] = 0.0f;
            if (inside) {
                if (p.has_color && p.g_color) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        dpix[k][c] = p.g_color[((size_t)v * 3 + c) * HW + pix];
                        r0 = __builtin_fmaf(p.f_color[((size_t)v * 3 + c) * HW + pix], dpix[k][c], r0);
                    }
                }
                if (p.g_feat) {
#pragma unroll
                    for (int c = 0; c < NCHP; ++c)
                        if (c >= coff && c - coff < p.C) {
                            dpix[k][c] = p.g_feat[((size_t)v * p.C + (c - coff)) * HW + pix];
                            r0 = __builtin_fmaf(p.f_feat[((size_t)v * p.C + (c - coff)) * HW + pix], dpix[k][c], r0);
                        }
                }
                if (p.g_mask) r0 = __builtin_fmaf(-Tfin, p.g_mask[vp], r0);  // mask = 1 - T_final
            }
            ddep[k] = (DEPTH_GRAD && inside) ? p.g_depth[vp] : 0.0f;
            if (DEPTH_GRAD && inside) r0 = __builtin_fmaf(p.f_depth[vp], ddep[k], r0);
            Rr[k] = r0;
        }
        // wave-uniform upper bound of the entries any owned pixel has to consider
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxlast = max(maxlast, (uint32_t)__shfl_xor((int)maxlast, off));
        maxlast = __builtin_amdgcn_readfirstlane(maxlast);

        for (int chunk = 0; (uint32_t)chunk * LSR_WAVE < maxlast; ++chunk) {
            const uint32_t rel = (uint32_t)chunk * LSR_WAVE + lane;  // 0-based position in the list
            uint32_t m = 0;
            if (rel < maxlast) {
                const uint32_t g = p.point_list[start + rel];
                const float4 *R = p.geo + (vG + g) * (size_t)p.rec_f4;
                const float4 a = R[0], b = R[1];
                m = quadrant_mask(a.x, a.y, a.z, a.w, b.x, b.y, (float)tx0, (float)ty0) & own;
                if (m) {
                    s_q0[lane] = a;
                    s_q1[lane] = make_float4(b.x, b.y, b.z, __uint_as_float(g));
#pragma unroll
                    for (int c4 = 0; c4 < NCHP / 4; ++c4) s_pay[lane][c4] = R[2 + c4];
                }
            }
            uint64_t qbits[PXL];
#pragma unroll
            for (int k = 0; k < PXL; ++k) qbits[k] = __ballot((m >> owned_quadrant<PXL>(part, k)) & 1u);
            uint64_t todo = __ballot(m != 0);
            wave_lds_fence_bwd();

            while (todo) {
                const int j = __builtin_ctzll(todo);  // front to back
                todo &= todo - 1;
                const float4 a = s_q0[j], b = s_q1[j];
                const uint32_t pos = (uint32_t)chunk * LSR_WAVE + (uint32_t)j + 1u;
                float pay[NCHP];
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4) {
                    const float4 t = s_pay[j][c4];
                    pay[4 * c4] = t.x; pay[4 * c4 + 1] = t.y; pay[4 * c4 + 2] = t.z; pay[4 * c4 + 3] = t.w;
                }
                const FoldedConic f2 = fold_conic(a.z, a.w, b.x, b.y);   // same folding as the forward
                const float inv_o = __builtin_amdgcn_rcpf(b.y);
                // sums over this wave's pixels (sign / 0.5 factors applied once, after the pixel loop)
                float sx = 0.0f, sy = 0.0f, sA = 0.0f, sB = 0.0f, sC = 0.0f, go = 0.0f, gz = 0.0f;
                float gpay[NCHP];
#pragma unroll
                for (int c = 0; c < NCHP; ++c) gpay[c] = 0.0f;
                uint64_t any_valid = 0;
#pragma unroll
                for (int k = 0; k < PXL; ++k) {
                    if (!(qbits[k] >> j & 1ull)) continue;  // wave-uniform
                    const float dx = a.x - pxf[k], dy = a.y - pyf[k];
                    const float ex = blend_exponent(dx, dy, f2.a2, f2.b2, f2.c2, f2.l2o);
                    const float araw = fast_exp2(ex);
                    const float aclamp = fminf(LSR_ALPHA_MAX, araw);
                    const uint64_t valid = __ballot(pos <= last[k]) & __ballot(ex <= f2.l2o) & __ballot(aclamp >= LSR_ALPHA_MIN);
                    any_valid |= valid;
                    const bool vb = __builtin_amdgcn_inverse_ballot_w64(valid);
                    const float alpha = vb ? aclamp : 0.0f;
                    const float Gv = vb ? araw * inv_o : 0.0f;      // exp(power)
                    const float om = 1.0f - alpha;
                    const float rcp1m = __builtin_amdgcn_rcpf(om);
                    const float Tk = Tr[k];          // transmittance in front of this entry
                    const float w = alpha * Tk;
                    float dsum = 0.0f;               // g . c_i
#pragma unroll
                    for (int c = 0; c < NCHP; ++c) {
                        dsum = __builtin_fmaf(pay[c], dpix[k][c], dsum);
                        gpay[c] = __builtin_fmaf(w, dpix[k][c], gpay[c]);
                    }
                    if (DEPTH_GRAD) {
                        dsum = __builtin_fmaf(b.z, ddep[k], dsum);
                        gz = __builtin_fmaf(w, ddep[k], gz);
                    }
                    Rr[k] = __builtin_fmaf(-w, dsum, Rr[k]);      // what is left behind this entry
                    Tr[k] = Tk * om;
                    const float dL_dalpha = __builtin_fmaf(Tk, dsum, -Rr[k] * rcp1m);
                    const float dL_dG = b.y * dL_dalpha;    // straight through the 0.99 clamp (A.6)
                    const float tA = Gv * dx * dL_dG, tC = Gv * dy * dL_dG;
                    sx = __builtin_fmaf(tA, a.z, __builtin_fmaf(tC, a.w, sx));     // -(dL/dx)
                    sy = __builtin_fmaf(tC, b.x, __builtin_fmaf(tA, a.w, sy));     // -(dL/dy)
                    sA = __builtin_fmaf(tA, dx, sA);                               // -2 dL/dA
                    sB = __builtin_fmaf(tA, dy, sB);                               // -  dL/dB
                    sC = __builtin_fmaf(tC, dy, sC);                               // -2 dL/dC
                    go = __builtin_fmaf(Gv, dL_dalpha, go);
                }
                if (!any_valid) continue;
                // ---- wave-wide sums, 16 record slots at a time; lane 4s ends up with slot s ----
                float *rec = p.rec + (size_t)(vG + __float_as_uint(b.w)) * p.rec_floats;
                {
                    constexpr uint32_t LIVE = 0x3Fu | (DEPTH_GRAD ? 0x40u : 0u) | (((1u << (NCHP < 8 ? NCHP : 8)) - 1u) << 8);
                    const float v16[16] = {-sx, -sy, -0.5f * sA, -sB, -0.5f * sC, go, DEPTH_GRAD ? gz : 0.0f, 0.0f,
                                           gpay[0], gpay[1], gpay[2], gpay[3],
                                           NCHP > 4 ? gpay[4 % NCHP] : 0.0f, NCHP > 4 ? gpay[5 % NCHP] : 0.0f,
                                           NCHP > 4 ? gpay[6 % NCHP] : 0.0f, NCHP > 4 ? gpay[7 % NCHP] : 0.0f};
                    const float tot = wave_reduce16_transposed<LIVE>(v16, lane);
                    const int slot = lane >> 2;
                    if ((lane & 3) == 0 && (LIVE >> slot & 1u) && (slot < 8 || slot - 8 < coff + p.C))
                        atomic_add_f32(rec + slot, tot);
                }
                if (NCHP > 8) {
#pragma unroll
                    for (int grp = 1; grp * 16 - 8 < NCHP; ++grp) {   // payload channels 16*grp-8 .. 16*grp+7
                        float v16[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) v16[i] = (16 * grp - 8 + i) < NCHP ? gpay[(16 * grp - 8 + i) % NCHP] : 0.0f;
                        const float tot = wave_reduce16_transposed<0xFFFFu>(v16, lane);
                        const int ch = 16 * grp - 8 + (lane >> 2);
                        if ((lane & 3) == 0 && ch < coff + p.C) atomic_add_f32(rec + 8 + ch, tot);
                    }
                }
            }
            wave_lds_fence_bwd();
        }
    }  // unit loop
}

static int pick_pxl_bwd(int nchp, int64_t tiles_total) {
    {
        const int x = env_int("LSR_PXL_BWD", 0);   // development knob, latched once
        if (x == 1 || x == 2 || x == 4) return nchp > 12 ? 1 : ((nchp > 8 && x == 4) ? 2 : x);
    }
    // aim for >= 4 waves on each of the 1024 SIMDs (the kernel is VALU bound and needs them)
    int pxl = tiles_total >= 4096 ? 4 : (tiles_total >= 2048 ? 2 : 1);
    if (nchp > 8 && pxl == 4) pxl = 2;  // per-pixel upstream gradients: PXL * NCHP registers
    if (nchp > 12) pxl = 1;
    return pxl;
}

template <int NCHP, int PXL, bool DG, int WPB>
static void launch_variant(const RenderBwdParams &p, hipStream_t s) {
    const size_t shm = (size_t)WPB * LSR_WAVE * (2 + NCHP / 4) * sizeof(float4);
    // function attributes are per device: set on every launch that needs it (a process may drive several GPUs)
    if (shm > 65536)
        (void)hipFuncSetAttribute((const void *)k_render_bwd<NCHP, PXL, DG, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL((k_render_bwd<NCHP, PXL, DG, WPB>), dim3(p.num_cus * (16 / WPB)), dim3(LSR_WAVE * WPB), shm, s, p);
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
    p.views = in.views;
    p.geo = (const float4 *)(geom + L.rec); p.rec_f4 = L.rec_floats / 4;
    p.tile_start = (const uint32_t *)(geom + L.tile_start);
    p.tile_lpt = (const uint32_t *)(geom + L.tile_lpt);
    p.point_list = (const uint32_t *)(bin + B.point_list);
    p.final_T = (const float *)(img + I.final_T); p.n_contrib = (const uint32_t *)(img + I.n_contrib);
    p.g_color = gout.color; p.g_feat = gout.feature; p.g_mask = gout.mask; p.g_depth = gout.depth;
    p.f_color = fwd.color; p.f_feat = fwd.feature; p.f_depth = fwd.depth;
    p.rec = (float *)(grad + R.rec); p.rec_floats = R.rec_floats;
    p.queue = (uint32_t *)(grad + R.total - 256);   // inside the zeroed tail of the gradient workspace
    (void)gin;
    const int nch = (p.has_color ? 3 : 0) + d.feat_channels;
    const int nchp = nch <= 4 ? 4 : (nch <= 8 ? 8 : (nch <= 12 ? 12 : 36));
    const int pxl = pick_pxl_bwd(nchp, (int64_t)p.T * d.num_views);
    p.num_units = (uint32_t)((int64_t)p.T * d.num_views * (4 / pxl));
    const bool dg = gout.depth != nullptr;
    prof_begin(kStRenderBwd, s);
#define LSR_RB(N, X, W)                                             \
    do {                                                            \
        if (dg) launch_variant<N, X, true, W>(p, s);                \
        else launch_variant<N, X, false, W>(p, s);                  \
    } while (0)
    if (nchp == 4) { if (pxl == 4) LSR_RB(4, 4, 16); else if (pxl == 2) LSR_RB(4, 2, 16); else LSR_RB(4, 1, 16); }
    else if (nchp == 8) { if (pxl == 4) LSR_RB(8, 4, 16); else if (pxl == 2) LSR_RB(8, 2, 16); else LSR_RB(8, 1, 16); }
    else if (nchp == 12) { if (pxl == 2) LSR_RB(12, 2, 16); else LSR_RB(12, 1, 16); }
    else LSR_RB(36, 1, 4);
#undef LSR_RB
    prof_end(kStRenderBwd, s);
    return hipGetLastError();
}

}  // namespace lsr
