// render_forward.hip — stage K6: front-to-back alpha compositing of the depth-sorted tile lists.
// Outputs colour (+ bg), features (bg 0), mask = 1 - T, depth = sum alpha T z, and keeps
// final_T / n_contrib for the backward pass.  Spec: SURVEY.md Appendix A.3 step 7 + A.4.
//
// Execution shape (CDNA4): persistent waves process work items — (view, tile, set of 8x8
// quadrants), sorted by list length (lsr_internal.h kItem*): the first one per wave by a static
// balanced assignment, further ones from a global queue.  A lane owns the same position
// in each of the 4 quadrants (4 pixels per lane); quadrants outside the item's set are simply
// never touched.  Per 64 staged list entries the wave walks, quadrant by quadrant, only the
// entries whose alpha >= 1/255 footprint can reach that quadrant (lsr_blend.h), UNR at a time.
// Measured on MI355X the kernel is bound by f32 VALU issue (~4 cycles per wave64 instruction),
// so the inner loop keeps per-pixel state decisions in scalar lane masks (SALU) rather than VGPR
// selects, and the quadrant culling + item splitting exist to cut / balance VALU work.
#include <stdio.h>

#include <vector>

#include "lsr_blend.h"

namespace lsr {

// One workgroup of 16 independent waves per CU (4 per SIMD; the waves never synchronise with each
// other).  Waves w, w+4, w+8, w+12 of a workgroup share a SIMD, which lets the kernel decide
// WHICH work items share a SIMD: items arrive sorted by cost, and SIMD-bin b takes items
// b, 2B-1-b, 2B+b, 4B-1-b, ... (B = number of bins) — pairing expensive with cheap tiles so every
// SIMD gets nearly the same total.  The kernel is VALU-throughput bound, so the makespan is the
// largest per-SIMD total; with one item per wave and no control over placement it was ~1.4x the
// mean (measured).
constexpr int kSimdBins = kWaveSlots / 4;
constexpr int kCUs = kSimdBins / 4;
__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving LDS
    // accesses across the staging / consuming phases and drains the wave's own LDS queue.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

typedef float float2_t __attribute__((ext_vector_type(2)));

struct RenderFwdParams {
    int H, W, gx, T, G, C, has_color;
    const uint32_t *items;        // work items, costliest first
    const uint32_t *header;       // geometry-workspace header (item count)
    uint32_t *queue;              // work-queue head (zeroed per forward)
    unsigned long long *trace;    // debug (LSR_TRACE): per item {start clk, end clk, hw id, evaluations << 32 | entries}
    const float *views;
    const float4 *rec;            // [V*G][rec_f4] screen-space records (lsr_internal.h)
    int rec_f4;
    const uint32_t *tile_start, *point_list;
    float *out_color, *out_feat, *out_mask, *out_depth;
    float *final_T;
    uint32_t *n_contrib;
};

// WPB = waves per workgroup: 16 (one workgroup per CU) unless the LDS slices do not fit, then 4
// workgroups of 4 waves stand in for it (same bins, placement then up to the dispatcher).
template <int NCHP, int UNR, int WPB>
__global__ void __launch_bounds__(LSR_WAVE * WPB)
k_render_fwd(RenderFwdParams p) {
    constexpr int PXL = 4;
    // Staged entries, one record per list entry: (x, y, a2, b2) (c2, log2 o, z, -) payload...
    // One LDS address per entry (a single VALU add, the parts at immediate offsets); the odd
    // float4 stride keeps the per-lane staging stores bank-conflict free.  Slot 64 is a null
    // record (alpha == 0) that pads partial groups.
    constexpr int kEnt = (2 + NCHP / 4) | 1;
    __shared__ float4 s_ent_all[WPB][LSR_WAVE + 1][kEnt];

    const int lane = threadIdx.x & (LSR_WAVE - 1);
    const int wid = threadIdx.x / LSR_WAVE;
    float4 (*s_ent)[kEnt] = s_ent_all[wid];
    if (lane == 0) {
        s_ent[LSR_WAVE][0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        s_ent[LSR_WAVE][1] = make_float4(0.0f, -INFINITY, 0.0f, 0.0f);  // log2(opacity) = -inf
#pragma unroll
        for (int c4 = 0; c4 < NCHP / 4; ++c4) s_ent[LSR_WAVE][2 + c4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const uint32_t num_items = p.header[kHdrNumItems];
    const int coff = p.has_color ? 3 : 0;
    const size_t HW = (size_t)p.H * p.W;

    const uint32_t vwave = (uint32_t)wid + (uint32_t)WPB * (blockIdx.x / (uint32_t)kCUs);   // 0..15
    const uint32_t bin = (blockIdx.x % (uint32_t)kCUs) * 4u + (vwave & 3u);
    // First item of every wave: static, folded (boustrophedon) over the cost-sorted list, so the 4
    // waves of a SIMD start with a balanced total.  Everything beyond the first kWaveSlots items
    // is pulled from a global queue (costliest first) as waves become free.
    const uint32_t j0 = vwave >> 2;
    bool first = true;
    for (;;) {
        uint32_t qi;
        if (first) {
            qi = (j0 & 1u) ? (j0 + 1u) * (uint32_t)kSimdBins - 1u - bin : j0 * (uint32_t)kSimdBins + bin;
            first = false;
            if (qi >= num_items) continue;   // fewer items than wave slots: go straight to the queue (empty)
        } else {
            if (num_items <= (uint32_t)kWaveSlots) break;
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(p.queue, 1u);
            qi = (uint32_t)kWaveSlots + __builtin_amdgcn_readfirstlane(t);
            if (qi >= num_items) break;
        }
        const unsigned long long t_begin = p.trace ? __builtin_readcyclecounter() : 0ull;
        uint32_t trace_evals = 0;   // debug (LSR_TRACE): (entry, quadrant) evaluations of this item
        const uint32_t item = p.items[qi];
        const uint32_t vt = item & kItemTileMask, own = item >> kItemOwnShift;
        const int tile = (int)(vt % (uint32_t)p.T), v = (int)(vt / (uint32_t)p.T);
        const int tx0 = (tile % p.gx) * LSR_TILE, ty0 = (tile / p.gx) * LSR_TILE;
        const size_t vG = (size_t)v * p.G;
        const uint32_t start = p.tile_start[vt], end = p.tile_start[vt + 1];

        float pxf[PXL], pyf[PXL], Tr[PXL], accd[PXL];
        float acc[PXL][NCHP];
        uint32_t stop_pos[PXL];
        // Per-pixel "finished" flags live in scalar registers as 64-bit lane masks, so the skip /
        // blend / stop decisions are SALU mask algebra instead of per-lane VALU selects.
        uint64_t done[PXL];
        bool inside[PXL];
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
            const int px = tx0 + 8 * (k & 1) + (lane & 7), py = ty0 + 8 * (k >> 1) + (lane >> 3);
            pxf[k] = (float)px; pyf[k] = (float)py;
            inside[k] = px < p.W && py < p.H && ((own >> k) & 1u);
            done[k] = __ballot(!inside[k]);
            Tr[k] = 1.0f; accd[k] = 0.0f; stop_pos[k] = 0;
#pragma unroll
            for (int c = 0; c < NCHP; ++c) acc[k][c] = 0.0f;
        }

        for (uint32_t base = start; base < end; base += LSR_WAVE) {
            uint64_t all_done = ~0ull;
#pragma unroll
            for (int k = 0; k < PXL; ++k) all_done &= done[k];
            if (all_done == ~0ull) break;

            // ---- stage up to 64 list entries (one per lane) ----
            const uint32_t e = base + lane;
            uint32_t m = 0;
            if (e < end) {
                const uint32_t g = p.point_list[e];
                const float4 *R = p.rec + (vG + g) * (size_t)p.rec_f4;
                const float4 a = R[0], b = R[1];  // (x,y,A,B) (C,o,z,-)
                m = quadrant_mask(a.x, a.y, a.z, a.w, b.x, b.y, (float)tx0, (float)ty0) & own;
                if (m) {
                    const FoldedConic f = fold_conic(a.z, a.w, b.x, b.y);
                    s_ent[lane][0] = make_float4(a.x, a.y, f.a2, f.b2);
                    s_ent[lane][1] = make_float4(f.c2, f.l2o, b.z, 0.0f);
#pragma unroll
                    for (int c4 = 0; c4 < NCHP / 4; ++c4) s_ent[lane][2 + c4] = R[2 + c4];  // payload, zero padded
                }
            }
            // per quadrant: which staged entries can touch it (wave-uniform 64-bit masks)
            uint64_t qbits[PXL];
#pragma unroll
            for (int k = 0; k < PXL; ++k) qbits[k] = __ballot((m >> k) & 1u);
            wave_lds_fence();  // staged records are visible to this wave's reads below
            if (p.trace) {
#pragma unroll
                for (int k = 0; k < PXL; ++k) trace_evals += (uint32_t)__builtin_popcountll(qbits[k]);
            }

            // Walk each quadrant's entries front to back, UNR at a time: the alpha evaluations of
            // the UNR entries are independent; only the short transmittance chain is serial.
#pragma unroll
            for (int k = 0; k < PXL; ++k) {
                uint64_t bits = qbits[k];
                while (bits) {
                    int jj[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        jj[u] = bits ? __builtin_ctzll(bits) : LSR_WAVE;  // slot 64 = null record
                        bits &= bits - 1;
                    }
                    float4 a[UNR], b[UNR];
                    float pay[UNR][NCHP];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        a[u] = s_ent[jj[u]][0]; b[u] = s_ent[jj[u]][1];
#pragma unroll
                        for (int c4 = 0; c4 < NCHP / 4; ++c4) {
                            const float4 t = s_ent[jj[u]][2 + c4];
                            pay[u][4 * c4] = t.x; pay[u][4 * c4 + 1] = t.y; pay[u][4 * c4 + 2] = t.z; pay[u][4 * c4 + 3] = t.w;
                        }
                    }
                    float alpha[UNR];
                    uint64_t ok[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const float dx = a[u].x - pxf[k], dy = a[u].y - pyf[k];
                        const float ex = blend_exponent(dx, dy, a[u].z, a[u].w, b[u].x, b[u].y);
                        alpha[u] = fminf(LSR_ALPHA_MAX, fast_exp2(ex));
                        // skip if power > 0 or alpha < 1/255 (NaN-safe: comparisons are "keep" tests)
                        ok[u] = __ballot(ex <= b[u].y) & __ballot(alpha[u] >= LSR_ALPHA_MIN);
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const float aT = alpha[u] * Tr[k];
                        const float tT = Tr[k] - aT;
                        const uint64_t live = ok[u] & ~done[k];
                        const uint64_t room = __ballot(tT >= LSR_T_EPS);
                        const uint64_t blend = live & room, stop = live & ~room;
                        const float w = __builtin_amdgcn_inverse_ballot_w64(blend) ? aT : 0.0f;
#pragma unroll
                        for (int c = 0; c < NCHP; ++c) acc[k][c] = __builtin_fmaf(pay[u][c], w, acc[k][c]);
                        // (accd, T) += w * (z, -1) as one packed FMA
                        const float2_t td = __builtin_elementwise_fma(float2_t{b[u].z, -1.0f}, float2_t{w, w}, float2_t{accd[k], Tr[k]});
                        accd[k] = td.x; Tr[k] = td.y;
                        if (stop) {  // rare, wave-uniform: a pixel's transmittance ran out here
                            const uint32_t pos = base - start + (uint32_t)jj[u] + 1u;
                            stop_pos[k] = __builtin_amdgcn_inverse_ballot_w64(stop) ? pos : stop_pos[k];
                            done[k] |= stop;
                        }
                    }
                }
            }
            wave_lds_fence();  // WAR on the LDS slice before the next batch is staged
        }

        const float *vw = p.views + (size_t)v * LSR_VIEW_FLOATS;
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
            if (!inside[k]) continue;
            const size_t pix = (size_t)pyf[k] * p.W + (size_t)pxf[k];
            const size_t vp = (size_t)v * HW + pix;
            p.final_T[vp] = Tr[k];
            // number of leading list entries the backward pass has to consider for this pixel: all
            // of them, or everything before the entry at which the transmittance test stopped it
            p.n_contrib[vp] = stop_pos[k] ? stop_pos[k] - 1u : end - start;
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
        if (p.trace && lane == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            p.trace[4 * (size_t)qi + 0] = t_begin;
            p.trace[4 * (size_t)qi + 1] = __builtin_readcyclecounter();
            p.trace[4 * (size_t)qi + 2] = ((unsigned long long)xcc << 32) | hwid;
            p.trace[4 * (size_t)qi + 3] = ((unsigned long long)trace_evals << 32) | (end - start);
        }
    }  // persistent item loop
}

hipError_t launch_render_forward(const lsr_dims &d, const lsr_inputs &in, const char *geom,
                                 const char *bin, int64_t num_pairs, char *img, const lsr_outputs &out,
                                 hipStream_t s) {
    const GeomLayout L = geom_layout(d);
    const ImgLayout I = img_layout(d);
    const BinLayout B = bin_layout(d, num_pairs, 0);
    RenderFwdParams p;
    p.H = d.height; p.W = d.width; p.gx = tiles_x(d); p.T = (int)num_tiles(d); p.G = d.num_gaussians;
    p.C = d.feat_channels; p.has_color = d.color_mode != LSR_COLOR_NONE;
    p.items = (const uint32_t *)(geom + L.tile_order);
    p.header = (const uint32_t *)(geom + L.header);
    p.queue = (uint32_t *)(geom + L.header) + kHdrQueueFwd;
    p.views = in.views;
    p.rec = (const float4 *)(geom + L.rec); p.rec_f4 = L.rec_floats / 4;
    p.tile_start = (const uint32_t *)(geom + L.tile_start);
    p.point_list = (const uint32_t *)(bin + B.point_list);
    p.out_color = out.color; p.out_feat = out.feature; p.out_mask = out.mask; p.out_depth = out.depth;
    p.final_T = (float *)(img + I.final_T); p.n_contrib = (uint32_t *)(img + I.n_contrib);
    const int nch = (p.has_color ? 3 : 0) + d.feat_channels;
    const int nchp = nch <= 4 ? 4 : (nch <= 8 ? 8 : (nch <= 12 ? 12 : 36));
    const int64_t max_items = 4 * (int64_t)p.T * d.num_views;

    p.trace = nullptr;
    const char *trace_path = getenv("LSR_TRACE");
    if (trace_path) {
        (void)hipMalloc((void **)&p.trace, (size_t)max_items * 32);
        (void)hipMemsetAsync(p.trace, 0, (size_t)max_items * 32, s);
    }
    prof_begin(kStRenderFwd, s);
#define LSR_RF(N, U, WPB) hipLaunchKernelGGL((k_render_fwd<N, U, WPB>), dim3(kCUs * (16 / WPB)), dim3(LSR_WAVE * WPB), 0, s, p)
    if (nchp == 4) LSR_RF(4, 2, 16);
    else if (nchp == 8) LSR_RF(8, 2, 16);
    else if (nchp == 12) LSR_RF(12, 2, 16);
    else LSR_RF(36, 1, 4);
#undef LSR_RF
    prof_end(kStRenderFwd, s);
    if (trace_path) {  // debug only: dump per-item timing of this launch
        std::vector<unsigned long long> host((size_t)max_items * 4);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(host.data(), p.trace, host.size() * 8, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(trace_path, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
        (void)hipFree(p.trace);
    }
    return hipGetLastError();
}

}  // namespace lsr
