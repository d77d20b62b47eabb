// render_backward.hip — stage K7: per-pixel gradient of the compositing, with the same sub-block
// decomposition, culling and work items as the forward (render_forward.hip, lsr_blend.h).
//
// The lists are walked FRONT TO BACK like the forward (the published kernel goes back to front and
// keeps, per pixel and channel, the colour accumulated behind the current entry: four operations
// per channel per evaluation).  Here the only per-pixel state is the transmittance T and ONE scalar
//   R_i = sum_{j>i} w_j (g . c_j)  [+ T_final (g . bg - g_mask)]  =  (g . C_rendered) - prefix_i,
// seeded from the rendered images of the matching forward; with d_i = g . c_i
//   dL/dalpha_i = T_i d_i - R_i / (1 - alpha_i),      R_i = R_{i-1} - w_i d_i,   T_{i+1} = T_i (1 - alpha_i)
// i.e. one dot product and one multiply (payload gradient) per channel.
// Where the forward order is NOT good enough (round 6): R_i is a difference of two float32 totals, and dL/dalpha_i divides it
// by (1 - alpha_i).  Behind an entry at the 0.99 alpha clamp R_i is 1 % of the total and the division multiplies its rounding
// by 100: one or two ulps of the total became 1e-4 of dL/dalpha (profiles/r05_fuzz_parity.md: seed 203 draw 1738).  The forward
// compositing kernels therefore flag every item in which they staged an entry of opacity >= kSteepOpacity (GeomLayout::
// item_flags), and flagged items are walked BACK TO FRONT with the published recurrence, which never forms
// the difference:  T_i = T_{i+1} / (1 - alpha_i),   dL/dalpha_i = T_i ((g . c_i) - S_i),   S_{i-1} = S_i + alpha_i ((g . c_i) - S_i),
// S_n = g . bg - g_mask  (S_i = R_i / T_{i+1}: the unattenuated composite of everything behind entry i).  Same items, lists,
// staging, reduction and flush; the batches and the per-sub-block lists are simply visited in reverse (`walk_item<REV>`).
// Unflagged items — every item of a scene with opacities below 0.75 — keep the forward-order arithmetic bit for bit.
// The items come in the order k_order_items (render_forward.hip) derived from the RECORD forward's per-item count of the
// iterations THIS kernel spends on them (header word kHdrOrder2Valid), else in the tile scan's order.
//
// Per (entry, pixel) the kernel recomputes alpha with the forward's exact arithmetic (exponent domain, units of
// 255 alpha: lsr_blend.h) and emits the
// MOMENTS of u = opacity * G * dL/dalpha over the entry's pixels:
//   m0 = sum u,   m1 = sum u (dx, dy),   m2 = sum u (dx^2, dx dy, dy^2),
// plus w * g per payload channel and w * g_depth.  The conic / opacity factors that turn moments
// into dL/d(x, y), dL/d(A, B, C), dL/d opacity are applied ONCE per (view, Gaussian) by
// k_preprocess_bwd (they are constant over the pixels), not per evaluation.
//
// Execution shape: the forward's work items (one 16x8 half of a tile per wave, walking that half's
// render list); one lane = two horizontally adjacent pixels, one 8-lane group = one 4x4 sub-block walking
// its own compacted list (as in the forward), so a wave instruction serves up to eight different entries.
// A lane first adds its two pixels' contributions, then the group's 8 lanes are summed by a transposed
// butterfly INSIDE the DPP row (row_half_mirror / quad_perm — no permlane, no LDS): afterwards lane j of
// the group holds the gradient record's slots (2j, 2j+1), and one 8-byte LDS read-modify-write per lane adds
// the eight groups' records into a per-batch table (row = staged entry).  When the batch is done the rows
// are flushed with coalesced global atomics, one 64-byte record per 16 lanes: one global record-add per
// entry of a half list (0.94 per (Gaussian, tile) pair).
// Launches that cannot fill the wave slots (a few views) split every list between 2 / 4 / 8 waves (SPLIT instance).
// Spec: SURVEY.md Appendix A.6.
#include <type_traits>

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
    const uint32_t *items;        // work items of the forward (view*T + tile | half << 28), costliest first
    const uint32_t *items2;       // the same items ordered by their true cost (k_sort_tiles; valid when header[kHdrOrder2Valid])
    const uint32_t *header;       // geometry-workspace header (item count)
    uint32_t *queue;              // work-queue head (zeroed with the gradient workspace)
    uint32_t *bin_queue;          // per-SIMD-bin queue heads (LSR_BWD_BINQ=1, experiment), or nullptr: one global queue
    int prio_pct;                 // issue priority by progress: percentage of the launch's mean tile list (0 = off; see the kernel)
    const uint32_t *item_flags;   // [2 V T] kItemFlagSteep per half-tile item, written by the forward compositing kernel (lsr_internal.h)
    int rev_mode;                 // which items are walked BACK TO FRONT: 0 none, 1 all, 2 the steep ones (and all when the forward left no flags)
    const float *views;
    const float4 *geo;            // [V*G][rec_f4] screen-space records (lsr_internal.h)
    int rec_f4;
    const uint32_t *tile_start, *half_list;
    IndexPacking ip;              // how the list entries carry the Gaussian index and the sub-block bits
    const float *final_T;
    const uint32_t *n_contrib;
    const float *g_color, *g_feat, *g_mask, *g_depth;  // dL/d outputs (any may be NULL)
    const float *f_color, *f_feat, *f_depth;           // images rendered by the matching forward
    int parts_log2;     // small view batches: every half-tile item is walked by 2^parts_log2 waves, each emitting the
                        // gradients of its share of the list's 64-entry batches (see the item loop)
    float *rec;         // [V*G][rec_floats] packed gradient records (zeroed by the caller)
    long long *rec_fixed;   // deterministic mode: the same records as 2^30 fixed point (zeroed by the caller), else NULL
    int rec_floats;
};

// ---- pair-transposed reduction inside an 8-lane group ---------------------------------------
// A lane group (8 lanes = half a DPP row) owns one 4x4 sub-block, two pixels per lane.  Every lane holds
// up to 16 values (its two pixels already added), the gradient record's slots 0..15 as eight PAIRS
// P_k = (slot 2k, slot 2k+1).  The sums over the group's 8 lanes are formed by a transposed butterfly — every
// step halves the number of pairs a lane still carries — after which lane j of the group holds the group
// total of pair P_j, i.e. two adjacent record slots: one 8-byte LDS read-modify-write per lane updates the
// whole 64-byte record row.
//   step A  row_half_mirror     (partner 7-j: other bit 2)   lanes with bit 2 keep P_{k+4}, the others P_k
//   step B  quad_perm [2,3,0,1] (partner j^2)                bit 1 selects k+2 / k
//   step C  quad_perm [1,0,3,2] (partner j^1)                bit 0 selects k+1 / k
// LIVE = compile-time mask of pairs that can be non-zero: a pair with one live member costs one DPP add per
// register, a dead pair nothing (lanes that end up with a dead pair hold junk and do not write).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {   // x + partner(x)
    return x + f_from_u((unsigned)__builtin_amdgcn_update_dpp(0, (int)u_from_f(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL, bool LO, bool HI>
__device__ __forceinline__ float2_b pair_step(bool bit, float2_b lo, float2_b hi) {
    if (LO && HI) {
        const float kx = bit ? hi.x : lo.x, ky = bit ? hi.y : lo.y, sx = bit ? lo.x : hi.x, sy = bit ? lo.y : hi.y;
        return float2_b{kx + f_from_u((unsigned)__builtin_amdgcn_update_dpp(0, (int)u_from_f(sx), CTRL, 0xf, 0xf, false)),
                        ky + f_from_u((unsigned)__builtin_amdgcn_update_dpp(0, (int)u_from_f(sy), CTRL, 0xf, 0xf, false))};
    }
    if (LO) return float2_b{dpp_add<CTRL>(lo.x), dpp_add<CTRL>(lo.y)};
    if (HI) return float2_b{dpp_add<CTRL>(hi.x), dpp_add<CTRL>(hi.y)};
    return float2_b{0.0f, 0.0f};
}
template <uint32_t LIVE>
__device__ __forceinline__ float2_b group_reduce_pairs(float2_b p0, float2_b p1, float2_b p2, float2_b p3,
                                                       float2_b p4, float2_b p5, float2_b p6, float2_b p7, int l8) {
    const bool b2 = l8 & 4, b1 = l8 & 2, b0 = l8 & 1;
#define LV(i) (((LIVE) >> (i)) & 1u)
    const float2_b q0 = pair_step<0x141, LV(0), LV(4)>(b2, p0, p4), q1 = pair_step<0x141, LV(1), LV(5)>(b2, p1, p5);
    const float2_b q2 = pair_step<0x141, LV(2), LV(6)>(b2, p2, p6), q3 = pair_step<0x141, LV(3), LV(7)>(b2, p3, p7);
    constexpr uint32_t LA = (LIVE | (LIVE >> 4)) & 0xFu;
#define LVA(i) ((LA >> (i)) & 1u)
    const float2_b r0 = pair_step<0x4E, LVA(0), LVA(2)>(b1, q0, q2), r1 = pair_step<0x4E, LVA(1), LVA(3)>(b1, q1, q3);
    constexpr uint32_t LB = (LA | (LA >> 2)) & 0x3u;
    return pair_step<0xB1, (LB & 1u) != 0, (LB & 2u) != 0>(b0, r0, r1);
#undef LV
#undef LVA
}

__device__ __forceinline__ void atomic_add_f32(float *addr, float v) { unsafeAtomicAdd(addr, v); }

// WPB = waves per workgroup, WGS = workgroups per compute unit: as many resident waves as the per-wave LDS slice
// and the register budget allow (the kernel is bound by instruction issue and the latencies of its dependent
// chain: DESIGN.md).
// SPLIT: the list-splitting instance for launches that cannot fill the wave slots (see the item loop); the plain instance
// carries none of its tests.
template <int NCHP, bool DEPTH_GRAD, int WPB, int WGS, bool SPLIT>
__global__ void __launch_bounds__(LSR_WAVE * WPB)
k_render_bwd(RenderBwdParams p) {
    // float4 per staged entry: odd (conflict-free staging stores) except for the 8-channel payload, whose
    // 16 floats are stored at a 64-byte stride (2-way conflicts in the staging stores only) so that the
    // slice fits 16 waves per CU
    constexpr int kEnt = NCHP == 8 ? 4 : ((2 + NCHP / 4) | 1);
    constexpr int RF = NCHP <= 8 ? 16 : (NCHP <= 12 ? 32 : 64);       // == rec_floats (lsr_internal.h)
    constexpr int NGRP = RF / 16;                                     // 16-value (8-pair) reduction passes per evaluation
    // words per row of the LDS gradient table: for the 16-float record without depth gradient only the live slots
    // (0..5 and 8..8+NCHP-1; the pair (6, 7) and the unused payload pairs have no column), the full record otherwise
    constexpr bool kPackRow = NCHP <= 8 && !DEPTH_GRAD;
    constexpr int RT = kPackRow ? 6 + NCHP : RF;
    constexpr int kListRow = LSR_WAVE + 2;    // u16 per list row: 33 words, so the eight lane groups' reads of list[b][i] hit eight banks
    struct Lds {
        float4 ent[WPB][LSR_WAVE + 1][kEnt];   // (x, y, a2, c2) (b2, log2 o, z, list position) payload...; slot 64 = null record
        float acc[WPB][LSR_WAVE + 1][RT];      // this batch's gradient records, row = staged entry (row 64: dump row of the null record)
        uint32_t gid[WPB][LSR_WAVE];           // Gaussian index of the staged entry
        uint16_t list[WPB][8][kListRow];       // per sub-block: staging slots of the entries that can reach it, in list order
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
            s_ent[LSR_WAVE][1] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));  // position beyond every list: never valid
#pragma unroll
            for (int c4 = 0; c4 < NCHP / 4; ++c4) s_ent[LSR_WAVE][2 + c4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (lane < 16) s_list[lane >> 1][LSR_WAVE + (lane & 1)] = (uint16_t)LSR_WAVE;   // the rows' pad words
        float *A = &s_acc[0][0];
        for (int i = lane; i < (LSR_WAVE + 1) * RT; i += LSR_WAVE) A[i] = 0.0f;
    }
    const int parts_log2 = SPLIT ? p.parts_log2 : 0;
    const uint32_t num_items = p.header[kHdrNumItems] << parts_log2;
    const int coff = p.has_color ? 3 : 0;
    const size_t HW = (size_t)p.H * p.W;
    // lane group -> sub-block (gcol, grow) of the half; lane -> its two pixels (lx, ly), (lx + 1, ly) (as in the forward)
    const int grp = lane >> 3, gcol = grp & 3, grow = grp >> 2, l8 = lane & 7;
    const int lx = 2 * (lane & 1), ly = (lane >> 1) & 3;
    // this lane's pair of record slots (2 l8, 2 l8 + 1) -> columns of the table row (pair 3 has none when packed)
    const int tcol = kPackRow ? (l8 < 4 ? 2 * l8 : 2 * l8 - 2) : 2 * l8;
    const int l16 = lane & 15, fgrp = lane >> 4;           // flush: one 64-byte record per 16 lanes
    const int fcol = kPackRow ? (l16 < 8 ? l16 : l16 - 2) : l16;

    const uint32_t simd_bins = (uint32_t)p.num_cus * 4u, slots = (uint32_t)p.num_cus * (uint32_t)(WPB * WGS);
    const uint32_t vwave = (uint32_t)wid + (uint32_t)WPB * (blockIdx.x / (uint32_t)p.num_cus);   // 0 .. WPB*WGS-1
    const uint32_t bin = (blockIdx.x % (uint32_t)p.num_cus) * 4u + (vwave & 3u);
    const uint32_t j0 = vwave >> 2;
    // Issue priority by progress (round 5).  A wave walks its list at HIGH priority (s_setprio 3) until prio_target entries are
    // left of it — prio_pct % of the launch's mean tile list (pair count / tiles) — and finishes at priority 0.  Waves close
    // to the end of their item yield the issue slots of their SIMD to the ones that still have far to go, so the four waves
    // of a SIMD finish their items closer together and the launch's drain — SIMDs down to one or two waves, which cannot fill
    // the issue slots — gets shorter.  Measured (profiles/r05_ab_knobs.md section 12; 18 % is the flat optimum between 15 and
    // 20): 16 views x 300 k 0.60 -> 0.574 ms, 8 / 32 views -6 %, 10^6 Gaussians -7 %, the list-splitting instance -8 % (6 views)
    // and -13 % (configs[3]); the 8-channel unsplit instance (configs[4]) LOSES 1.5-10 % at every setting and lists shorter than
    // a few batches (100 k Gaussians: +4 %) cannot resolve the switch point, so the launcher leaves those at 0 and targets
    // below three batches switch it off here.  (A static priority by item cost did nothing: section 6; the
    // inverse order — the end of an item at high priority — costs 0-10 %; only the order matters: 3 -> 1 and 1 -> 0 measure the same
    // as 3 -> 0.  The forward, six waves per SIMD, does not gain.)  Results do not depend on it.
    uint32_t prio_target = p.prio_pct ? (uint32_t)(((uint64_t)p.header[kHdrPairs] * (uint32_t)p.prio_pct) / (50ull * (uint64_t)max(p.header[kHdrNumItems], 1u))) : 0u;
    if (prio_target < 3u * LSR_WAVE) prio_target = 0u;
    const bool flags_valid = p.rev_mode == 2 && p.header[kHdrFlagsValid] != 0u;
    const uint32_t *items = p.header[kHdrOrder2Valid] ? p.items2 : p.items;   // (uniform: a scalar load)
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
            if (p.bin_queue) {
                // per-SIMD-bin lists as in k_render_fwd (LSR_BWD_BINQ=1; measured no gain here: 0.570 -> 0.568 ms at 16 views x 300 k,
                // +2.5 % on opaque scenes, +3 % at 10^6 Gaussians: off by default)
                if (lane == 0) t = atomicAdd(p.bin_queue + bin, 1u);
                const uint32_t k = (uint32_t)(WPB * WGS) / 4u + __builtin_amdgcn_readfirstlane(t);
                qi = (k & 1u) ? (k + 1u) * simd_bins - 1u - bin : k * simd_bins + bin;
            } else {
                if (lane == 0) t = atomicAdd(p.queue, 1u);
                qi = slots + __builtin_amdgcn_readfirstlane(t);
            }
            if (qi >= num_items) break;
        }
        qi = __builtin_amdgcn_readfirstlane(qi);
        const uint32_t item = items[qi >> parts_log2];
        const uint32_t part = qi & ((1u << parts_log2) - 1u);
        const uint32_t vt = item & kItemTileMask, half = item >> kItemHalfShift;
        const int tile = (int)(vt % (uint32_t)p.T), v = (int)(vt / (uint32_t)p.T);
        const int tx0 = (tile % p.gx) * LSR_TILE, ty0 = (tile / p.gx) * LSR_TILE + 8 * (int)half;
        const size_t vG = (size_t)v * p.G;
        const uint32_t tstart = p.tile_start[vt], tn = p.tile_start[vt + 1] - tstart;
        const uint32_t *hlist = p.half_list + 2 * (size_t)tstart + (size_t)half * tn;
        // back to front for items in which an alpha can come close to the clamp (the header comment says why)
        const bool rev = p.rev_mode == 1 || (p.rev_mode == 2 && (!flags_valid || (p.item_flags[2 * (size_t)vt + half] & kItemFlagSteep)));

        auto walk_item = [&](auto rev_tag) __attribute__((always_inline)) {
        constexpr bool REV = decltype(rev_tag)::value;
        // per-pixel state of the lane's two pixels as register pairs (pixel 0, pixel 1); R2: forward order — what is left behind
        // the current entry (R_i); reverse order — S_i / 255, the unattenuated composite of everything behind it
        float2_b pxx, T2, R2, ddep2;
        float pyf;
        float2_b dpix[NCHP];          // dL/d output channel c at the two pixels
        uint32_t last0, last1, maxlast;
        {
            const int px = tx0 + 4 * gcol + lx, py = ty0 + 4 * grow + ly;
            pxx = float2_b{(float)px, (float)(px + 1)};
            pyf = (float)py;
            const bool in0 = px < p.W && py < p.H, in1 = px + 1 < p.W && py < p.H;
            // every per-pixel load is issued UNCONDITIONALLY at a clamped pixel (all of a row's loads in
            // flight together) and masked afterwards: loads under the per-lane `inside` test were waited
            // for one by one — ~30 serial round trips at the head of every item
            const int pyc = min(py, p.H - 1);
            const size_t pix0 = (size_t)pyc * p.W + min(px, p.W - 1), pix1 = (size_t)pyc * p.W + min(px + 1, p.W - 1);
            const size_t vp0 = (size_t)v * HW + pix0, vp1 = (size_t)v * HW + pix1;
            const float Tf0 = p.final_T[vp0], Tf1 = p.final_T[vp1];
            const uint32_t nc0 = p.n_contrib[vp0], nc1 = p.n_contrib[vp1];
            float2_b gl[NCHP], fl[NCHP];
#pragma unroll
            for (int c = 0; c < NCHP; ++c) gl[c] = fl[c] = float2_b{0.0f, 0.0f};
            if (p.has_color && p.g_color) {          // kernel-argument uniform
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const size_t o = ((size_t)v * 3 + c) * HW;
                    gl[c] = float2_b{p.g_color[o + pix0], p.g_color[o + pix1]};
                    if (!REV) fl[c] = float2_b{p.f_color[o + pix0], p.f_color[o + pix1]};
                }
            }
            if (p.g_feat) {
#pragma unroll
                for (int c = 0; c < NCHP; ++c)
                    if (c >= coff && c - coff < p.C) {   // uniform
                        const size_t o = ((size_t)v * p.C + (c - coff)) * HW;
                        gl[c] = float2_b{p.g_feat[o + pix0], p.g_feat[o + pix1]};
                        if (!REV) fl[c] = float2_b{p.f_feat[o + pix0], p.f_feat[o + pix1]};
                    }
            }
            const float2_b gmask = p.g_mask ? float2_b{p.g_mask[vp0], p.g_mask[vp1]} : float2_b{0.0f, 0.0f};
            const float2_b gdep = DEPTH_GRAD ? float2_b{p.g_depth[vp0], p.g_depth[vp1]} : float2_b{0.0f, 0.0f};
            const float2_b fdep = (DEPTH_GRAD && !REV) ? float2_b{p.f_depth[vp0], p.f_depth[vp1]} : float2_b{0.0f, 0.0f};
            const float2_b inm = float2_b{in0 ? 1.0f : 0.0f, in1 ? 1.0f : 0.0f};
            const float2_b Tfin = float2_b{in0 ? Tf0 : 1.0f, in1 ? Tf1 : 1.0f};
            T2 = float2_b{1.0f, 1.0f};
            last0 = in0 ? nc0 : 0u; last1 = in1 ? nc1 : 0u;
            maxlast = max(last0, last1);
            // R_0 = g . (rendered - T_final * bg)  +  T_final * (g . bg - g_mask)  =  g . rendered - T_final * g_mask
            float2_b r0 = float2_b{0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < NCHP; ++c) {
                const float2_b g = gl[c] * inm;
                r0 = __builtin_elementwise_fma(fl[c], g, r0);     // channels that are not rendered hold zeros
                dpix[c] = g * kInv255;                            // the loop works in units of 255 alpha (lsr_blend.h)
            }
            r0 = __builtin_elementwise_fma(-Tfin, gmask * inm, r0);  // mask = 1 - T_final
            if (DEPTH_GRAD) r0 = __builtin_elementwise_fma(fdep, gdep * inm, r0);
            ddep2 = DEPTH_GRAD ? gdep * inm * kInv255 : float2_b{0.0f, 0.0f};
            R2 = r0;
            if (REV) {
                // back to front: start behind the pixel's last entry with T = T_final and S / 255 = (g . bg - g_mask) / 255
                // (the background reaches the colour channels only; features and depth have none)
                T2 = Tfin;
                float2_b s0 = -(gmask * inm) * kInv255;
                if (p.has_color && p.g_color) {
                    typedef const float __attribute__((address_space(4))) *kfloat_ptr;
                    const kfloat_ptr vw = (kfloat_ptr)(p.views + (size_t)v * LSR_VIEW_FLOATS);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { const float bgc = vw[37 + c]; s0 = __builtin_elementwise_fma(dpix[c], float2_b{bgc, bgc}, s0); }
                }
                R2 = s0;
            }
        }
        // wave-uniform upper bound of the list entries any pixel has to consider
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxlast = max(maxlast, (uint32_t)__shfl_xor((int)maxlast, off));
        maxlast = __builtin_amdgcn_readfirstlane(maxlast);
        // List splitting for launches that cannot fill the wave slots (a few views): part k of 2^parts_log2 EMITS the
        // gradients of batches [b0, b1) of the list only.  The per-pixel state in front of batch b0 — T and R — is a function
        // of the entries before it, so the wave first walks batches [0, b0) updating just T and R (a quarter of an emitting
        // step's instructions: no reciprocal, no moments, no cross-lane reduction, no table, no flush), with the operations of
        // the emitting walk — every entry's gradient is the unsplit kernel's bit for bit.
        // (Back to front the roles swap: the part walks batches nb - 1 ... b0 and emits those below b1 — the state behind
        // batch b1 - 1 is a function of the entries behind it.)
        uint32_t emit_from = 0, walk_end = maxlast;
        const uint32_t nb = (maxlast + LSR_WAVE - 1) / LSR_WAVE;
        uint32_t rev_stop = 0u, rev_emit_below = nb;     // REV: last batch walked, batches [rev_stop, rev_emit_below) emit
        if (SPLIT) {
            const uint32_t b0 = (part * nb) >> parts_log2, b1 = ((part + 1u) * nb) >> parts_log2;
            if (b0 == b1) return;          // a list shorter than the number of parts: this part has no batch
            emit_from = b0 * LSR_WAVE;
            walk_end = min(maxlast, b1 * LSR_WAVE);
            rev_stop = b0; rev_emit_below = b1;
        }
        // walk order: batch k of the walk starts at list position wbase(k)
        const uint32_t nwalk = REV ? nb - rev_stop : (walk_end + LSR_WAVE - 1) / LSR_WAVE;
        const uint32_t wfirst = REV ? (nb - 1u) * LSR_WAVE : 0u;
        const uint32_t wstep = REV ? 0u - (uint32_t)LSR_WAVE : (uint32_t)LSR_WAVE;   // (batches below zero wrap and are clamped by load_ent)

        // software-pipelined staging as in the forward: records of batch b+1 and list entries of
        // batch b+2 are in flight while batch b is processed
        // (unconditional loads at clamped positions, see the forward: a load under a per-lane condition is
        // waited for at the join)
        struct StageRec { float4 a, b, pay[NCHP / 4]; uint32_t w; };
        const uint32_t lastrel = maxlast - 1u;   // only used when maxlast > 0
        auto load_ent = [&](uint32_t rel) -> uint32_t { return hlist[min(rel, lastrel)]; };
        auto load_rec = [&](uint32_t w) {
            StageRec r;
            r.w = w;
            const float4 *R = p.geo + (vG + (w & p.ip.index_mask)) * (size_t)p.rec_f4;
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
            w_ahead = load_ent(wfirst + lane);
            nxt = load_rec(w_ahead);
            w_ahead = load_ent(wfirst + wstep + lane);
        }

        // (the priority switch counts in walked entries: burn_end = entries walked at high priority)
        uint32_t burn_end = 0u;
        const uint32_t wlen = REV ? nwalk * LSR_WAVE : walk_end;
        if (prio_target && wlen > prio_target) { burn_end = wlen - prio_target; __builtin_amdgcn_s_setprio(3); }
        uint32_t cbase = wfirst;
        for (uint32_t wk = 0; wk < nwalk; ++wk, cbase += wstep) {
            const bool emit = !SPLIT || (REV ? cbase < rev_emit_below * LSR_WAVE : cbase >= emit_from);   // wave-uniform
            if (burn_end && wk * LSR_WAVE >= burn_end) { __builtin_amdgcn_s_setprio(0); burn_end = 0u; }
            const StageRec cur = nxt;
            nxt = load_rec(w_ahead);
            w_ahead = load_ent(cbase + 2u * wstep + lane);
            // ---- stage up to 64 list entries (one per lane) ----
#pragma unroll
            for (int b = 0; b < 8; ++b) s_list[b][lane] = (uint16_t)LSR_WAVE;   // every list slot starts as the null record's slot
            const uint32_t rel = cbase + lane;  // 0-based position in the half's list
            const uint32_t m = rel < maxlast ? (((cur.w >> kListBitsShift) | p.ip.all_bits) & 0xFFu) : 0u;
            if (m) {
                const float4 a = cur.a, b = cur.b;
                const FoldedConic f = fold_conic(a.z, a.w, b.x, b.y);   // same folding as the forward
                s_ent[lane][0] = make_float4(a.x, a.y, f.a2, f.c2);
                s_ent[lane][1] = make_float4(f.b2, f.l2o, b.z, __uint_as_float(rel + 1u));
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4) s_ent[lane][2 + c4] = cur.pay[c4];
                s_gid[lane] = cur.w & p.ip.index_mask;
            }
            const uint64_t staged = __ballot(m != 0);
            // compaction: per sub-block, the staged entries that can reach it, in list order.  An entry
            // that sits at the SAME position in the lists of several sub-blocks will be processed by several
            // lane groups in the same iteration; their updates of the entry's table row are ordered by a rank
            // (number of lower sub-blocks holding the entry at that position), computed here once per
            // staged entry and stored with the list element:   list element = staging slot | rank << 8.
            uint32_t nk = 0, at[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                at[b] = 0xFFFFu;           // 0xFFFF: not in this list
                const uint64_t bal = __ballot((m >> b) & 1u);
                const uint32_t nb_b = (uint32_t)__builtin_popcountll(bal);
                nk = max(nk, nb_b);
                if (__builtin_amdgcn_inverse_ballot_w64(bal)) {
                    at[b] = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (REV) at[b] = nb_b - 1u - at[b];     // the sub-block's list in reverse order
                }
            }
            nk = __builtin_amdgcn_readfirstlane(nk);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (at[j] == 0xFFFFu) continue;
                uint32_t rank = 0;
                if (emit) {
#pragma unroll
                    for (int jj = 0; jj < j; ++jj) rank += (uint32_t)(at[jj] == at[j]);
                }
                s_list[j][at[j]] = (uint16_t)((uint32_t)lane | (rank << 8));
            }
            wave_lds_fence_bwd();

            const uint16_t *lp = &s_list[grp][0];
            for (uint32_t i = 0; i < nk; ++i) {
                const uint32_t lel = lp[i];
                const uint32_t slot = lel & 0xFFu, rank = lel >> 8;
                const float4_b *E = (const float4_b *)&s_ent[slot][0];
                float2_b *row = (float2_b *)&s_acc[slot][tcol];
                float2_b acc_old[NGRP];                          // this lane's pair(s) of the entry's table row, read early
#pragma unroll
                for (int gi = 0; gi < NGRP; ++gi) acc_old[gi] = float2_b{0.0f, 0.0f};
                if (emit) {
#pragma unroll
                    for (int gi = 0; gi < NGRP; ++gi) acc_old[gi] = row[8 * gi];
                }
                float4_b a = E[0], b = E[1], t4[NCHP / 4];
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4) t4[c4] = E[2 + c4];
                // Several lane groups can be at the SAME staged entry in this iteration: their
                // read-modify-writes of the entry's table row are ordered by the rank stored with the list
                // element.  Rank 0 updates with the early-read pair; rank r > 0 re-reads the row after rank
                // r-1 has written (a wave's LDS operations execute in order).
                auto accumulate = [&](float2_b *dst, float2_b old, float2_b tot, bool live) {
                    if (live && rank == 0u) *dst = old + tot;
                    uint64_t later = __ballot(rank != 0u);
                    for (uint32_t r = 1; later; ++r) {      // wave-uniform, usually no or one round
                        wave_lds_fence_bwd();
                        if (live && rank == r) *dst = *dst + tot;
                        later &= ~__ballot(rank == r);
                    }
                };
                float pay[NCHP];
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4) {
                    pay[4 * c4] = t4[c4].x; pay[4 * c4 + 1] = t4[c4].y; pay[4 * c4 + 2] = t4[c4].z; pay[4 * c4 + 3] = t4[c4].w;
                }
                // the forward's exponent e' and keep test, operation for operation (both pixels packed; lsr_blend.h)
                const float2_b d2 = float2_b{a.x, a.x} - pxx;
                const float dy = a.y - pyf;
                const float tt = b.x * dy;
                const float ss = __builtin_fmaf(a.w * dy, dy, b.y);
                const float2_b p1 = __builtin_elementwise_fma(float2_b{a.z, a.z}, d2, float2_b{tt, tt});
                const float2_b ex = __builtin_elementwise_fma(p1, d2, float2_b{ss, ss});
                const float er0 = fast_exp2(ex.x), er1 = fast_exp2(ex.y);                     // 255 opacity exp(power)
                const float ec0 = fminf(kAlphaMax255, er0), ec1 = fminf(kAlphaMax255, er1);   // 255 alpha
                const uint32_t pos = __float_as_uint(b.w), lim = __float_as_uint(b.y);
                const bool valid0 = (pos <= last0) & (__float_as_uint(ex.x) <= lim);
                const bool valid1 = (pos <= last1) & (__float_as_uint(ex.y) <= lim);
                const float2_b al = float2_b{valid0 ? ec0 : 0.0f, valid1 ? ec1 : 0.0f};      // 255 alpha (0: skipped)
                const float2_b av = float2_b{valid0 ? er0 : 0.0f, valid1 ? er1 : 0.0f};      // 255 opacity exp(power)
                const float2_b om = float2_b{255.0f, 255.0f} - al;                           // 255 (1 - alpha)
                const float2_b rcp1m = float2_b{__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                float2_b Tk = T2;                                // transmittance in front of this entry
                if (REV) {
                    // T_i = T_{i+1} / (1 - alpha_i): the hardware reciprocal (1 ulp) refined by one Newton step, so that the
                    // quotient does not drift over a long list; entries the pixel skips leave T alone
                    const float2_b e1 = __builtin_elementwise_fma(-om, rcp1m, float2_b{1.0f, 1.0f});
                    const float2_b q = __builtin_elementwise_fma(e1, rcp1m, rcp1m) * 255.0f;
                    const float2_b Tq = T2 * q;
                    Tk = float2_b{valid0 ? Tq.x : T2.x, valid1 ? Tq.y : T2.y};
                }
                const float2_b w = al * Tk;                      // 255 alpha T
                float2_b dsum = float2_b{pay[0], pay[0]} * dpix[0];   // (g . c_i) / 255 at the two pixels
#pragma unroll
                for (int c = 1; c < NCHP; ++c) dsum = __builtin_elementwise_fma(float2_b{pay[c], pay[c]}, dpix[c], dsum);
                if (DEPTH_GRAD) dsum = __builtin_elementwise_fma(float2_b{b.z, b.z}, ddep2, dsum);
                float2_b dL_dalpha;
                if (REV) {
                    // dL/dalpha / 255 = T_i ((g . c_i) - S_i) / 255;   S_{i-1} = S_i + alpha_i ((g . c_i) - S_i)
                    // (the published recurrence, SURVEY A.6: nothing is ever formed as a difference of two totals)
                    const float2_b diff = dsum - R2;
                    dL_dalpha = Tk * diff;
                    R2 = __builtin_elementwise_fma(al * kInv255, diff, R2);
                    T2 = Tk;
                } else {
                    R2 = __builtin_elementwise_fma(-w, dsum, R2);    // what is left behind this entry
                    T2 = __builtin_elementwise_fma(w, float2_b{-kInv255, -kInv255}, Tk);
                    // dL/dalpha / 255 = T (g . c) / 255 - R / (255 (1 - alpha))
                    dL_dalpha = __builtin_elementwise_fma(Tk, dsum, -(R2 * rcp1m));
                }
                if (!emit) continue;                             // another part's batch: only the state moves on
                const float2_b u = av * dL_dalpha;               // opacity exp(power) dL/dalpha: straight through the 0.99 clamp (A.6)
                // moments of u over the lane's two pixels (same dy): sum u (dx, dy), u (dx^2, dx dy, dy^2), u
                const float2_b ud = u * d2;
                const float2_b udd = ud * d2;
                const float usum = u.x + u.y;
                const float mx = ud.x + ud.y, my = usum * dy;
                const float mxx = udd.x + udd.y, mxy = mx * dy, myy = my * dy;
                // dL/d payload channel c and depth, the two pixels added
                float gp[NCHP];
#pragma unroll
                for (int c = 0; c < NCHP; ++c) { const float2_b t = dpix[c] * w; gp[c] = t.x + t.y; }
                float gz = 0.0f;
                if (DEPTH_GRAD) { const float2_b t = ddep2 * w; gz = t.x + t.y; }
                // ---- sum over the sub-block's 8 lanes; lane j of the group ends up with the pair of slots (2j, 2j+1) ----
                {
                    // pairs: P0 (mx, my) P1 (mxx, mxy) P2 (myy, u) P3 (gz, -) P4..P7 payload channels 0..7
                    constexpr uint32_t LIVE = 0x7u | (DEPTH_GRAD ? 0x8u : 0u) | (((1u << ((NCHP < 8 ? NCHP : 8) / 2)) - 1u) << 4);
#define GP2(k) float2_b{(k) < NCHP ? gp[(k) % NCHP] : 0.0f, (k) + 1 < NCHP ? gp[((k) + 1) % NCHP] : 0.0f}
                    const float2_b tot = group_reduce_pairs<LIVE>(float2_b{mx, my}, float2_b{mxx, mxy}, float2_b{myy, usum}, float2_b{gz, 0.0f},
                                                                  GP2(0), GP2(2), GP2(4), GP2(6), l8);
                    // plain read-modify-write: only this wave touches its table and a wave's LDS operations
                    // execute in order
                    accumulate(row, acc_old[0], tot, LIVE >> l8 & 1u);
                }
#pragma unroll
                for (int gi = 1; gi < NGRP; ++gi) {   // payload channels 16 gi - 8 .. 16 gi + 7
                    const float2_b tot = group_reduce_pairs<0xFFu>(GP2(16 * gi - 8), GP2(16 * gi - 6), GP2(16 * gi - 4), GP2(16 * gi - 2),
                                                                   GP2(16 * gi), GP2(16 * gi + 2), GP2(16 * gi + 4), GP2(16 * gi + 6), l8);
                    accumulate(row + 8 * gi, acc_old[gi], tot, true);
                }
#undef GP2
            }
            wave_lds_fence_bwd();
            // ---- flush: one global record-add per staged entry (every list entry reaches one of the half's sub-blocks) ----
#pragma unroll 1
            for (int e0 = 0; e0 < LSR_WAVE; e0 += 4) {
                if (!emit || !((staged >> e0) & 0xFull)) continue;   // wave-uniform
                const int e = e0 + fgrp;
                const bool hit = (staged >> e) & 1ull;
                const uint32_t g = s_gid[e];
#pragma unroll
                for (int gi = 0; gi < NGRP; ++gi) {
                    const bool mine = !kPackRow || l16 < 6 || (l16 >= 8 && l16 < 8 + NCHP);   // slots without a column in a packed row
                    const float val = mine ? s_acc[e][16 * gi + fcol] : 0.0f;
                    if (mine) s_acc[e][16 * gi + fcol] = 0.0f;
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
        if (burn_end) __builtin_amdgcn_s_setprio(0);
        };   // walk_item
        if (rev) walk_item(std::true_type{});
        else walk_item(std::false_type{});
    }  // item loop
}

__global__ void __launch_bounds__(256) k_fixed_to_float(const long long *__restrict__ in, float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (float)((double)in[i] * (1.0 / kFixedPointScale));
}

// Zeroes what the compositing backward accumulates into.  (Round 5 measured clearing the records of the VISIBLE slots only
// — radii > 0, 68 % of them — with a kernel that reads the radius and stores under the test: the forward + backward step
// got 0.01-0.03 ms SLOWER than with this plain streaming clear of everything; what stayed is that the per-Gaussian backward
// kernels no longer READ the records of culled slots: GradLayout's zero line.)
hipError_t launch_clear_grad(const lsr_dims &d, const int32_t *radii, char *grad, hipStream_t s) {
    (void)radii;
    return launch_clear(grad, grad_layout(d).total, s);
}

template <int NCHP, bool DG, int WPB, int WGS>
static void launch_variant(RenderBwdParams p, uint64_t items, hipStream_t s) {
    // list splitting: as many waves per half-tile item (a power of two, at most 8) as the launch's wave slots hold.
    // LSR_BWD_PARTS = log2 of the number of parts forces it (0: never split).
    const uint64_t slots = (uint64_t)p.num_cus * WPB * WGS;
    const int forced = env_int("LSR_BWD_PARTS", -1);
    int pl = 0;
    if (forced >= 0) pl = forced > 3 ? 3 : forced;
    else while (pl < 3 && (items << (pl + 1)) <= slots) ++pl;
    p.parts_log2 = pl;
    // issue priority by progress (see the kernel): on for the instances it was measured to help, LSR_BWD_PRIO_PCT >= 0 forces a value
    const int prio_knob = env_int("LSR_BWD_PRIO_PCT", -1);
    p.prio_pct = prio_knob >= 0 ? prio_knob : ((!DG && (NCHP == 4 || (NCHP == 8 && pl > 0))) ? 18 : 0);
    if (pl) hipLaunchKernelGGL((k_render_bwd<NCHP, DG, WPB, WGS, true>), dim3(p.num_cus * WGS), dim3(LSR_WAVE * WPB), 0, s, p);
    else hipLaunchKernelGGL((k_render_bwd<NCHP, DG, WPB, WGS, false>), dim3(p.num_cus * WGS), dim3(LSR_WAVE * WPB), 0, s, p);
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
    p.items2 = (const uint32_t *)(geom + L.tile_order2);
    p.header = (const uint32_t *)(geom + L.header);
    p.views = in.views;
    p.geo = (const float4 *)(geom + L.rec); p.rec_f4 = L.rec_floats / 4;
    p.tile_start = (const uint32_t *)(geom + L.tile_start);
    p.half_list = (const uint32_t *)(bin + B.half_list);
    p.ip = index_packing(d);
    p.final_T = (const float *)(img + I.final_T); p.n_contrib = (const uint32_t *)(img + I.n_contrib);
    p.g_color = gout.color; p.g_feat = gout.feature; p.g_mask = gout.mask; p.g_depth = gout.depth;
    p.f_color = fwd.color; p.f_feat = fwd.feature; p.f_depth = fwd.depth;
    p.rec = (float *)(grad + R.rec); p.rec_floats = R.rec_floats; p.parts_log2 = 0;
    const bool det = deterministic_backward();
    p.rec_fixed = det ? (long long *)(grad + R.fixed) : nullptr;
    p.queue = (uint32_t *)(grad + R.fixed - 512);   // the zeroed slack behind the float records
    p.bin_queue = (env_int("LSR_BWD_BINQ", 0) && 4 * p.num_cus <= (int)(kBinQueueBytes / 4)) ? (uint32_t *)(grad + R.fixed - 512 - kBinQueueBytes) : nullptr;
    p.item_flags = (const uint32_t *)(geom + L.item_flags);
    {   // LSR_BWD_REV: 0 never walk back to front, 1 always, 2 (default) the items the forward flagged as steep
        const int rm = env_int("LSR_BWD_REV", 2);
        p.rev_mode = rm < 0 || rm > 2 ? 2 : rm;
    }
    (void)gin;
    const int nch = (p.has_color ? 3 : 0) + d.feat_channels;
    const int nchp = nch <= 4 ? 4 : (nch <= 8 ? 8 : (nch <= 12 ? 12 : 36));
    const bool dg = gout.depth != nullptr;
    prof_begin(kStRenderBwd, s);
    const uint64_t items = 2ull * (uint64_t)d.num_views * (uint64_t)p.T;   // upper bound of the header's item count
#define LSR_RB(N, WPB, WGS)                                      \
    do {                                                         \
        if (dg) launch_variant<N, true, WPB, WGS>(p, items, s);  \
        else launch_variant<N, false, WPB, WGS>(p, items, s);    \
    } while (0)
    // LSR_BWD_VARIANT (development knob, read once) selects the alternatives measured in DESIGN.md
    const int variant = env_int("LSR_BWD_VARIANT", 0);
    if (nchp == 4 && !dg) { if (variant == 1) launch_variant<4, false, 10, 2>(p, items, s); else if (variant == 2) launch_variant<4, false, 12, 1>(p, items, s); else launch_variant<4, false, 16, 1>(p, items, s); }
    else if (nchp == 4) LSR_RB(4, 16, 1);
    // packed table rows: 16 waves per CU fit with 64-byte staged entries.  (Round 4: this instance shows 34 % LDS bank-conflict
    // cycles — eight lane groups reading eight 64-byte records start in one of only four bank groups — but the odd
    // 80-byte stride costs a wave: 15 waves 0.969-0.977 ms, 14 waves 1.02 against 0.945-0.954 at configs[4].)
    else if (nchp == 8 && !dg) launch_variant<8, false, 16, 1>(p, items, s);
    else if (nchp == 8) launch_variant<8, true, 12, 1>(p, items, s);
    else if (nchp == 12) LSR_RB(12, 8, 1);
    else LSR_RB(36, 4, 1);
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
