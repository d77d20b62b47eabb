// render_forward.hip — stage K6: front-to-back alpha compositing of the depth-sorted tile lists.
// Outputs colour (+ bg), features (bg 0), mask = 1 - T, depth = sum alpha T z, and keeps
// final_T / n_contrib for the backward pass.  Spec: SURVEY.md Appendix A.3 step 7 + A.4.
//
// Execution shape (CDNA4): persistent waves process work items — one HALF (16 x 8 pixels) of one (view,
// tile) each (lsr_internal.h kItem*), sorted by list length: the first one per wave by a static
// balanced assignment, further ones from a global queue.
//
// A wave walks ITS half's render list (written by k_sort_tiles: the depth-ordered entries whose
// alpha >= 1/255 footprint reaches the half, each with an 8-bit mask of the half's 4x4-pixel
// SUB-BLOCKS it can reach).  One lane = TWO horizontally adjacent pixels, one 8-lane group = one
// sub-block.  The 64 list entries staged per batch are compacted into 8 per-sub-block lists in LDS
// (ballot + mbcnt, entries keep their depth order); every lane group then walks ITS sub-block's
// list, so one wave instruction evaluates up to eight different entries, each only on a sub-block it
// can reach (lossless).
// Why two pixels per lane (round 3): in-situ ablations showed that these kernels are bound
// by TOTAL instruction issue — a scalar mask operation or an LDS read costs a SIMD about as much as a
// vector operation (tools/microbench/valu_rates.hip), v_pk_* about 1.4 plain operations.  With two
// pixels per lane the list read, the three 16-byte record reads, the loop control and the row terms of
// the exponent (dy, b2 dy, c2 dy^2 + log2 o) are paid once per two pixel evaluations, and the two
// pixels' independent transmittance chains interleave.
// History: round 1 evaluated wave-uniform entries on whole 8x8 quadrants (77 pixel evaluations
// per (Gaussian, tile) pair); round 2 introduced the sub-block lists but staged the CANONICAL tile list
// per item and derived the masks per staged entry (39 evaluations; a quarter of the staged pairs reached
// no pixel).
// Per-pixel blend / skip / stop decisions are scalar lane-mask algebra, one mask set per pixel of the lane.
// Launches that cannot fill the wave slots get finer items (k_render_fwd_small): ROW items (two waves per half tile, one
// pixel per lane) and, for the longest lists while wave slots are left, SUB-BLOCK items (a wave per sub-block, four list
// entries per step).
#include <stdio.h>

#include <vector>

#include "lsr_blend.h"

namespace lsr {

__device__ __forceinline__ void wave_lds_fence() {
    // The LDS slice is private to the wave and a wave's LDS operations execute in order, so no
    // hardware wait is needed between staging and consuming; this is a compiler-only barrier (a
    // memory fence here would also drain the global loads that prefetch the next batch).
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

typedef float float2_t __attribute__((ext_vector_type(2)));

// acc += splat(src.lo or src.hi) * ww as ONE packed instruction: the broadcast of one half of a register pair is the
// instruction's op_sel modifier.  (Left to the compiler, every other broadcast of a HIGH half cost a v_mov first:
// four per iteration of the compositing loop, 1.1e7 of 9.7e7 VALU instructions per 16-view launch.)
// The operands are never results of the immediately preceding instruction (they come from LDS reads, selects and
// the previous iteration), which is what the packed-result forwarding hazard of gfx940+ would need a wait state for.

// packed multiply / multiply-add as inline asm: the compiler's hazard recognizer puts a wait state behind every
// packed-f32 instruction whose result the next instruction reads (it keys on a modifier bit that is set by default);
// tools/microbench/pk_hazard.hip shows the hardware needs none.
__device__ __forceinline__ float2_t pk_mul_nw(float2_t a, float2_t b) {
    float2_t d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a * splat(b.hi) + c
__device__ __forceinline__ float2_t pk_fma_bhi_nw(float2_t a, float2_t b, float2_t c) {
    float2_t d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
template <bool HI>
__device__ __forceinline__ void pk_fma_splat(float2_t &acc, float2_t src, float2_t ww) {
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(src), "v"(ww));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(src), "v"(ww));
}

struct RenderFwdParams {
    int H, W, gx, T, G, C, has_color;
    int num_cus;                  // compute units
    int waves_per_cu;             // resident compositing waves per CU the launch provides
    const uint32_t *items;        // work items, costliest first
    uint32_t num_items;           // items of this launch (2 per (view, tile))
    uint32_t quad_items;          // k_render_fwd_small: leading (costliest) items rendered as sub-block items
    uint32_t all_live;            // LSR_FWD_LIVE=0 (development knob): finished sub-blocks keep their lists (the round-5 behaviour)
    uint32_t *queue;              // work-queue head (zeroed per forward)
    uint32_t *bin_queue;          // per-SIMD-bin queue heads (zeroed per forward), or nullptr: one global queue
    const uint32_t *header;       // geometry-workspace header (pair count: the launch's mean list length)
    int prio_pct;                 // issue priority by progress as in k_render_bwd (percentage of the mean tile list; 0 = off)
    unsigned long long *trace;    // debug builds (LSR_ENABLE_TRACE): per item {start clk, end clk, hw id, iterations << 32 | entries}
    const float *views;
    const float4 *rec;            // [V*G][rec_f4] screen-space records (lsr_internal.h)
    int rec_f4;
    const uint32_t *tile_start, *half_count, *half_list;
    uint32_t *half_list_rw;       // RECORD instances: the same lists, written back with the sub-block bits refined (below)
    uint32_t *item_cost;          // RECORD instances: [2 V T] lock-step iterations the backward will spend on the item (for k_order_items), or nullptr
    uint32_t *item_flags;         // [2 V T] per half-tile item: kItemFlagSteep (lsr_internal.h), for the compositing backward; cleared per forward
    uint32_t *header_rw;          // kHdrFlagsValid: set by the instances that fill in item_flags
    IndexPacking ip;              // how the list entries carry the Gaussian index and the sub-block bits
    float *out_color, *out_feat, *out_mask, *out_depth;
    float *final_T;
    uint32_t *n_contrib;
};

// Workgroups of WPB independent waves (the waves never synchronise with each other); the launch
// provides waves_per_cu / WPB workgroups per CU.  Waves w, w+4, w+8, ... of a workgroup share a SIMD,
// which lets the kernel decide WHICH work items share a SIMD: items arrive sorted by cost, and SIMD-bin b
// takes items b, 2B-1-b, 2B+b, 4B-1-b, ... (B = number of bins) — pairing expensive with cheap items
// so every SIMD starts with nearly the same total; everything beyond the first item per wave comes from
// the queue.
//
// RECORD (round 5; forwards that a backward will follow: lsr_dims::forward_flags): while it composites, the kernel notes for
// every staged entry on WHICH of the half's eight sub-blocks it contributed to at least one pixel (kept by the alpha and
// the transmittance test) and, when a batch is done, writes those bits back over the entry's sub-block bits in the half
// list.  The bits k_sort_tiles put there describe the footprint's bounding box (39 pixel evaluations per pair, 34 for the
// exact ellipse, fewer where pixels have run out of transmittance); the compositing BACKWARD walks the same lists and is
// 2.5 x as expensive per evaluation — with the refined bits it evaluates an entry only where it has a gradient at all.
// Cost here: one LDS atomic-or per loop iteration on the word of the staged record that otherwise holds the constant
// -1/255 (kept in a register pair instead).  Lossless for the backward by construction: an (entry, sub-block) without a
// contributing pixel has alpha T = 0 everywhere on the sub-block, i.e. no gradient (the stopping entry of a pixel is not
// blended and lies beyond the pixel's n_contrib).
// EMU (timing experiments only, LSR_FWD_VARIANT 5 / 6; WRONG images): every item is walked by EMU waves, wave k taking the
// k-th share of its batches from T = 1 — what a split along the list through the linearity of the composite would
// execute, without its partial writes and combine pass (profiles/r06_ab_knobs.md: the bound of that design).
template <int NCHP, int WPB, bool RECORD = false, int EMU = 0>
__global__ void __launch_bounds__(LSR_WAVE * WPB, WPB == 14 ? 7 : ((RECORD && WPB == 12) ? 6 : 1))   // (2 x 14 waves: 72 registers, seven waves per SIMD; the RECORD instance of 2 x 12 must stay within six)
k_render_fwd(RenderFwdParams p) {
    // Staged entries, one record per list entry: (x, y, a2, c2) (b2, log2(255 o), z / 255, -1 / 255) payload / 255 ...
    // (lsr_blend.h: the loop works in units of 255 alpha).  The odd float4 stride keeps the per-lane staging
    // stores bank-conflict free.  Slot 64 is a null record (x = NaN: never kept) that pads the sub-block lists.
    constexpr int kEnt = (2 + NCHP / 4) | 1;
    // One shared object, entries first: the byte offsets kept in the lists are then plain LDS
    // addresses (base 0 folds into the ds_read immediate, no address arithmetic per entry).
    struct Lds {
        float4 ent[WPB][LSR_WAVE + 1][kEnt];
        // list[b][i] = offset of the i-th staged entry that can reach sub-block b.  Rows are 65 words apart:
        // the eight lane groups read list[b][i] for eight b at the same i — with a 64-word stride that would be
        // one LDS bank for all of them (the 11 % bank-conflict cycles of round 2's profile)
        uint32_t list[WPB][8][LSR_WAVE + 1];
    };
    __shared__ Lds s_lds;

    const int lane = threadIdx.x & (LSR_WAVE - 1);
    // wave-uniform values are made provably so (readfirstlane): loop control, list bounds and the
    // per-pixel lane masks then live in SGPRs and the item loop is scalar control flow
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / LSR_WAVE);
    float4 (*s_ent)[kEnt] = s_lds.ent[wid];
    uint32_t (*s_list)[LSR_WAVE + 1] = s_lds.list[wid];
    const char *ent_base = (const char *)&s_lds.ent[0][0][0];
    const uint32_t wave_off = (uint32_t)(wid * (LSR_WAVE + 1) * kEnt * 16);
    const uint32_t my_off = wave_off + (uint32_t)(lane * kEnt * 16);
    const uint32_t null_off = wave_off + (uint32_t)(LSR_WAVE * kEnt * 16);
    if (lane < 8) s_list[lane][LSR_WAVE] = null_off;   // the rows' pad word (never a real entry)
    if (lane == 0) {
        s_ent[LSR_WAVE][0] = make_float4(__builtin_nanf(""), 0.0f, 0.0f, 0.0f);     // e' = NaN fails the keep test
        s_ent[LSR_WAVE][1] = make_float4(0.0f, 0.0f, 0.0f, RECORD ? 0.0f : -kInv255);
#pragma unroll
        for (int c4 = 0; c4 < NCHP / 4; ++c4) s_ent[LSR_WAVE][2 + c4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    constexpr uint32_t kEmu = EMU ? (uint32_t)EMU : 1u;     // (a divisor the EMU = 0 instances can compile)
    const uint32_t num_items = p.num_items * kEmu;
    const int coff = p.has_color ? 3 : 0;
    // RECORD: -1/255 lives in a register pair (the staged record's fourth word of its second quad collects the hit bits)
    float2_t kz2 = float2_t{-kInv255, -kInv255};
    asm volatile("" : "+s"(kz2));     // (a scalar register pair: the one constant-bus operand of the two packed operations below)
    const uint32_t gbit = 1u << (lane >> 3);
    const size_t HW = (size_t)p.H * p.W;
    // lane group -> sub-block (gcol, grow) of the half (bit 4*grow + gcol of the list entries' mask);
    // lane -> its two pixels (lx, ly), (lx + 1, ly) of the sub-block
    const int grp = lane >> 3, gcol = grp & 3, grow = grp >> 2;
    const int lx = 2 * (lane & 1), ly = (lane >> 1) & 3;

    const uint32_t simd_bins = (uint32_t)p.num_cus * 4u, slots = (uint32_t)p.num_cus * (uint32_t)p.waves_per_cu;
    const uint32_t vwave = (uint32_t)wid + (uint32_t)WPB * (blockIdx.x / (uint32_t)p.num_cus);   // 0 .. waves_per_cu - 1
    // issue priority by progress (k_render_bwd's comment): high until prio_target entries are left of the wave's list, then low
    uint32_t prio_target = p.prio_pct ? (uint32_t)(((uint64_t)p.header[kHdrPairs] * (uint32_t)p.prio_pct) / (50ull * (uint64_t)max(num_items, 1u))) : 0u;
    if (prio_target < 3u * LSR_WAVE) prio_target = 0u;
    const uint32_t bin = (blockIdx.x % (uint32_t)p.num_cus) * 4u + (vwave & 3u);
    // First item of every wave: static, folded (boustrophedon) over the cost-sorted list, so the
    // waves of a SIMD start with a balanced total.  Everything beyond the first `slots` items
    // is pulled from a global queue (costliest first) as waves become free.
    const uint32_t j0 = vwave >> 2;
    bool first = true;
    for (;;) {
        uint32_t qi;
        if (first) {
            qi = (j0 & 1u) ? (j0 + 1u) * simd_bins - 1u - bin : j0 * simd_bins + bin;
            first = false;
            if (qi >= num_items) continue;   // fewer items than wave slots: go straight to the queue (empty)
        } else {
            if (num_items <= slots) break;
            uint32_t t = 0;
            if (p.bin_queue) {
                // per-SIMD-bin lists (round 6): the bin goes on with ITS ranks b, 2B-1-b, 2B+b, ... past the static first items
                // instead of pulling from one global queue.  With the global queue the SIMDs of the 16-view launch end up with
                // 8 +- 1 items — one item is a tenth of a SIMD's work, and the per-SIMD iteration totals (which set the
                // launch time: every SIMD runs at 141 cycles per iteration from its first to its last item, item trace in
                // profiles/r06_ab_knobs.md) spread by +-10 %; the bins' totals are balanced by construction
                if (lane == 0) t = atomicAdd(p.bin_queue + bin, 1u);
                const uint32_t k = (uint32_t)p.waves_per_cu / 4u + __builtin_amdgcn_readfirstlane(t);
                qi = (k & 1u) ? (k + 1u) * simd_bins - 1u - bin : k * simd_bins + bin;
            } else {
                if (lane == 0) t = atomicAdd(p.queue, 1u);
                qi = slots + __builtin_amdgcn_readfirstlane(t);
            }
            if (qi >= num_items) break;
        }
#ifdef LSR_ENABLE_TRACE
        const unsigned long long t_begin = p.trace ? __builtin_readcyclecounter() : 0ull;
        unsigned long long t_first = 0ull, t_loop = 0ull;   // first batch staged / batch loop left
        uint32_t trace_iters = 0;   // lock-step iterations of this item
#endif
        qi = __builtin_amdgcn_readfirstlane(qi);
        const uint32_t item = p.items[qi / kEmu];
        const uint32_t vt = item & kItemTileMask, half = item >> kItemHalfShift;
        const int tile = (int)(vt % (uint32_t)p.T), v = (int)(vt / (uint32_t)p.T);
        const int tx0 = (tile % p.gx) * LSR_TILE, ty0 = (tile / p.gx) * LSR_TILE + 8 * (int)half;
        const size_t vG = (size_t)v * p.G;
        const uint32_t tstart = p.tile_start[vt], tn = p.tile_start[vt + 1] - tstart;
        uint32_t hn = p.half_count[2 * (size_t)vt + half];
        const uint32_t *hlist = p.half_list + 2 * (size_t)tstart + (size_t)half * tn;
        if (EMU) {   // this wave's share of the item's batches
            const uint32_t nbat = (hn + LSR_WAVE - 1) / LSR_WAVE, part = qi % kEmu;
            const uint32_t b0 = part * nbat / kEmu, b1 = (part + 1u) * nbat / kEmu;
            hlist += b0 * LSR_WAVE;
            hn = min(hn, b1 * LSR_WAVE) - min(hn, b0 * LSR_WAVE);
        }

        // per-pixel state of the lane's two pixels as register pairs (pixel 0, pixel 1): the blend runs on
        // packed f32 instructions across the two pixels
        const int px = tx0 + 4 * gcol + lx, py = ty0 + 4 * grow + ly;
        const bool inside0 = px < p.W && py < p.H, inside1 = px + 1 < p.W && py < p.H;
        // a finished (or outside) pixel gets x = NaN: its exponent is NaN and fails the keep test from then on, so
        // the loop needs no "done" masks
        float2_t pxx = float2_t{inside0 ? (float)px : __builtin_nanf(""), inside1 ? (float)(px + 1) : __builtin_nanf("")};
        const float pyf = (float)py;
        float2_t T2 = float2_t{1.0f, 1.0f};          // transmittance
        float2_t D2 = float2_t{0.0f, 0.0f};          // sum alpha T z
        float2_t acc[NCHP];
#pragma unroll
        for (int c = 0; c < NCHP; ++c) acc[c] = float2_t{0.0f, 0.0f};
        uint32_t stop_pos0 = 0, stop_pos1 = 0;
        float kmax = kAlphaMax255;
        asm volatile("" : "+v"(kmax));   // kept in a register: as a literal it makes both v_min 64-bit encodings
        // lanes whose two pixels are both finished (only consulted between batches)
        uint64_t done0 = __ballot(!inside0), done1 = __ballot(!inside1);

        // Software-pipelined staging: while batch b is composited, the records of batch b+1 and the
        // list entries of batch b+2 are already in flight (two dependent global latencies per batch
        // otherwise sit on the serial path of the wave).
        // The loads are UNCONDITIONAL at clamped positions (slots past the end re-read the last entry, which
        // the `e < hn` tests below ignore): a load under a per-lane condition is followed by a wait for it at
        // the join, which put both latencies back on the serial path of every batch.
        struct StageRec { float4 a, b, pay[NCHP / 4]; uint32_t w; };
        const uint32_t last = hn - 1u;   // only used when the list is not empty
        auto load_ent = [&](uint32_t e) -> uint32_t { return hlist[min(e, last)]; };
        auto load_rec = [&](uint32_t w) {
            StageRec r;
            r.w = w;
            const float4 *R = p.rec + (vG + (w & p.ip.index_mask)) * (size_t)p.rec_f4;
            r.a = R[0]; r.b = R[1];  // (x,y,A,B) (C,o,z,-)
#pragma unroll
            for (int c4 = 0; c4 < NCHP / 4; ++c4) r.pay[c4] = R[2 + c4];  // payload, zero padded
            return r;
        };
        uint32_t w_ahead = 0;
        StageRec nxt;
        nxt.w = 0;
        nxt.a = nxt.b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int c4 = 0; c4 < NCHP / 4; ++c4) nxt.pay[c4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (hn > 0) {   // wave-uniform
            w_ahead = load_ent(lane);
            nxt = load_rec(w_ahead);
            w_ahead = load_ent(LSR_WAVE + lane);
        }

        uint32_t burn_end = 0u;
        if (prio_target && hn > prio_target) { burn_end = hn - prio_target; __builtin_amdgcn_s_setprio(3); }
        // RECORD: lanes that staged an entry of opacity >= kSteepOpacity — a pixel of this item may have blended an alpha near
        // the 0.99 clamp, and the backward then walks the item back to front (render_backward.hip); a scalar register pair
        uint64_t steep = 0ull;
        uint32_t spent = 0u;      // RECORD: lock-step iterations the BACKWARD will spend on this item (its place in the backward's work order)
        for (uint32_t base = 0; base < hn; base += LSR_WAVE) {
            if ((done0 & done1) == ~0ull) break;

            const StageRec cur = nxt;
            nxt = load_rec(w_ahead);
            w_ahead = load_ent(base + 2 * LSR_WAVE + lane);
            // ---- stage up to 64 list entries (one per lane) ----
            // every list slot starts as the null record; the compaction below overwrites a prefix
#pragma unroll
            for (int b = 0; b < 8; ++b) s_list[b][lane] = null_off;
            const uint32_t e = base + lane;
            // sub-blocks that still have a pixel to blend (scalar: the lane masks of the finished pixels).  An entry is staged
            // and listed only for those — a finished sub-block's list would still count into the lock-step length of every
            // batch behind its stop.  Scenes whose pixels run out of transmittance — opaque 1-10 px splats: this kernel 0.282 ->
            // 0.222 ms per 16 views, the row items of 4 views 0.214 -> 0.111; encoder-shaped configs[4] 0.451 -> 0.421; the bench
            // scene's pixels practically never finish: unchanged (profiles/r06_ab_knobs.md section 13)
            uint32_t live = 0xFFu;
            if ((done0 | done1) && !p.all_live) {
                const uint64_t nd = ~(done0 & done1);
                live = 0u;
#pragma unroll
                for (int g = 0; g < 8; ++g) live |= (uint32_t)(((nd >> (8 * g)) & 0xFFull) != 0ull) << g;
            }
            const uint32_t m_full = e < hn ? (((cur.w >> kListBitsShift) | p.ip.all_bits) & 0xFFu) : 0u;   // sub-blocks of this half the entry can reach (never 0 for a list entry)
            const uint32_t m = m_full & live;
            if (RECORD) steep |= __ballot(m != 0u && cur.b.y >= kSteepOpacity);
            if (m) {
                const float4 a = cur.a, b = cur.b;
                const FoldedConic f = fold_conic(a.z, a.w, b.x, b.y);
                s_ent[lane][0] = make_float4(a.x, a.y, f.a2, f.c2);
                s_ent[lane][1] = make_float4(f.b2, f.l2o, b.z * kInv255, RECORD ? 0.0f : -kInv255);   // (RECORD: hit bits, none yet)
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4)
                    s_ent[lane][2 + c4] = make_float4(cur.pay[c4].x * kInv255, cur.pay[c4].y * kInv255, cur.pay[c4].z * kInv255, cur.pay[c4].w * kInv255);
            }
            // compaction: per sub-block, the staged entries that can reach it, in list order
            uint32_t nk = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t bal = __ballot((m >> b) & 1u);
                nk = max(nk, (uint32_t)__builtin_popcountll(bal));
                if (__builtin_amdgcn_inverse_ballot_w64(bal)) {
                    const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    s_list[b][at] = my_off;
                }
            }
            nk = __builtin_amdgcn_readfirstlane(nk);
            wave_lds_fence();  // staged records and lists are visible to this wave's reads below

            // Lane group g walks the list of ITS sub-block, front to back; shorter lists are padded with the
            // null record.  Per entry: the row terms once, then both pixels in packed arithmetic.
            const uint32_t *lp = &s_list[grp][0];
#ifdef LSR_ENABLE_TRACE
            trace_iters += nk;
            if (p.trace && base == 0u) t_first = __builtin_readcyclecounter();
#endif
            uint32_t hist = 0u;    // RECORD: one bit per iteration of the current chunk, newest in bit 0: this lane's pixels took part
            // RECORD: the list word of the NEXT iteration is read while this one computes (row 64 of a list is its pad word).
            // Measured between two builds (profiles/r05_ab_knobs.md): the RECORD instance 0.228 -> 0.221 ms per 16 views, the
            // plain instance 0.2051 -> 0.2080 — so only here.
            uint32_t off_next = RECORD ? lp[0] : 0u;
            auto entry_step = [&](uint32_t i) __attribute__((always_inline)) {
                const uint32_t off = RECORD ? off_next : lp[i];
                const float4 *E = (const float4 *)(ent_base + off);
                const float4 a = E[0], b = E[1];
                if (RECORD) off_next = lp[i + 1];
                float2_t pay[NCHP / 2];     // (c, c+1) pairs: either half is broadcast to both pixels by the packed ops' op_sel
#pragma unroll
                for (int c4 = 0; c4 < NCHP / 4; ++c4) {
                    const float4 t = E[2 + c4];
                    pay[2 * c4] = float2_t{t.x, t.y}; pay[2 * c4 + 1] = float2_t{t.z, t.w};
                }
                const float2_t zk = float2_t{b.z, b.w};   // (z / 255, -1 / 255)
                // e'(dx) = dx (a2 dx + b2 dy) + (c2 dy^2 + l2o') for dx = dx0, dx0 - 1 (lsr_blend.h; the backward
                // performs the identical sequence)
                const float2_t d2 = float2_t{a.x, a.x} - pxx;
                const float dy = a.y - pyf;
                const float t = b.x * dy;
                const float s = __builtin_fmaf(a.w * dy, dy, b.y);
                const float2_t p1 = __builtin_elementwise_fma(float2_t{a.z, a.z}, d2, float2_t{t, t});
                const float2_t ex = __builtin_elementwise_fma(p1, d2, float2_t{s, s});
                float2_t al;                                                                                   // 255 alpha
                if (RECORD) {   // (v_min as written: fminf makes the compiler re-canonicalise kmax in every iteration of this register-tight instance)
                    const float e0 = fast_exp2(ex.x), e1 = fast_exp2(ex.y);
                    float m0, m1;
                    asm("v_min_f32 %0, %1, %2" : "=v"(m0) : "v"(kmax), "v"(e0));
                    asm("v_min_f32 %0, %1, %2" : "=v"(m1) : "v"(kmax), "v"(e1));
                    al = float2_t{m0, m1};
                } else al = float2_t{fminf(kmax, fast_exp2(ex.x)), fminf(kmax, fast_exp2(ex.y))};
                // keep <=> 0 <= e' <= l2o' (alpha >= 1/255 and power <= 0): one unsigned comparison of the float bits
                const uint32_t lim = __float_as_uint(b.y);
                const uint64_t ok0 = __ballot(__float_as_uint(ex.x) <= lim), ok1 = __ballot(__float_as_uint(ex.y) <= lim);
                // aT = 255 alpha T, tT = T (1 - alpha): ONE asm block, so that the compiler's hazard recognizer (which puts a wait
                // state behind every packed-f32 result that the next instruction reads; the hardware needs none:
                // tools/microbench/pk_hazard.hip) cannot separate the pair
                float2_t aT, tT;
                if (RECORD) asm("v_pk_mul_f32 %0, %2, %3\n\tv_pk_fma_f32 %1, %0, %4, %3" : "=&v"(aT), "=v"(tT) : "v"(al), "v"(T2), "s"(kz2));
                else asm("v_pk_mul_f32 %0, %2, %3\n\tv_pk_fma_f32 %1, %0, %4, %3 op_sel:[0,1,0]" : "=&v"(aT), "=v"(tT) : "v"(al), "v"(T2), "v"(zk));
                const uint64_t room0 = __ballot(tT.x >= LSR_T_EPS), room1 = __ballot(tT.y >= LSR_T_EPS);
                const uint64_t stop0 = ok0 & ~room0, stop1 = ok1 & ~room1;
                const uint64_t hit0 = ok0 & room0, hit1 = ok1 & room1;
                const float w0 = __builtin_amdgcn_inverse_ballot_w64(hit0) ? aT.x : 0.0f;
                const float w1 = __builtin_amdgcn_inverse_ballot_w64(hit1) ? aT.y : 0.0f;
                const float2_t ww = float2_t{w0, w1};
                if (RECORD) {   // hist = 2 hist + (one of this lane's pixels blended the entry): one add-with-carry, the carry-in being the lane mask
                    unsigned long long carry_out;
                    asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(hist), "=s"(carry_out) : "s"(hit0 | hit1));
                }
#pragma unroll
                for (int c = 0; c < NCHP; c += 2) {
                    pk_fma_splat<false>(acc[c], pay[c / 2], ww);
                    pk_fma_splat<true>(acc[c + 1], pay[c / 2], ww);
                }
                pk_fma_splat<false>(D2, zk, ww);     // depth  += (z / 255) w'
                if (RECORD) asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(T2) : "s"(kz2), "v"(ww));   // T -= w' / 255
                else pk_fma_splat<true>(T2, zk, ww);
                if (stop0 | stop1) {  // rare, wave-uniform: a pixel's transmittance ran out here
                    // 1-based list position of the stopping entry, from its staging slot
                    const uint32_t pos = base + 1u + (off - wave_off) / (uint32_t)(kEnt * 16);
                    const bool st0 = __builtin_amdgcn_inverse_ballot_w64(stop0), st1 = __builtin_amdgcn_inverse_ballot_w64(stop1);
                    stop_pos0 = st0 ? pos : stop_pos0;
                    stop_pos1 = st1 ? pos : stop_pos1;
                    pxx = float2_t{st0 ? __builtin_nanf("") : pxx.x, st1 ? __builtin_nanf("") : pxx.y};   // never kept again
                    done0 |= stop0; done1 |= stop1;
                }
            };
            if (!RECORD) {
                for (uint32_t i = 0; i < nk; ++i) entry_step(i);
            } else {
                // chunks of 32 iterations (the history register); after a chunk the eight lanes of a group OR their
                // histories and lane j marks the group's bit on the records of the entries of iterations j, j + 8, ...
                const int l8 = lane & 7;
                for (uint32_t c0 = 0; c0 < nk; c0 += 32u) {
                    const uint32_t c1 = min(nk, c0 + 32u);
                    hist = 0u;
                    for (uint32_t i = c0; i < c1; ++i) entry_step(i);
                    hist |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hist, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
                    hist |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hist, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
                    hist |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hist, 0x141, 0xf, 0xf, false);   // row_half_mirror
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t i = c0 + (uint32_t)(l8 + 8 * r);
                        if (i < c1 && ((hist >> (c1 - 1u - i)) & 1u)) atomicOr((unsigned int *)(ent_base + lp[i] + 28), gbit);
                    }
                }
            }
            wave_lds_fence();  // WAR on the LDS slice before the next batch is staged
            if (RECORD) {
                // the staged entry's sub-block bits <- the sub-blocks it contributed to (a subset of them)
                uint32_t hits = 0u;
                if (m_full) {     // (an entry all of whose sub-blocks had finished was not staged: no hits)
                    if (m) hits = ((const uint32_t *)&s_ent[lane][1])[3] & 0xFFu;
                    if (hits != m_full) p.half_list_rw[(hlist - p.half_list) + e] = (cur.w & kListIndexMask) | (hits << kListBitsShift);
                }
                if (p.item_cost) {     // the iterations the BACKWARD will spend on this batch: its longest narrowed sub-block list
                    uint32_t longest = 0u;
#pragma unroll
                    for (int g = 0; g < 8; ++g) longest = max(longest, (uint32_t)__builtin_popcountll(__ballot((hits >> g) & 1u)));
                    spent += longest;
                }
                wave_lds_fence();
            }
            if (burn_end && base + LSR_WAVE >= burn_end) { __builtin_amdgcn_s_setprio(0); burn_end = 0u; }
        }
        if (burn_end) __builtin_amdgcn_s_setprio(0);
#ifdef LSR_ENABLE_TRACE
        if (p.trace) t_loop = __builtin_readcyclecounter();
#endif
        if (RECORD && lane == 0) {
            if (p.item_cost) p.item_cost[2 * (size_t)vt + half] = spent;
            if (steep) p.item_flags[2 * (size_t)vt + half] = kItemFlagSteep;
            if (qi == 0u) p.header_rw[kHdrFlagsValid] = 1u;
        }

        // background colour through the scalar cache (constant address space; the table was written by an
        // earlier launch) — as plain loads these were three waited-for vector loads per pixel row
        typedef const float __attribute__((address_space(4))) *kfloat_ptr;
        const kfloat_ptr vw = (kfloat_ptr)(p.views + (size_t)v * LSR_VIEW_FLOATS);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!(k ? inside1 : inside0)) continue;
            const size_t pix = (size_t)py * p.W + (size_t)(px + k);
            const size_t vp = (size_t)v * HW + pix;
            const float Tk = T2[k];
            const uint32_t sp = k ? stop_pos1 : stop_pos0;
            p.final_T[vp] = Tk;
            // number of leading entries of the HALF's render list the backward pass has to consider for
            // this pixel: all of them, or everything before the entry at which the transmittance test stopped it
            p.n_contrib[vp] = sp ? sp - 1u : hn;
            p.out_mask[vp] = 1.0f - Tk;
            p.out_depth[vp] = D2[k];
            if (p.has_color) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    p.out_color[((size_t)v * 3 + c) * HW + pix] = __builtin_fmaf(Tk, vw[37 + c], acc[c][k]);
            }
#pragma unroll
            for (int c = 0; c < NCHP; ++c)
                if (c >= coff && c - coff < p.C)
                    p.out_feat[((size_t)v * p.C + (c - coff)) * HW + pix] = acc[c][k];
        }
#ifdef LSR_ENABLE_TRACE
        if (p.trace && lane == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            p.trace[6 * (size_t)qi + 0] = t_begin;
            p.trace[6 * (size_t)qi + 1] = __builtin_readcyclecounter();
            p.trace[6 * (size_t)qi + 2] = ((unsigned long long)xcc << 32) | hwid;
            p.trace[6 * (size_t)qi + 3] = ((unsigned long long)trace_iters << 32) | hn;
            p.trace[6 * (size_t)qi + 4] = t_first;
            p.trace[6 * (size_t)qi + 5] = t_loop;
        }
#endif
    }  // persistent item loop
}

// Work order of the compositing BACKWARD (round 6): one workgroup turns the costs the RECORD forward wrote (lock-step iterations
// per item) into tile_order2 — the items costliest first, a counting sort over kCostClasses classes of two iterations in LDS —
// and raises header[kHdrOrder2Valid].  The tile scan's order (by the tile's pair count, both halves adjacent) misjudges an
// item by up to +-25 %; with the true cost the backward's longest-processing-time-first queue ends 6-8 % earlier.
constexpr int kOrderThreads = 1024;
__global__ void __launch_bounds__(kOrderThreads)
k_order_items(const uint32_t *__restrict__ cost, uint32_t *__restrict__ order2, uint32_t *header, uint32_t num_items) {
    __shared__ uint32_t s_cnt[kCostClasses], s_wsum[kOrderThreads / LSR_WAVE];
    const int tid = threadIdx.x, lane = tid & (LSR_WAVE - 1), wid = tid / LSR_WAVE;
    static_assert(kCostClasses == kOrderThreads, "one class per thread in the scan");
    s_cnt[tid] = 0u;
    __syncthreads();
    constexpr int kPer = kMaxReorderItems / kOrderThreads;
    uint32_t cls[kPer], rnk[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const uint32_t i = (uint32_t)tid + (uint32_t)q * kOrderThreads;
        cls[q] = 0u; rnk[q] = 0u;
        if (i < num_items) {
            cls[q] = kCostClasses - 1u - min((uint32_t)kCostClasses - 1u, cost[i] >> 1);     // class 0 = the costliest
            rnk[q] = atomicAdd(&s_cnt[cls[q]], 1u);
        }
    }
    __syncthreads();
    const uint32_t mine = s_cnt[tid];
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < LSR_WAVE; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == LSR_WAVE - 1) s_wsum[wid] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (int w = 0; w < wid; ++w) run += s_wsum[w];
    __syncthreads();
    s_cnt[tid] = run;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const uint32_t i = (uint32_t)tid + (uint32_t)q * kOrderThreads;
        if (i < num_items) order2[s_cnt[cls[q]] + rnk[q]] = (i >> 1) | ((i & 1u) << kItemHalfShift);
    }
    if (tid == 0) header[kHdrOrder2Valid] = 1u;
}

// ---- small view batches (round 4): one wave = one ROW of four sub-blocks (16 x 4 pixels), one pixel per lane ----
// A launch over few views cannot fill the wave slots: 512 half tiles of a single 256 x 256 view on 6144 slots leave every
// wave alone on its SIMD, walking a ~900-entry list with every dependent latency exposed (V = 1: 0.078 ms, V = 4: 0.088 ms
// for 1/16 and 1/4 of the 16-view launch's work).  Here each half tile is rendered by TWO waves — rows 0-3 and rows 4-7 of
// the half — each walking the same half-tile list but compacting only its own four sub-block lists (16-lane groups, one
// pixel per lane): twice the waves, fewer iterations per batch (the longest of four lists instead of eight) and a
// shorter, unpacked loop body.  Per evaluation it costs ~1.5x the instructions of the two-pixel kernel, which is why the
// launcher uses it only when the two-pixel launch would leave wave slots empty.
// The arithmetic per pixel is the two-pixel kernel's, operation for operation (a packed f32 operation is two IEEE
// operations): images, final_T and n_contrib are bitwise identical, and so are the keep / skip decisions the backward
// (which recomputes them with lsr_blend.h's sequence) relies on.
// ---- small view batches: finer work items for launches that cannot fill the wave slots ----
// Per-wave LDS slice of k_render_fwd_small and what both item kinds need of it.
template <int NCHP>
struct SmallWave {
    static constexpr int kEnt = (2 + NCHP / 4) | 1;
    float4 (*s_ent)[kEnt];        // [64 staged entries + the null entry][kEnt]
    uint32_t *s_flat;             // 4 x 65 list words: row items use them as [4][65], sub-block items the first 68
    const char *ent_base;
    uint32_t wave_off, my_off, null_off;
    int lane, coff;
    size_t HW;
};

// ROW item: one wave = one row of four sub-blocks (16 x 4 pixels) of a half tile; 16-lane groups, one pixel per lane, the
// two-pixel kernel's arithmetic per pixel (bitwise identical outputs).
// REC (forwards a backward follows): the wave notes which of its row's four sub-blocks blended each entry and clears the
// other bits of ITS nibble of the entry's list word (atomically: the half tile's other row wave owns the other nibble) —
// the narrowing of k_render_fwd's RECORD instance, per row.
template <int NCHP, bool REC>
__device__ __forceinline__ void render_row_item(const RenderFwdParams &p, const SmallWave<NCHP> &w_, uint32_t item, int grow) {
    constexpr int kEnt = SmallWave<NCHP>::kEnt;
    const int lane = w_.lane, coff = w_.coff;
    const size_t HW = w_.HW;
    float4 (*s_ent)[kEnt] = w_.s_ent;
    uint32_t (*s_list)[LSR_WAVE + 1] = (uint32_t (*)[LSR_WAVE + 1])w_.s_flat;
    const char *ent_base = w_.ent_base;
    const uint32_t wave_off = w_.wave_off, my_off = w_.my_off, null_off = w_.null_off;
    const int gcol = lane >> 4, lx = lane & 3, ly = (lane >> 2) & 3;   // 16-lane group -> sub-block column; lane -> pixel of the sub-block
    const uint32_t vt = item & kItemTileMask, half = item >> kItemHalfShift;
    const int tile = (int)(vt % (uint32_t)p.T), v = (int)(vt / (uint32_t)p.T);
    const int tx0 = (tile % p.gx) * LSR_TILE, ty0 = (tile / p.gx) * LSR_TILE + 8 * (int)half + 4 * grow;
    const size_t vG = (size_t)v * p.G;
    const uint32_t tstart = p.tile_start[vt], tn = p.tile_start[vt + 1] - tstart;
    const uint32_t hn = p.half_count[2 * (size_t)vt + half];
    const uint32_t *hlist = p.half_list + 2 * (size_t)tstart + (size_t)half * tn;

    const int px = tx0 + 4 * gcol + lx, py = ty0 + ly;
    const bool inside = px < p.W && py < p.H;
    float pxx = inside ? (float)px : __builtin_nanf("");   // a finished (or outside) pixel gets x = NaN: never kept again
    const float pyf = (float)py;
    float T = 1.0f, D = 0.0f;
    float acc[NCHP];
#pragma unroll
    for (int c = 0; c < NCHP; ++c) acc[c] = 0.0f;
    uint32_t stop_pos = 0;
    float kmax = kAlphaMax255;
    asm volatile("" : "+v"(kmax));
    uint64_t done = __ballot(!inside);
    uint64_t steep = 0ull;    // lanes that staged an entry of opacity >= kSteepOpacity (see k_render_fwd)
    float kz = -kInv255;      // REC: -1/255 lives in a register (the staged record's word collects the hit bits)
    asm volatile("" : "+v"(kz));

    struct StageRec { float4 a, b, pay[NCHP / 4]; uint32_t w; };
    const uint32_t last = hn - 1u;
    auto load_ent = [&](uint32_t e) -> uint32_t { return hlist[min(e, last)]; };
    auto load_rec = [&](uint32_t w) {
        StageRec r;
        r.w = w;
        const float4 *R = p.rec + (vG + (w & p.ip.index_mask)) * (size_t)p.rec_f4;
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
    if (hn > 0) {
        w_ahead = load_ent(lane);
        nxt = load_rec(w_ahead);
        w_ahead = load_ent(LSR_WAVE + lane);
    }
    uint32_t base = 0;
    for (; base < hn; base += LSR_WAVE) {
        if (done == ~0ull) break;
        const StageRec cur = nxt;
        nxt = load_rec(w_ahead);
        w_ahead = load_ent(base + 2 * LSR_WAVE + lane);
#pragma unroll
        for (int b = 0; b < 4; ++b) s_list[b][lane] = null_off;
        const uint32_t e = base + lane;
        // the four sub-blocks of THIS row the entry can reach (bits 4 grow .. 4 grow + 3 of its half mask) ...
        const uint32_t m_full = e < hn ? (((((cur.w >> kListBitsShift) | p.ip.all_bits) & 0xFFu) >> (4 * grow)) & 0xFu) : 0u;
        // ... and of those the ones that still have a pixel to blend (see k_render_fwd)
        uint32_t live = 0xFu;
        if (done && !p.all_live) {
            live = 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g) live |= (uint32_t)(((~done >> (16 * g)) & 0xFFFFull) != 0ull) << g;
        }
        const uint32_t m = m_full & live;
        steep |= __ballot(m != 0u && cur.b.y >= kSteepOpacity);
        if (m) {
            const float4 a = cur.a, b = cur.b;
            const FoldedConic f = fold_conic(a.z, a.w, b.x, b.y);
            s_ent[lane][0] = make_float4(a.x, a.y, f.a2, f.c2);
            s_ent[lane][1] = make_float4(f.b2, f.l2o, b.z * kInv255, REC ? 0.0f : -kInv255);   // (REC: hit bits, none yet)
#pragma unroll
            for (int c4 = 0; c4 < NCHP / 4; ++c4)
                s_ent[lane][2 + c4] = make_float4(cur.pay[c4].x * kInv255, cur.pay[c4].y * kInv255, cur.pay[c4].z * kInv255, cur.pay[c4].w * kInv255);
        }
        uint32_t nk = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint64_t bal = __ballot((m >> b) & 1u);
            nk = max(nk, (uint32_t)__builtin_popcountll(bal));
            if (__builtin_amdgcn_inverse_ballot_w64(bal)) {
                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                s_list[b][at] = my_off;
            }
        }
        nk = __builtin_amdgcn_readfirstlane(nk);
        wave_lds_fence();
        const uint32_t *lp = &s_list[gcol][0];
        uint32_t hist = 0u;    // REC: one bit per iteration of the current chunk, newest in bit 0: this lane's pixel blended the entry
        auto entry_step = [&](uint32_t i) {
            const uint32_t off = lp[i];
            const float4 *E = (const float4 *)(ent_base + off);
            const float4 a = E[0];
            float4 b = E[1];
            if (REC) b.w = kz;
            float pay[NCHP];
#pragma unroll
            for (int c4 = 0; c4 < NCHP / 4; ++c4) {
                const float4 t = E[2 + c4];
                pay[4 * c4] = t.x; pay[4 * c4 + 1] = t.y; pay[4 * c4 + 2] = t.z; pay[4 * c4 + 3] = t.w;
            }
            // e' = dx (a2 dx + b2 dy) + (c2 dy^2 + l2o'): the two-pixel kernel's sequence for one pixel
            const float d = a.x - pxx;
            const float dy = a.y - pyf;
            const float t = b.x * dy;
            const float s = __builtin_fmaf(a.w * dy, dy, b.y);
            const float p1 = __builtin_fmaf(a.z, d, t);
            const float ex = __builtin_fmaf(p1, d, s);
            const float al = fminf(kmax, fast_exp2(ex));                       // 255 alpha
            const uint32_t lim = __float_as_uint(b.y);
            const uint64_t ok = __ballot(__float_as_uint(ex) <= lim);
            const float aT = al * T;                                          // 255 alpha T
            const float tT = __builtin_fmaf(aT, b.w, T);                      // T (1 - alpha)   (b.w = -1 / 255)
            const uint64_t room = __ballot(tT >= LSR_T_EPS);
            const uint64_t stop = ok & ~room;
            const uint64_t hit = ok & room;
            const float w = __builtin_amdgcn_inverse_ballot_w64(hit) ? aT : 0.0f;
            if (REC) {   // hist = 2 hist + (this lane's pixel blended the entry): one add-with-carry, the carry-in being the lane mask
                uint64_t carry_out;
                asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(hist), "=s"(carry_out) : "s"(hit));
            }
#pragma unroll
            for (int c = 0; c < NCHP; ++c) acc[c] = __builtin_fmaf(pay[c], w, acc[c]);
            D = __builtin_fmaf(b.z, w, D);                                    // depth += (z / 255) w'
            T = __builtin_fmaf(b.w, w, T);                                    // T     -= w' / 255
            if (stop) {
                const uint32_t pos = base + 1u + (off - wave_off) / (uint32_t)(kEnt * 16);
                const bool st = __builtin_amdgcn_inverse_ballot_w64(stop);
                stop_pos = st ? pos : stop_pos;
                pxx = st ? __builtin_nanf("") : pxx;
                done |= stop;
            }
        };
        if (!REC) {
            for (uint32_t i = 0; i < nk; ++i) entry_step(i);
        } else {
            // chunks of 32 iterations (the history register); after a chunk the sixteen lanes of a group OR their histories
            // and lane j marks the group's bit on the records of the entries of iterations j and j + 16
            const int l16 = lane & 15;
            const uint32_t gbit = 1u << gcol;
            for (uint32_t c0 = 0; c0 < nk; c0 += 32u) {
                const uint32_t c1 = min(nk, c0 + 32u);
                hist = 0u;
                for (uint32_t i = c0; i < c1; ++i) entry_step(i);
                hist |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hist, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
                hist |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hist, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
                hist |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hist, 0x141, 0xf, 0xf, false);   // row_half_mirror
                hist |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hist, 0x140, 0xf, 0xf, false);   // row_mirror
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t i = c0 + (uint32_t)(l16 + 16 * r);
                    if (i < c1 && ((hist >> (c1 - 1u - i)) & 1u)) atomicOr((unsigned int *)(ent_base + lp[i] + 28), gbit);
                }
            }
        }
        wave_lds_fence();
        if (REC) {
            // this row's nibble of the entry's list word <- the sub-blocks of the row it contributed to (a subset)
            if (m_full) {     // (an entry all of whose sub-blocks had finished was not staged: no hits)
                const uint32_t hits = m ? ((const uint32_t *)&s_ent[lane][1])[3] & 0xFu : 0u;
                if (hits != m_full) atomicAnd((unsigned int *)&p.half_list_rw[(hlist - p.half_list) + e], ~(((m_full & ~hits) << (4 * grow)) << kListBitsShift));
            }
            wave_lds_fence();
        }
    }
    if (REC) {
        // every pixel of the row is finished: the entries behind blend nothing here — without this the other row's longer walk
        // would leave this row's sub-blocks their footprint-box bits there, and the backward's lock-step batches their length
        const uint32_t nib = 0xFu << (4 * grow + kListBitsShift);
        for (uint32_t e = base + (uint32_t)lane; e < hn; e += LSR_WAVE)
            if (hlist[e] & nib) atomicAnd((unsigned int *)&p.half_list_rw[(hlist - p.half_list) + e], ~nib);
    }
    if (steep && lane == 0) p.item_flags[2 * (size_t)vt + half] = kItemFlagSteep;
    typedef const float __attribute__((address_space(4))) *kfloat_ptr;
    const kfloat_ptr vw = (kfloat_ptr)(p.views + (size_t)v * LSR_VIEW_FLOATS);
    if (inside) {
        const size_t pix = (size_t)py * p.W + (size_t)px;
        const size_t vp = (size_t)v * HW + pix;
        p.final_T[vp] = T;
        p.n_contrib[vp] = stop_pos ? stop_pos - 1u : hn;
        p.out_mask[vp] = 1.0f - T;
        p.out_depth[vp] = D;
        if (p.has_color) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                p.out_color[((size_t)v * 3 + c) * HW + pix] = __builtin_fmaf(T, vw[37 + c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NCHP; ++c)
            if (c >= coff && c - coff < p.C)
                p.out_feat[((size_t)v * p.C + (c - coff)) * HW + pix] = acc[c];
    }
}

// One view per call (the reference's own call pattern, cuda_splatting.py:124-162): 512 half tiles on 6144 wave slots, and
// the launch lasts as long as the longest list takes ONE wave.  Here a wave renders one 4x4 SUB-BLOCK (eight waves per half
// tile) and its four 16-lane rows evaluate FOUR consecutive entries of the sub-block's list at once: the exponent, exp2 and
// the payload multiply-adds — most of an entry's instructions — run in parallel, and only the transmittance recurrence
// T' = fma(-1/255, keep ? 255 alpha T : 0, T) stays serial: every row receives the four 255-alpha values of its pixel
// (v_permlane32_swap / v_permlane16_swap, gfx950) and runs the four steps redundantly, so T, the keep / stop decisions,
// final_T and n_contrib are bit for bit those of the other two kernels.  The accumulators are per-row partial sums added
// across the rows once per item: colour / feature / depth differ from the serial kernels in the order of that sum (~1e-7).
template <int NCHP>
__device__ __forceinline__ void render_subblock_item(const RenderFwdParams &p, const SmallWave<NCHP> &w_, uint32_t item, int sub) {
    constexpr int kEnt = SmallWave<NCHP>::kEnt;
    const int lane = w_.lane, coff = w_.coff;
    const size_t HW = w_.HW;
    float4 (*s_ent)[kEnt] = w_.s_ent;
    uint32_t *s_list = w_.s_flat;
    const char *ent_base = w_.ent_base;
    const uint32_t wave_off = w_.wave_off, my_off = w_.my_off, null_off = w_.null_off;
    const int slot = lane >> 4, lx = lane & 3, ly = (lane >> 2) & 3;   // 16-lane row -> entry slot of a step; lane -> pixel of the sub-block
    const uint32_t *my_list = s_list + slot;
    // value of x in this lane's pixel of row 0, 1, 2, 3 — in every row
    auto rows_of = [](float x, float &x0, float &x1, float &x2, float &x3) {
        const uint32_t u = __float_as_uint(x);
        const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);    // h[0] = rows (0, 1, 0, 1), h[1] = rows (2, 3, 2, 3)
        const auto lo = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);   // lo[0] = row 0 everywhere, lo[1] = row 1 everywhere
        const auto hi = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);
        x0 = __uint_as_float(lo[0]); x1 = __uint_as_float(lo[1]); x2 = __uint_as_float(hi[0]); x3 = __uint_as_float(hi[1]);
    };
    // (row 0 + row 1) + (row 2 + row 3) of this lane's pixel, in every row
    auto rows_sum = [](float x) -> float {
        const uint32_t u = __float_as_uint(x);
        const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);    // a[0] = rows (0, 0, 2, 2), a[1] = rows (1, 1, 3, 3)
        const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
        const uint32_t v = __float_as_uint(y);
        const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);    // b[0] = lower half everywhere, b[1] = upper half
        return __uint_as_float(b[0]) + __uint_as_float(b[1]);
    };

    const uint32_t vt = item & kItemTileMask, half = item >> kItemHalfShift;
    const int tile = (int)(vt % (uint32_t)p.T), v = (int)(vt / (uint32_t)p.T);
    const int tx0 = (tile % p.gx) * LSR_TILE + 4 * (sub & 3), ty0 = (tile / p.gx) * LSR_TILE + 8 * (int)half + 4 * (sub >> 2);
    const size_t vG = (size_t)v * p.G;
    const uint32_t tstart = p.tile_start[vt], tn = p.tile_start[vt + 1] - tstart;
    const uint32_t hn = p.half_count[2 * (size_t)vt + half];
    const uint32_t *hlist = p.half_list + 2 * (size_t)tstart + (size_t)half * tn;

    const int px = tx0 + lx, py = ty0 + ly;
    const bool inside = px < p.W && py < p.H;
    float pxx = inside ? (float)px : __builtin_nanf("");   // a finished (or outside) pixel gets x = NaN: never kept again
    const float pyf = (float)py;
    float T = 1.0f, D = 0.0f;                              // T: the pixel's transmittance, the same in all four rows; D, acc: this row's partial sums
    float acc[NCHP];
#pragma unroll
    for (int c = 0; c < NCHP; ++c) acc[c] = 0.0f;
    uint32_t stop_pos = 0;
    float kmax = kAlphaMax255;
    asm volatile("" : "+v"(kmax));
    uint64_t done = __ballot(!inside);
    uint64_t steep = 0ull;    // lanes that staged an entry of opacity >= kSteepOpacity (see k_render_fwd)

    struct StageRec { float4 a, b, pay[NCHP / 4]; uint32_t w; };
    const uint32_t last = hn - 1u;
    auto load_ent = [&](uint32_t e) -> uint32_t { return hlist[min(e, last)]; };
    auto load_rec = [&](uint32_t w) {
        StageRec r;
        r.w = w;
        const float4 *R = p.rec + (vG + (w & p.ip.index_mask)) * (size_t)p.rec_f4;
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
    if (hn > 0) {
        w_ahead = load_ent(lane);
        nxt = load_rec(w_ahead);
        w_ahead = load_ent(LSR_WAVE + lane);
    }
    uint32_t base = 0;
    for (; base < hn; base += LSR_WAVE) {
        if (done == ~0ull) break;
        const StageRec cur = nxt;
        nxt = load_rec(w_ahead);
        w_ahead = load_ent(base + 2 * LSR_WAVE + lane);
        s_list[lane] = null_off;
        if (lane < 4) s_list[LSR_WAVE + lane] = null_off;       // the padding of the last group of four (row items use these words)
        const uint32_t e = base + lane;
        const bool m = e < hn && (((((cur.w >> kListBitsShift) | p.ip.all_bits) & 0xFFu) >> sub) & 1u);
        steep |= __ballot(m && cur.b.y >= kSteepOpacity);
        if (m) {
            const float4 a = cur.a, b = cur.b;
            const FoldedConic f = fold_conic(a.z, a.w, b.x, b.y);
            s_ent[lane][0] = make_float4(a.x, a.y, f.a2, f.c2);
            s_ent[lane][1] = make_float4(f.b2, f.l2o, b.z * kInv255, -kInv255);
#pragma unroll
            for (int c4 = 0; c4 < NCHP / 4; ++c4)
                s_ent[lane][2 + c4] = make_float4(cur.pay[c4].x * kInv255, cur.pay[c4].y * kInv255, cur.pay[c4].z * kInv255, cur.pay[c4].w * kInv255);
        }
        const uint64_t bal = __ballot(m);
        const uint32_t nk = (uint32_t)__builtin_popcountll(bal);
        if (m) {
            const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            s_list[at] = my_off;
        }
        wave_lds_fence();
        for (uint32_t i = 0; i < nk; i += 4) {
            if (done == ~0ull) break;
            const uint32_t off = my_list[i];             // row r takes entry i + r of the list (null entries beyond its end)
            const float4 *E = (const float4 *)(ent_base + off);
            const float4 a = E[0], b = E[1];
            float pay[NCHP];
#pragma unroll
            for (int c4 = 0; c4 < NCHP / 4; ++c4) {
                const float4 t = E[2 + c4];
                pay[4 * c4] = t.x; pay[4 * c4 + 1] = t.y; pay[4 * c4 + 2] = t.z; pay[4 * c4 + 3] = t.w;
            }
            // e' = dx (a2 dx + b2 dy) + (c2 dy^2 + l2o'): the operation sequence of the other kernels
            const float d = a.x - pxx;
            const float dy = a.y - pyf;
            const float t = b.x * dy;
            const float s = __builtin_fmaf(a.w * dy, dy, b.y);
            const float p1 = __builtin_fmaf(a.z, d, t);
            const float ex = __builtin_fmaf(p1, d, s);
            const float al = fminf(kmax, fast_exp2(ex));                       // 255 alpha
            const float alz = __float_as_uint(ex) <= __float_as_uint(b.y) ? al : 0.0f;   // 0: the entry does not reach the pixel
            float a4[4];
            rows_of(alz, a4[0], a4[1], a4[2], a4[3]);
            // The four steps of the recurrence, every row for its pixel.  A pixel that is alive has T >= T_EPS (so does a
            // finished one: its T is the value before the entry that stopped it, and its 255 alpha is 0 from then on), so
            // a step with 255 alpha = 0 has room and changes nothing.  Fast path: nobody runs out of room in these four
            // steps — no bookkeeping beyond one mask AND per step.
            const float T0 = T;
            uint64_t allroom = ~0ull;
            float w4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float aT = a4[k] * T;                                   // 255 alpha T
                const float tT = __builtin_fmaf(aT, -kInv255, T);             // T (1 - alpha)
                const uint64_t room = __ballot(tT >= LSR_T_EPS);
                allroom &= room;
                w4[k] = __builtin_amdgcn_inverse_ballot_w64(room) ? aT : 0.0f;
                T = __builtin_fmaf(-kInv255, w4[k], T);                       // T -= w' / 255
            }
            if (allroom != ~0ull) {
                // some pixel stops inside this group of four (at most once per pixel and item): redo the steps from T0,
                // a stopped pixel keeping its T and contributing nothing afterwards
                T = T0;
                uint64_t alive = ~done, stop = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float aT = a4[k] * T;
                    const float tT = __builtin_fmaf(aT, -kInv255, T);
                    const uint64_t room = __ballot(tT >= LSR_T_EPS);
                    const uint64_t stopped = alive & ~room;
                    w4[k] = __builtin_amdgcn_inverse_ballot_w64(alive & room) ? aT : 0.0f;
                    T = __builtin_fmaf(-kInv255, w4[k], T);
                    alive &= room;
                    if (stopped) {
                        const uint32_t offk = s_list[i + k];
                        const uint32_t pos = base + 1u + (offk - wave_off) / (uint32_t)(kEnt * 16);
                        stop_pos = __builtin_amdgcn_inverse_ballot_w64(stopped) ? pos : stop_pos;
                        stop |= stopped;
                    }
                }
                pxx = __builtin_amdgcn_inverse_ballot_w64(stop) ? __builtin_nanf("") : pxx;
                done |= stop;
            }
            const float w = slot == 0 ? w4[0] : (slot == 1 ? w4[1] : (slot == 2 ? w4[2] : w4[3]));
#pragma unroll
            for (int c = 0; c < NCHP; ++c) acc[c] = __builtin_fmaf(pay[c], w, acc[c]);
            D = __builtin_fmaf(b.z, w, D);                                    // depth += (z / 255) w'
        }
        wave_lds_fence();
    }
    if (steep && lane == 0) p.item_flags[2 * (size_t)vt + half] = kItemFlagSteep;
    // the rows' partial sums -> the pixel's sums (the same fixed order in every row)
    D = rows_sum(D);
#pragma unroll
    for (int c = 0; c < NCHP; ++c) acc[c] = rows_sum(acc[c]);
    typedef const float __attribute__((address_space(4))) *kfloat_ptr;
    const kfloat_ptr vw = (kfloat_ptr)(p.views + (size_t)v * LSR_VIEW_FLOATS);
    if (inside && slot == 0) {
        const size_t pix = (size_t)py * p.W + (size_t)px;
        const size_t vp = (size_t)v * HW + pix;
        p.final_T[vp] = T;
        p.n_contrib[vp] = stop_pos ? stop_pos - 1u : hn;
        p.out_mask[vp] = 1.0f - T;
        p.out_depth[vp] = D;
        if (p.has_color) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                p.out_color[((size_t)v * 3 + c) * HW + pix] = __builtin_fmaf(T, vw[37 + c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NCHP; ++c)
            if (c >= coff && c - coff < p.C)
                p.out_feat[((size_t)v * p.C + (c - coff)) * HW + pix] = acc[c];
    }
}

// The launch for small view batches: the p.quad_items costliest half-tile items are rendered as eight SUB-BLOCK items
// each, the others as two ROW items each.  The launcher asks for all or none (see there); the mix is an experiment knob.
template <int NCHP, int WPB, bool REC = false>
__global__ void __launch_bounds__(LSR_WAVE * WPB, (REC && WPB == 12) ? 6 : 1)   // (the REC instances must stay within six waves per SIMD: 2 x 12 per CU)
k_render_fwd_small(RenderFwdParams p) {
    constexpr int kEnt = SmallWave<NCHP>::kEnt;
    struct Lds {
        float4 ent[WPB][LSR_WAVE + 1][kEnt];
        uint32_t list[WPB][4 * (LSR_WAVE + 1)];
    };
    __shared__ Lds s_lds;
    SmallWave<NCHP> w;
    w.lane = threadIdx.x & (LSR_WAVE - 1);
    const int lane = w.lane;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / LSR_WAVE);
    w.s_ent = s_lds.ent[wid];
    w.s_flat = s_lds.list[wid];
    w.ent_base = (const char *)&s_lds.ent[0][0][0];
    w.wave_off = (uint32_t)(wid * (LSR_WAVE + 1) * kEnt * 16);
    w.my_off = w.wave_off + (uint32_t)(lane * kEnt * 16);
    w.null_off = w.wave_off + (uint32_t)(LSR_WAVE * kEnt * 16);
    w.coff = p.has_color ? 3 : 0;
    w.HW = (size_t)p.H * p.W;
    if (lane < 4) w.s_flat[lane * (LSR_WAVE + 1) + LSR_WAVE] = w.null_off;     // the row items' four list terminators
    if (lane == 0) {
        w.s_ent[LSR_WAVE][0] = make_float4(__builtin_nanf(""), 0.0f, 0.0f, 0.0f);
        w.s_ent[LSR_WAVE][1] = make_float4(0.0f, 0.0f, 0.0f, -kInv255);
#pragma unroll
        for (int c4 = 0; c4 < NCHP / 4; ++c4) w.s_ent[LSR_WAVE][2 + c4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const uint32_t quad_end = 8u * p.quad_items;
    const uint32_t num_items = quad_end + 2u * (p.num_items - p.quad_items);

    const uint32_t simd_bins = (uint32_t)p.num_cus * 4u, slots = (uint32_t)p.num_cus * (uint32_t)p.waves_per_cu;
    const uint32_t vwave = (uint32_t)wid + (uint32_t)WPB * (blockIdx.x / (uint32_t)p.num_cus);
    const uint32_t bin = (blockIdx.x % (uint32_t)p.num_cus) * 4u + (vwave & 3u);
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
        if (qi == 0u && lane == 0) p.header_rw[kHdrFlagsValid] = 1u;
        if (qi < quad_end) {
            render_subblock_item<NCHP>(p, w, p.items[qi >> 3], (int)(qi & 7u));
        } else {
            const uint32_t r = qi - quad_end;
            render_row_item<NCHP, REC>(p, w, p.items[p.quad_items + (r >> 1)], (int)(r & 1u));
        }
    }
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
    p.num_cus = device_cus();
    // the costliest-first item list (written with the tile offsets) and the queue word
    p.items = (const uint32_t *)(geom + L.tile_order);
    p.num_items = 2u * (uint32_t)d.num_views * (uint32_t)p.T;
    p.queue = (uint32_t *)(geom + L.header) + kHdrQueueFwd;
    p.bin_queue = nullptr;
    p.header = (const uint32_t *)(geom + L.header); p.prio_pct = 0;
    p.views = in.views;
    p.rec = (const float4 *)(geom + L.rec); p.rec_f4 = L.rec_floats / 4;
    p.tile_start = (const uint32_t *)(geom + L.tile_start);
    p.half_count = (const uint32_t *)(geom + L.half_count);
    p.half_list = (const uint32_t *)(bin + B.half_list);
    p.half_list_rw = (uint32_t *)(const_cast<char *>(bin) + B.half_list);
    p.item_cost = nullptr;
    p.item_flags = (uint32_t *)(const_cast<char *>(geom) + L.item_flags);
    p.header_rw = (uint32_t *)(const_cast<char *>(geom) + L.header);
    p.ip = index_packing(d);
    p.out_color = out.color; p.out_feat = out.feature; p.out_mask = out.mask; p.out_depth = out.depth;
    p.final_T = (float *)(img + I.final_T); p.n_contrib = (uint32_t *)(img + I.n_contrib);
    const int nch = (p.has_color ? 3 : 0) + d.feat_channels;
    const int nchp = nch <= 4 ? 4 : (nch <= 8 ? 8 : (nch <= 12 ? 12 : 36));

    p.trace = nullptr;
    p.quad_items = 0;
    p.all_live = env_int("LSR_FWD_LIVE", 1) == 0 ? 1u : 0u;
#ifdef LSR_ENABLE_TRACE
    const int64_t max_items = 2 * (int64_t)p.T * d.num_views * 4;     // (x 4: the EMU instances' units)
    const char *trace_path = getenv("LSR_TRACE");
    if (trace_path) {
        (void)hipMalloc((void **)&p.trace, (size_t)max_items * 48);
        (void)hipMemsetAsync(p.trace, 0, (size_t)max_items * 48, s);
    }
#endif
    prof_begin(kStRenderFwd, s);
    // WPC = resident waves per CU = (workgroups per CU) x WPB: as many as registers and LDS slices allow.
    // LSR_FWD_VARIANT (development knob, read once) selects the alternatives measured in DESIGN.md.
#define LSR_RF(N, WPB, WPC)                                                                                \
    do {                                                                                                   \
        p.waves_per_cu = (WPC);                                                                            \
        hipLaunchKernelGGL((k_render_fwd<N, WPB>), dim3(p.num_cus * ((WPC) / (WPB))), dim3(LSR_WAVE * WPB), 0, s, p); \
    } while (0)
    // Small view batches (k_render_fwd_small): ROW items when two per half-tile item still fit the wave slots the two-pixel
    // launch would leave empty (a few views), SUB-BLOCK items when even eight per half-tile item fit (one view).
    // LSR_FWD_ROWS = 0 / 1 forces the choice of the kernel, LSR_FWD_QUAD = 0 / 1 no / only sub-block items.
    const int rows_knob = env_int("LSR_FWD_ROWS", -1), quad_knob = env_int("LSR_FWD_QUAD", -1);
    const uint64_t slots = (uint64_t)p.num_cus * 24u;
    // Forwards that a backward follows narrow the render lists for it while they composite: row items do it per row
    // (render_row_item REC), sub-block items cannot (eight waves share an entry's word and one view's backward gains little).
    // LSR_FWD_ROWREC = 0: the round-6a policy (3+ views of 256 x 256 take the half-tile RECORD kernel instead of row items).
    const bool for_bwd = (d.forward_flags & LSR_FWD_FOR_BACKWARD) != 0 && p.ip.all_bits == 0u && env_int("LSR_FWD_RECORD", 1) != 0;
    const bool row_rec = env_int("LSR_FWD_ROWREC", 1) != 0;
    const bool small = (nchp == 4 || nchp == 8) && (quad_knob == 1 || (rows_knob >= 0 ? rows_knob != 0 : (2ull * p.num_items <= slots && !(for_bwd && !row_rec && 4ull * p.num_items >= slots))));
    if (small) {
        p.waves_per_cu = 24;
        // Sub-block items for ALL items or none.  (Measured: the costliest (slots - 2 items) / 6 items as sub-block items next
        // to row items for the rest fill every slot, and with six waves per SIMD the sub-block items — the longest lists —
        // take twice as long as alone: V = 2 0.085 against 0.068 ms for row items only, V = 4 0.088 / 0.078, configs[3]
        // 0.160 / 0.136.  LSR_FWD_QUAD_ITEMS = n forces n leading items for experiments.)
        const int forced = env_int("LSR_FWD_QUAD_ITEMS", -1);
        p.quad_items = quad_knob == 0 ? 0u : ((quad_knob == 1 || 8ull * p.num_items <= slots) ? p.num_items : 0u);
        if (forced >= 0) p.quad_items = std::min<uint32_t>(p.num_items, (uint32_t)forced);
        const bool rec = for_bwd && row_rec && p.quad_items == 0u;
        if (nchp == 4) {
            if (rec) hipLaunchKernelGGL((k_render_fwd_small<4, 12, true>), dim3(p.num_cus * 2), dim3(LSR_WAVE * 12), 0, s, p);
            else hipLaunchKernelGGL((k_render_fwd_small<4, 12>), dim3(p.num_cus * 2), dim3(LSR_WAVE * 12), 0, s, p);
        } else {
            if (rec) hipLaunchKernelGGL((k_render_fwd_small<8, 12, true>), dim3(p.num_cus * 2), dim3(LSR_WAVE * 12), 0, s, p);
            else hipLaunchKernelGGL((k_render_fwd_small<8, 12>), dim3(p.num_cus * 2), dim3(LSR_WAVE * 12), 0, s, p);
        }
        prof_end(kStRenderFwd, s);
        return hipGetLastError();
    }
    const int variant = env_int("LSR_FWD_VARIANT", 0);
    // a forward that a backward will follow (lsr_dims::forward_flags) refines the lists' sub-block bits while it composites
    // (k_render_fwd RECORD; the entries must carry bits: scenes of up to 2^24 Gaussians).  LSR_FWD_RECORD=0 turns it off.
    const bool record = (d.forward_flags & LSR_FWD_FOR_BACKWARD) != 0 && p.ip.all_bits == 0u && env_int("LSR_FWD_RECORD", 1) != 0;
#define LSR_RFR(N, WPB, WPC)                                                                               \
    do {                                                                                                   \
        p.waves_per_cu = (WPC);                                                                            \
        hipLaunchKernelGGL((k_render_fwd<N, WPB, true>), dim3(p.num_cus * ((WPC) / (WPB))), dim3(LSR_WAVE * WPB), 0, s, p); \
    } while (0)
    // Issue priority by progress (render_backward.hip has the mechanism and the measurements): of the forward instances only the
    // list-narrowing 8-channel one — four waves per SIMD, like the backward — gains (configs[4]: 0.349 -> 0.320 ms at 18 %; the
    // plain 8-channel instance 0.298 either way; the 4-channel instances, six waves per SIMD, -1 % ... +8 %: profiles/r05_ab_knobs.md
    // section 12).  LSR_FWD_PRIO_PCT >= 0 forces a value for every instance.
    {
        const int prio_knob = env_int("LSR_FWD_PRIO_PCT", -1);
        p.prio_pct = prio_knob >= 0 ? prio_knob : ((record && nchp == 8) ? 18 : 0);
    }
    // Per-SIMD-bin item lists instead of the global queue (see the kernel) for the 4-channel half-tile instances while the
    // lists are short enough for an item to be a tenth of a SIMD's work.  Measured (profiles/r06_ab_knobs.md): 16 views x 300 k
    // 0.2015 -> 0.1965 ms, the list-narrowing instance 0.2224 -> 0.2082; a scene of opaque 1-10 px splats 0.175 -> 0.162; 10^6
    // Gaussians (6000-entry tile lists) +1 %, the 8-channel instances (configs[4]) +1.5 %: off there.  LSR_FWD_BINQ = 0 / 1 forces.
    {
        const int knob = env_int("LSR_FWD_BINQ", -1);
        const bool on = knob >= 0 ? knob != 0 : (nchp == 4 && num_pairs <= 3000 * (int64_t)d.num_views * p.T);
        if (on && 4 * p.num_cus <= (int)(kBinQueueBytes / 4)) p.bin_queue = (uint32_t *)(const_cast<char *>(geom) + L.bin_queue);
    }
    // the backward's work order from the RECORD forward's own iteration counts (k_order_items), when an item is a sizeable
    // part of a wave slot's work in the backward (more items than its 16 x CUs slots, at most kMaxReorderItems).  LSR_REORDER=0: off
    const bool reorder = record && (nchp == 8 || (nchp == 4 && variant == 0)) && p.num_items > 16u * (uint32_t)p.num_cus &&
                         p.num_items <= kMaxReorderItems && env_int("LSR_REORDER", 1) != 0;
    if (reorder) p.item_cost = (uint32_t *)(const_cast<char *>(geom) + L.item_cost);
    if (record && nchp == 4 && variant == 0) LSR_RFR(4, 12, 24);
    else if (record && nchp == 8) LSR_RFR(8, 16, 16);
    else
    if (nchp == 4 && (variant == 5 || variant == 6)) {   // timing emulation of list splitting (see the kernel's EMU parameter)
        p.waves_per_cu = 24;
        if (variant == 5) hipLaunchKernelGGL((k_render_fwd<4, 12, false, 2>), dim3(p.num_cus * 2), dim3(LSR_WAVE * 12), 0, s, p);
        else hipLaunchKernelGGL((k_render_fwd<4, 12, false, 4>), dim3(p.num_cus * 2), dim3(LSR_WAVE * 12), 0, s, p);
    } else
    if (nchp == 4) { if (variant == 2) LSR_RF(4, 16, 16); else if (variant == 3) LSR_RF(4, 14, 28); else LSR_RF(4, 12, 24); }   // 2 x 12 waves per CU: 0.245 vs 0.281 ms per 16 views with 16
    // 7 / 8 channels (colour + 4 latent channels: the reference's configs[3] / [4] payload; 7.3 KB of LDS and 90 VGPRs per
    // wave).  Round 4 measured 2 x 10 waves per CU (what LDS and registers allow at most) and 2 x 8: configs[3] 0.1426 /
    // 0.1367 ms against 0.1377 for one 16-wave workgroup, configs[4] 0.3019 / 0.3006 / 0.3017 — no gain, the shape stays.
    else if (nchp == 8) LSR_RF(8, 16, 16);
    else if (nchp == 12) LSR_RF(12, 12, 12);
    else LSR_RF(36, 4, 8);
#undef LSR_RF
#undef LSR_RFR
    if (reorder)
        hipLaunchKernelGGL(k_order_items, dim3(1), dim3(kOrderThreads), 0, s, (const uint32_t *)p.item_cost,
                           (uint32_t *)(const_cast<char *>(geom) + L.tile_order2), p.header_rw, p.num_items);
    prof_end(kStRenderFwd, s);
#ifdef LSR_ENABLE_TRACE
    if (trace_path) {  // debug only: dump per-item timing of this launch
        std::vector<unsigned long long> host((size_t)max_items * 6);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(host.data(), p.trace, host.size() * 8, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(trace_path, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
        (void)hipFree(p.trace);
    }
#endif
    return hipGetLastError();
}

}  // namespace lsr
