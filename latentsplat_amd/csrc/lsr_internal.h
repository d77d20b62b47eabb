// lsr_internal.h — shared device/host declarations of the lsr HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsr_rasterizer.h"

// gfx950 (MI355X) only, by construction: wave64 lane maps, DPP row operations, the packed-f32 inline asm without the
// compiler's wait states (tools/microbench/pk_hazard.hip measured the forwarding on this part only) and the LDS sizing all
// assume it.  A device pass for any other target is a build error, not a slower fallback.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "liblsr_hip is written for gfx950 (MI355X / CDNA4) only: build with --offload-arch=gfx950"
#endif
#define LSR_WAVE 64
#define LSR_NEAR_CULL 0.2f
#define LSR_LOWPASS 0.3f
#define LSR_ALPHA_MAX 0.99f
#define LSR_ALPHA_MIN (1.0f / 255.0f)
#define LSR_T_EPS 0.0001f

namespace lsr {

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct GeomLayout {
    size_t header, rec, bin, tile_count, tile_start, tile_cursor, item_flags, bin_queue, item_cost, tile_order2, tile_order, half_count, sh_clamp, seg_keys, total;
    int rec_floats;
    uint32_t seg_cap;      // keys per (view, tile) segment of the single-pass binning; 0 = two-phase binning (k_scatter)
};
struct ImgLayout {
    size_t final_T, n_contrib, total;
};
struct BinLayout {
    size_t keys, point_list, half_list, tmp, total;
};
// One packed gradient record per (view, Gaussian), accumulated by the compositing backward.  With
// u = opacity * G * dL/dalpha per (pixel, Gaussian) and d = mean_pix - pixel:
//   [0] sum u dx  [1] sum u dy  [2] sum u dx^2  [3] sum u dx dy  [4] sum u dy^2  [5] sum u  [6] d/dz  [7] -
//   [8 + c] d/d payload channel c  (rgb first when colour is rendered, then features)
// (k_preprocess_bwd turns the moments into dL/d mean, conic, opacity: the conic and the opacity
// are constant over a Gaussian's pixels.)  16 floats (one 64-byte line) for <= 8 payload
// channels, 32 for <= 12, 64 beyond.
// per-SIMD-bin queue heads of the compositing kernels (experiment knobs LSR_FWD_BINQ / LSR_BWD_BINQ): room for 4096 bins
constexpr size_t kBinQueueBytes = 4096 * 4;
// Work items are handed to the compositing waves costliest first.  The tile scan orders them by the tile's pair count (all it
// has).  A forward that a backward follows knows better when it is done: the RECORD compositing kernel has counted the lock-step
// iterations it spent on every item, and k_order_items (render_forward.hip, one workgroup behind it) re-orders the items by that
// count, in kCostClasses classes of two iterations, for the compositing BACKWARD: 0.570 -> 0.535 ms at 16 views x 300 k for 5 us
// of ordering (profiles/r06_ab_knobs.md section 6).
constexpr int kCostClasses = 1024;
constexpr uint32_t kMaxReorderItems = 16384;      // calls with more items keep the tile scan's order (many items per wave slot: balanced anyway)
struct GradLayout {
    size_t rec, fixed, total;
    int rec_floats;
};
inline int grad_rec_floats(const lsr_dims &d);

inline int tiles_x(const lsr_dims &d) { return (d.width + LSR_TILE - 1) / LSR_TILE; }
inline int tiles_y(const lsr_dims &d) { return (d.height + LSR_TILE - 1) / LSR_TILE; }
inline int64_t num_tiles(const lsr_dims &d) { return (int64_t)tiles_x(d) * tiles_y(d); }

// Screen-space record of one (view, Gaussian), written by k_preprocess and gathered (one 64-byte
// line for <= 8 payload channels) by the compositing kernels:
//   [0] x_pix [1] y_pix [2] conicA [3] conicB [4] conicC [5] opacity [6] view z [7] unused (0)
//   [8 + c] payload channel c: rgb first when colour is rendered, then the feature channels; zero padded.
// The packed gradient record of the backward pass (GradLayout) uses the same slot numbering.
inline int rec_floats(const lsr_dims &d) {
    const int nch = (d.color_mode != LSR_COLOR_NONE ? 3 : 0) + d.feat_channels;
    return nch <= 8 ? 16 : (nch <= 12 ? 32 : 64);   // always >= 8 + the compositing kernels' padded channel count
}
// What the binning stage needs of a (view, Gaussian), kept dense so k_scatter streams it: the tile
// rectangle [minx, miny, maxx, maxy) and the view depth (sort key bits; 0 = culled).  Written for ALL
// V*G slots by k_preprocess and read back once by k_scatter — both stages are bandwidth-bound, so the
// record is 12 bytes whenever the tile grid fits byte coordinates (images up to 4080 px a side) and
// 16 bytes otherwise.
struct BinRec {            // narrow form (12 bytes)
    uint32_t rect;         // minx | miny << 8 | maxx << 16 | maxy << 24
    float depth;
    uint32_t span;         // footprint span in 4-pixel cells (below)
};
struct BinRecWide {
    ushort4 rect;
    float depth;
    uint32_t span;
};
inline bool narrow_bins(const lsr_dims &d) { return tiles_x(d) <= 255 && tiles_y(d) <= 255; }
inline size_t bin_stride(const lsr_dims &d) { return narrow_bins(d) ? sizeof(BinRec) : sizeof(BinRecWide); }

// Footprint span of a (view, Gaussian), 4 bytes of its bin record: the axis-aligned bounding box of
// { alpha >= 1/255 } (lsr_blend.h footprint_cells: conservative, 0.1 % + 0.05 px of slack) in units of
// 4-pixel CELLS (cell c = pixel centres 4c .. 4c+3 — the grid of the compositing kernels' 4x4 sub-blocks),
// relative to the first cell of the tile rectangle and saturated to a byte:
//   x0 | x1 << 8 | y0 << 16 | y1 << 24;   first / last reached cell; a last cell of 255 means "255 or
//   beyond"; x0 > x1 (kSpanNone) = reaches no pixel at all.
// k_scatter turns it into the 8-bit sub-block code of every (Gaussian, tile) pair, carried in the low byte of
// the sort key (below), from which k_sort_tiles builds the two half-tile render lists.
constexpr uint32_t kSpanNone = 0x00010001u;   // x0 = 1 > x1 = 0
constexpr uint32_t kSpanAll = 0xFF00FF00u;    // conic not trustworthy: every cell
// Sort key of a (Gaussian, tile) pair: depth bits << 32 | index << 8 | code, code = c0 | c1 << 2 | r0 << 4 | r1 << 6:
// the columns c0..c1 and rows r0..r1 (0..3) of the tile's 4x4-pixel sub-blocks the pair can reach
// (c0 > c1: none).  Indices are distinct inside a tile, so the order is still the published
// (depth, index) order.  Needs index < 2^24 (kMaxGaussians per scene; lsr_* calls return LSR_EUNSUPPORTED beyond).
constexpr int kKeyIndexShift = 8;
constexpr uint32_t kCodeNone = 0x11u;   // c0 = 1 > c1 = 0, r0 = 1 > r1 = 0

// ---- single-pass binning (round 5): fixed-capacity key segments ----
// Rounds 1-4 binned in two phases: the projection kernel counted pairs per tile and wrote a 12-byte binning record per
// (view, Gaussian); after the tile scan (and, in the synchronous forward, the host's read-back of the pair count that
// sizes the binning workspace) k_scatter read the records back and wrote the sort keys into exactly sized tile segments.
// Now every (view, tile) owns a FIXED-capacity key segment at the end of the geometry workspace, and the projection
// kernel writes the keys itself: a workgroup reserves its slots with the one global atomic per (workgroup, tile) it
// already spent on the pair count (now a returning one) and emits the keys of its 2048 (Gaussian, view) items before it
// retires — no k_scatter launch, no second machine-wide pass over the binning records (the workgroup re-reads its own
// 24 KB of them while they are still in its L2).  The tile scan still produces the exact offsets, so k_sort_tiles
// writes the canonical point_list and the half-tile lists exactly where they were.
// A tile with more pairs than a segment holds keeps only its first `cap` keys there (the surplus lands on the segment's
// last slot); such tiles are binned a second time by k_scatter's overfull-tiles-only instance into their exact
// segments of the binning workspace, and k_sort_tiles reads them from there (binning.hip): correct for any list length,
// slower only for the scenes that need it.
// segment_capacity(d) (api.hip) is a pure function of the dims and the LSR_SEGMENTS / LSR_SEG_BUDGET_MB knobs: 0 = the
// call takes the two-phase path.  The segments sit BEHIND everything else in the layout, so no other offset depends
// on it.
uint32_t segment_capacity(const lsr_dims &d);
struct SegOut {            // kernel argument of the projection kernels
    uint64_t *keys;        // [V*T][cap] (nullptr: two-phase binning)
    uint32_t cap;
    uint32_t key_shift;    // IndexPacking::key_shift
    uint32_t ablate;       // LSR_SEG_ABLATE (timing experiments only, wrong results): 1 no key stores, 2 no emission pass
};

inline GeomLayout geom_layout(const lsr_dims &d) {
    GeomLayout L;
    const size_t VG = (size_t)d.num_views * (size_t)d.num_gaussians;
    const size_t VT = (size_t)d.num_views * (size_t)num_tiles(d);
    size_t o = 0;
    L.rec_floats = rec_floats(d);
    L.rec = o; o = align_up(o + VG * (size_t)L.rec_floats * 4);
    L.bin = o; o = align_up(o + VG * bin_stride(d));
    L.header = o; o += 256;                        // header .. tile_cursor are cleared by ONE memset per forward
    L.tile_count = o; o = align_up(o + VT * 4);
    L.tile_cursor = o; o = align_up(o + VT * 4);   // adjacent to tile_count: one memset clears both
    // one word per half-tile work item, inside the cleared range: kItemFlagSteep = the forward compositing kernel staged an
    // entry of opacity >= kSteepOpacity for this item, i.e. a pixel may have blended an alpha close to the 0.99 clamp; the
    // compositing backward then walks the item's list BACK TO FRONT (render_backward.hip)
    L.item_flags = o; o = align_up(o + 2 * VT * 4);
    L.bin_queue = o; o += kBinQueueBytes;          // (cleared with the rest of the range)
    L.tile_start = o; o = align_up(o + (VT + 1) * 4);
    L.tile_order = o; o = align_up(o + 2 * VT * 4);   // work items (see kItem*), costliest first
    L.item_cost = o; o = align_up(o + 2 * VT * 4);    // per item (2 (view T + tile) + half): lock-step iterations the RECORD forward spent on it
    L.tile_order2 = o; o = align_up(o + 2 * VT * 4);  // the work items again, ordered by that cost for the backward (header word kHdrOrder2Valid)
    L.half_count = o; o = align_up(o + 2 * VT * 4);   // entries of the two half-tile render lists of every (view, tile)
    L.sh_clamp = o; o = align_up(o + VG);             // per (view, Gaussian): colour channels clamped at 0 (sh.hip)
    L.seg_cap = segment_capacity(d);
    L.seg_keys = o; o = align_up(o + VT * (size_t)L.seg_cap * 8);
    L.total = o;
    return L;
}
inline ImgLayout img_layout(const lsr_dims &d) {
    ImgLayout L;
    const size_t n = (size_t)d.num_views * d.height * d.width;
    L.final_T = 0;
    L.n_contrib = align_up(n * 4);
    L.total = L.n_contrib + align_up(n * 4);
    return L;
}
// Longest per-tile list the LDS sort handles; longer lists go through the global-memory path.
constexpr int kSortLdsMax = 16384;
inline BinLayout bin_layout(const lsr_dims &d, int64_t num_pairs, int32_t max_tile_pairs) {
    BinLayout L;
    const size_t P = (size_t)(num_pairs > 0 ? num_pairs : 1);
    L.keys = 0;
    L.point_list = align_up(P * 8);
    // Half-tile render lists (k_sort_tiles -> compositing kernels): the tile whose canonical list is
    // point_list[s, s + n) owns half_list[2 s, 2 s + 2 n); the list of its upper (h = 0: pixel rows 0-7) / lower
    // (h = 1) half starts at 2 s + h n and holds half_count[2 (view T + tile) + h] entries
    // `index | sub-block bits << 24` (bit 4 r + c: the entry can reach the 4x4-pixel sub-block (c, r) of the half).
    L.half_list = L.point_list + align_up(P * 4);
    L.tmp = L.half_list + align_up(P * 8);
    L.total = L.tmp + (max_tile_pairs > kSortLdsMax ? align_up(P * 8) : 0);
    return L;
}
inline int grad_rec_floats(const lsr_dims &d) { return rec_floats(d); }
// LSR_DETERMINISTIC=1 (debugging aid, read once per process): the compositing backward's cross-tile
// sums go through 64-bit FIXED-POINT integer atomics (order independent => bitwise reproducible
// gradients) in a second record array behind the float records, converted back by one extra kernel.
bool deterministic_backward();
bool projection_contraction();                      // lsr_set_projection_contraction (api.hip): fused multiply-adds in k_preprocess
constexpr double kFixedPointScale = 1073741824.0;   // 2^30: 9.3e-10 resolution, +-8.6e9 range per record slot
inline GradLayout grad_layout(const lsr_dims &d) {
    GradLayout L;
    const size_t VG = (size_t)d.num_views * (size_t)d.num_gaussians;
    L.rec_floats = grad_rec_floats(d);
    L.rec = 0;
    // behind the float records: 512 zeroed bytes — the compositing backward's work-queue word (fixed - 512) and one all-zero
    // record line (fixed - 256) that the per-Gaussian backward kernels read in place of the records of CULLED (view,
    // Gaussian) slots: one cached line instead of a third of the record array
    L.fixed = align_up(VG * (size_t)L.rec_floats * 4 + 512 + kBinQueueBytes);   // [V*G][rec_floats] int64, deterministic mode only
    L.total = L.fixed + (deterministic_backward() ? align_up(VG * (size_t)L.rec_floats * 8) : 0);
    return L;
}

// Compositing work items: one wave renders one HALF (16 x 8 pixels) of one (view, tile), walking that half's
// render list (BinLayout::half_list: only the entries whose alpha >= 1/255 footprint box reaches the half,
// each with the 8-bit mask of the half's 4x4 sub-blocks it can reach).
//   item = (view*T + tile) | half << 28.  k_tile_scan emits the two items of every tile, longest
//   canonical list first (the work-queue order of both compositing kernels).
// Rounds 1-2 walked the canonical tile list per item and re-derived the masks per staged entry (a quarter of
// the staged pairs reached no pixel at all).  With per-half lists an item stages exactly what it evaluates.

// Input slice a view reads: its own (per-view strides), its group's, or the shared one (stride 0).
__host__ __device__ inline int input_slice(const lsr_dims &d, int v) { return d.views_per_group > 1 ? v / d.views_per_group : v; }

constexpr uint32_t kItemFlagSteep = 1u;
// opacity from which an entry marks its item: 1 / (1 - alpha) — the factor by which the forward-order backward amplifies
// the rounding of the pixel's total — can reach 4 from here on (100 at the clamp)
constexpr float kSteepOpacity = 0.75f;
constexpr uint32_t kItemTileMask = 0x0FFFFFFFu;
constexpr int kItemHalfShift = 28;
constexpr uint32_t kListIndexMask = 0x00FFFFFFu;   // half_list entry = Gaussian index | sub-block bits << 24
constexpr int kListBitsShift = 24;
constexpr int kMaxGaussians = 1 << 24;             // index field of the sort keys (index << 8 | code) and of the list entries
// Scenes with MORE Gaussians than that (round 4; round 3 rejected them) keep the full 32-bit index in the key's low word
// and in the list entries and give up the footprint culling instead: every entry of a tile's list goes to both half lists
// and is evaluated on every sub-block — the published algorithm's work, correct, slower.  The kernels take the two
// constants below as launch arguments (uniform; two scalar operations per staged entry).
struct IndexPacking {
    uint32_t key_shift;      // sort key low word = index << key_shift | code
    uint32_t index_mask;     // list entry & index_mask = Gaussian index
    uint32_t all_bits;       // OR-ed onto a list entry's 8 sub-block bits (0xFF when the entries carry none)
};
// LSR_FWD_REACHED_ONLY: the binning drops the pairs whose footprint code is kCodeNone (needs the codes: up to 2^24 Gaussians)
__host__ __device__ inline bool reached_only(const lsr_dims &d) { return (d.forward_flags & LSR_FWD_REACHED_ONLY) != 0 && d.num_gaussians <= (1 << 24); }
inline IndexPacking index_packing(const lsr_dims &d) {
    return d.num_gaussians > kMaxGaussians ? IndexPacking{0u, 0xFFFFFFFFu, 0xFFu} : IndexPacking{(uint32_t)kKeyIndexShift, kListIndexMask, 0u};
}
// The compositing kernels run one 16-wave workgroup (4 waves per SIMD) per compute unit; the number
// of CUs is queried per device (api.hip), so a partitioned (CPX) or binned part gets its own static
// assignment.  wave slots = CUs x SIMDs x resident compositing waves per SIMD.
int device_cus();                                   // multiProcessorCount of the current device (cached per device)
inline int wave_slots(int cus) { return cus * 4 * 4; }   // (at 4 resident compositing waves per SIMD)
// Environment knobs are development aids; each is read ONCE per process (never on the launch path).
int env_int(const char *name, int fallback);        // api.hip: latched on first use
// header words of the geometry workspace (kHdrQueueFwd: work-queue head of the forward compositing kernel)
enum { kHdrPairs = 0, kHdrMaxTile = 1, kHdrNumItems = 3, kHdrOverflow = 4, kHdrPreDone = 5,
       kHdrFlagsValid = 6 /* the forward compositing kernel of this call filled in GeomLayout::item_flags */,
       kHdrQueueFwd = 8, kHdrOrder2Valid = 9 /* GeomLayout::tile_order2 holds the work items ordered by true cost */,
       kHdrLongTiles = 16 /* + 0, + 1: number of tiles beyond the first sort tier / beyond the second */ };

// ---- optional per-stage hipEvent timing (api.hip); no-ops unless lsr_profile_enable(1) ----
enum Stage { kStPreprocess = 0, kStTileScan, kStScatter, kStSort, kStRenderFwd, kStRenderBwd, kStPreprocessBwd, kStShFwd, kStShBwd, kStAdapterFwd, kStAdapterBwd, kStLatentFwd, kStLatentBwd, kNumStages };
void prof_begin(int stage, hipStream_t s);
void prof_end(int stage, hipStream_t s);
void note_hip_error(int hip_error);   // what lsr_last_hip_error() returns for this thread

// Workspace clears are kernels, not hipMemsetAsync: captured memset nodes were observed not to clear on
// back-to-back hipGraph replays (ROCm 7.2), which the latency mode relies on.
hipError_t launch_clear(void *ptr, size_t bytes, hipStream_t s);

// ---- stage launchers (defined one per .hip file) ----
// The tile scan (tile offsets, header, compositing work items, the host's two numbers: lsr_tile_scan.h) runs in the
// LAST workgroup of k_preprocess whenever the V*T counts fit its LDS staging array (fold_tile_scan: up to kFoldTiles);
// larger calls launch k_tile_scan behind it.
//   host_words: device pointer to three mapped host words (pair count, longest list, sequence number) or nullptr;
//   capacity  : pairs the binning workspace can hold (UINT32_MAX = exact sizing after the host read-back).
constexpr int kPreThreadsScan = 256, kFoldTiles = 4096;   // workgroup of k_preprocess; most (view, tile) counts its last workgroup scans
struct FoldedScan {
    int enabled;
    uint32_t *host_words; uint32_t host_seq;
    uint32_t capacity;
    uint32_t *tile_start, *tile_order;     // filled in by launch_preprocess
};
inline bool fold_tile_scan(const lsr_dims &d) {
    return d.num_gaussians > 0 && (int64_t)d.num_views * num_tiles(d) <= kFoldTiles && env_int("LSR_FOLD_SCAN", 1) != 0;
}
// seg: single-pass binning — the kernel writes the sort keys into the geometry workspace's tile segments (see
// segment_capacity); the caller then skips k_scatter (launch_binning's `seg`)
hipError_t launch_preprocess(const lsr_dims &d, const lsr_inputs &in, char *geom, int32_t *radii, const FoldedScan &fs,
                             bool seg, hipStream_t s);
hipError_t launch_tile_scan(const lsr_dims &d, char *geom, uint32_t *host_words, uint32_t host_seq, uint32_t pair_capacity, hipStream_t s);
hipError_t launch_pack_view(const float *viewmatrix, const float *projmatrix, const float *campos, const float *bg,
                            float tanfovx, float tanfovy, const float *tanfovx_dev, const float *tanfovy_dev, float *out,
                            hipStream_t s);
hipError_t launch_build_views(int V, const float *extrinsics, const float *intrinsics, const float *near,
                              const float *far, const float *bg, int bg_stride, int scale_invariant,
                              float *out, hipStream_t s);
// Stages that sum over views (SH kernels, geometry backward) work per view group (lsr_dims::views_per_group:
// b scenes x n views in one call).  They take the WHOLE call's dims / inputs and run all groups in ONE launch,
// the group in blockIdx.y: inside the kernel a group looks like a call of its own n views with shared inputs
// (strides 0), its pointers advanced by the group's slices (GroupStrides).
struct GroupStrides {     // element offsets between consecutive groups
    int64_t views, means, cov, opac, color, feat;   // inputs (and the matching gradient outputs)
    int64_t slots;                                   // (view, Gaussian) slots: views_per_group * G
};
inline int num_view_groups(const lsr_dims &d) { return d.views_per_group > 1 ? d.num_views / d.views_per_group : 1; }
inline GroupStrides group_strides(const lsr_dims &d) {
    GroupStrides g{};
    if (d.views_per_group > 1) {
        g.views = (int64_t)d.views_per_group * LSR_VIEW_FLOATS;
        g.means = d.vs_means; g.cov = d.vs_cov; g.opac = d.vs_opac; g.color = d.vs_color; g.feat = d.vs_feat;
        g.slots = (int64_t)d.views_per_group * d.num_gaussians;
    }
    return g;
}
// dims of ONE group of a view-group call (the whole call when there are no groups)
inline lsr_dims group_dims(const lsr_dims &d) {
    lsr_dims ds = d;
    if (d.views_per_group > 1) {
        ds.num_views = d.views_per_group; ds.views_per_group = 0;
        ds.vs_means = ds.vs_cov = ds.vs_opac = ds.vs_color = ds.vs_feat = 0;
    }
    return ds;
}
hipError_t launch_sh_forward(const lsr_dims &d, const lsr_inputs &in, char *geom, hipStream_t s);
// Projection + SH payload as one kernel (sh.hip k_preprocess_sh) for calls whose payload is harmonics only and whose
// views share their inputs; replaces launch_preprocess + launch_sh_forward when fused_preprocess_sh(d).
bool fused_preprocess_sh(const lsr_dims &d);
// (seg: single-pass binning, as launch_preprocess; segment_capacity(d) > 0 only when fused_segments_fit(d))
hipError_t launch_preprocess_sh(const lsr_dims &d, const lsr_inputs &in, char *geom, int32_t *radii, const FoldedScan &fs, bool seg, hipStream_t s);
bool fused_segments_fit(const lsr_dims &d);   // sh.hip: the fused kernel's LDS holds the single-pass binning's arrays for these dims
hipError_t launch_sh_backward(const lsr_dims &d, const lsr_inputs &in, const char *geom, const char *grad,
                              const lsr_in_grads &gin, hipStream_t s);
// device_counts: the pair count / longest list are NOT known on the host (no-sync forward):
// `num_pairs` is then the workspace capacity and `max_tile_pairs` only a hint for the sort variant
hipError_t launch_binning(const lsr_dims &d, char *geom, char *bin, int64_t num_pairs,
                          int32_t max_tile_pairs, const int32_t *radii, hipStream_t s, bool device_counts, bool seg,
                          bool speculative = false);
hipError_t launch_render_forward(const lsr_dims &d, const lsr_inputs &in, const char *geom,
                                 const char *bin, int64_t num_pairs, char *img, const lsr_outputs &out,
                                 hipStream_t s);
// Zeroes what the compositing backward accumulates into: the gradient records, the queue / zero-line slack and, in
// deterministic mode, the fixed-point records.
hipError_t launch_clear_grad(const lsr_dims &d, const int32_t *radii, char *grad, hipStream_t s);
hipError_t launch_render_backward(const lsr_dims &d, const lsr_inputs &in, const char *geom,
                                  const char *bin, int64_t num_pairs, const char *img,
                                  const lsr_outputs &fwd, const lsr_out_grads &gout, char *grad,
                                  const lsr_in_grads &gin, hipStream_t s);
hipError_t launch_preprocess_backward(const lsr_dims &d, const lsr_inputs &in, const char *geom,
                                      const int32_t *radii, const char *grad,
                                      const lsr_in_grads &gin, hipStream_t s);

}  // namespace lsr
