/*
 * lsr_rasterizer.h — C ABI of the MI355X-native latentSplat rasterizer ("lsr").
 *
 * This is the drop-in boundary for the ONE hot path named by BASELINE.json:north_star: the
 * Gaussian-splat rasterizer behind `diff_gaussian_rasterization.GaussianRasterizer`.
 *
 * What it replaces in the reference (all paths relative to /root/reference):
 *   - the pybind entry points `_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward` of the
 *     external package `git+https://github.com/Chrixtar/latent-gaussian-rasterization`
 *     (requirements.txt:33), which the reference reaches through
 *     `GaussianRasterizer(settings)(means3D=..., means2D=..., shs=..., colors_precomp=...,
 *     features=..., opacities=..., cov3D_precomp=...)` at
 *     src/model/decoder/cuda_splatting.py:132-158 (perspective) and :257-283 (orthographic).
 *   - lsr_forward_prepare + lsr_forward_render  <->  `_C.rasterize_gaussians`
 *     (forward of the autograd Function called at cuda_splatting.py:150-158);
 *   - lsr_backward                              <->  `_C.rasterize_gaussians_backward`
 *     (reached from `self.manual_backward(...)`, src/model/model_wrapper.py:440).
 *
 * Differences from the per-view reference call, by design (MI355X-first):
 *   - one call renders `num_views` views; every per-Gaussian input carries a per-view element
 *     stride, and stride 0 means "one scene shared by all views" (removes the v-fold `repeat`
 *     at src/model/decoder/decoder_splatting_cuda.py:71-87);
 *   - camera parameters live in a device-side view table, so no host<->device sync is needed to
 *     fetch tan(fov) (removes the `.item()` at cuda_splatting.py:135-136);
 *   - scratch memory is provided by the caller (two-phase: query size, then run), so the library
 *     is allocator-agnostic; with PyTorch it comes from the caching allocator.
 *
 * Plain C: pointers and sizes only, no torch / HIP types in the signatures.  All data pointers
 * are DEVICE pointers unless the name ends in `_host`.  All functions return 0 on success or a
 * negative LSR_E* code; nothing here throws.  The library never allocates device memory (its one
 * allocation is a 64-byte pinned host buffer per calling thread, through which lsr_forward_prepare
 * receives the pair count without a copy command).
 */
#ifndef LSR_RASTERIZER_H
#define LSR_RASTERIZER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSR_ABI_VERSION 10
#define LSR_TILE 16            /* tile edge in pixels (16x16 = the published algorithm's tile) */
#define LSR_MAX_FEAT_CHANNELS 32
#define LSR_MAX_SH_DEGREE 4

/* Per-view camera record, LSR_VIEW_FLOATS consecutive floats:
 *   [0..15]  viewmatrix  — memory of `settings.viewmatrix` (row-major tensor holding the
 *                          transposed world->view matrix, cuda_splatting.py:116,139)
 *   [16..31] projmatrix  — memory of `settings.projmatrix` (transposed world->clip, :117,140)
 *   [32..34] campos      — `settings.campos` (:142)
 *   [35]     tanfovx, [36] tanfovy   (:135-136)
 *   [37..39] bg          — `settings.bg` (3 entries; feature background is 0) (:137)
 *   [40]     scene scale — means3D are multiplied by it and covariances by its square before
 *                          projection (the `1/near` scale invariance of cuda_splatting.py:75-82
 *                          applied in-kernel; 1.0 when the caller already scaled its inputs)
 *   [41..43] reserved (0)                                                                      */
#define LSR_VIEW_FLOATS 44

enum { LSR_COLOR_NONE = 0, LSR_COLOR_SH = 1, LSR_COLOR_PRECOMP = 2 };
enum { LSR_FEAT_DIRECT = 0, LSR_FEAT_SH = 1 };
enum { LSR_SH_AXES_3DGS = 0, LSR_SH_AXES_REFERENCE = 1 };

enum {
    LSR_OK = 0,
    LSR_EINVAL = -1,     /* bad dimensions / argument combination */
    LSR_ENULL = -2,      /* required pointer is NULL */
    LSR_ELAUNCH = -3,    /* HIP runtime reported an error (see lsr_last_hip_error) */
    LSR_ECAPACITY = -4,  /* num_pairs smaller than what prepare() counted */
    LSR_EUNSUPPORTED = -5
};

typedef void *lsr_stream_t; /* a hipStream_t */

typedef struct lsr_dims {
    int32_t num_views;      /* V >= 1 */
    int32_t num_gaussians;  /* G >= 0 */
    int32_t height, width;  /* image size in pixels */
    int32_t feat_channels;  /* C: 0 = no `features=` input (feature_map is None) */
    int32_t color_mode;     /* LSR_COLOR_* : none / `shs=` / `colors_precomp=` */
    int32_t sh_degree;      /* `settings.sh_degree`, 0..4 */
    int32_t sh_coeffs;      /* K = shs.shape[1] >= (sh_degree+1)^2 */
    /* element strides between consecutive views (or view groups, see views_per_group);
     * 0 = shared by all views */
    int64_t vs_means;       /* means3D   (G,3)   */
    int64_t vs_cov;         /* cov3D_precomp (G,6): xx,xy,xz,yy,yz,zz (cuda_splatting.py:148,157) */
    int64_t vs_opac;        /* opacities (G,1)   */
    int64_t vs_color;       /* shs (G,K,3) or colors_precomp (G,3) */
    int64_t vs_feat;        /* features  (G,C) or feature SH coefficients (G,C,Kf) */
    /* --- fused scene-level inputs (what the reference computes in PyTorch before the call) --- */
    int32_t cov_elems;      /* 6: packed upper triangle (the reference boundary); 9: row-major 3x3
                               `gaussian_covariances`, upper triangle read, gradient written to it */
    int32_t feat_mode;      /* LSR_FEAT_DIRECT: `features` are per-Gaussian channel values (boundary);
                               LSR_FEAT_SH: `features` holds latent SH coefficients (G,C,Kf) and the
                               kernel evaluates 0.5 + eval_sh(dir) itself (cuda_splatting.py:94-101) */
    int32_t feat_sh_degree; /* 0..2 when feat_mode == LSR_FEAT_SH */
    int32_t feat_sh_coeffs; /* Kf >= (feat_sh_degree+1)^2, C*Kf <= 120 */
    int32_t color_sh_channel_major; /* 0: shs (G,K,3) as the reference hands them to the rasterizer;
                               1: (G,3,K) = `gaussian_color_sh_coefficients` as stored (skips the
                               `rearrange(...).contiguous()` copy of cuda_splatting.py:91) */
    int32_t views_per_group; /* 0 or 1: a non-zero stride advances per view.  n > 1: consecutive blocks
                               of n views form a group that shares its inputs, the strides advance
                               per GROUP (b scenes x n views in one call: what
                               DecoderSplattingCUDA.forward receives); num_views % n == 0 and every
                               per-Gaussian input must then be strided (!= 0).  Gradients of strided
                               inputs have one slice per group, summed over the group's views. */
    int32_t color_sh_convention; /* axis convention of the COLOUR SH basis at degree >= 1 (ABI v4):
                               LSR_SH_AXES_3DGS (0, default): the published 3DGS rasterizer's basis,
                                 l=1 terms -C1*y, +C1*z, -C1*x of the unit direction (x,y,z);
                               LSR_SH_AXES_REFERENCE (1): the reference's own eval_sh naming
                                 (src/misc/sh_utils.py:62-65: -C1*x, +C1*y, -C1*z), i.e. the same
                                 polynomials evaluated at (z,x,y) — what the fused latent-feature SH
                                 path always uses, and what the encoder's rotate_sh (e3nn Wigner-D)
                                 is consistent with.  Which of the two the external CUDA fork uses for
                                 colour cannot be determined offline (SURVEY.md Appendix A.4 [UNK];
                                 tools/dump_fork_vectors.py case fork_probe_sh_axes decides it). */
    int32_t forward_flags;  /* (ABI v8) bit 0, LSR_FWD_FOR_BACKWARD: an lsr_backward of this forward will follow.  The
                               forward compositing kernel then records on which 4x4-pixel sub-blocks every list entry
                               actually contributed and narrows the render lists' sub-block bits to those (identical
                               images; the backward evaluates ~14 % fewer (entry, sub-block) pairs, the forward pays
                               one LDS atomic per loop iteration).  Ignored by lsr_backward itself.  Coverage: the 4- and
                               8-channel half-tile instances and the row items of small view batches (2-6 views of 256 x 256);
                               the sub-block items of single-view-sized calls write only the item flags of
                               lsr_layout.geom_item_flags; forwards of 9-32 payload channels and of scenes beyond
                               2^24 Gaussians ignore the bit — identical results, the backward then evaluates the
                               footprint-box masks and walks every item back to front.
                               (ABI v9) bit 1, LSR_FWD_CLEARS_GRAD: the forward zeroes the gradient workspace of the
                               lsr_backward that follows (lsr_outputs.grad_ws, sized by lsr_grad_workspace_bytes) — beside
                               its compositing kernel, on a library-owned side stream (workspaces below ~110 MB: in line behind it) forked from and
                               joined back into `stream` with events inside the call (graph-capturable; 16 views x 300 k:
                               0.05 ms of the forward + backward step) — and lsr_backward, handed dims with the same bit,
                               skips its clear.  Set it for BOTH calls or
                               neither.
                               (ABI v9) bit 2, LSR_FWD_REACHED_ONLY: the binning keeps only the (Gaussian, tile) pairs whose
                               alpha >= 1/255 footprint box reaches the tile at all.  The published algorithm pairs a Gaussian
                               with every tile of the 3-sigma square around it; a quarter of those pairs (bench scene; a third
                               for the reference's encoder-shaped scenes) cannot touch a pixel, and the half-tile render lists
                               the compositing kernels walk never held them.  With the bit set they are not counted, keyed or
                               sorted either: images, n_contrib, the render lists and the gradients are the same bit for bit
                               (tests/test_reached_only_gpu.py); what changes is what NO consumer of this path reads —
                               num_pairs, tile_start and bin_point_list describe the reduced lists (the canonical list with
                               the unreachable pairs removed, order kept) instead of the published ones.  0 keeps the
                               published lists (the bit-exact index contract; the default of the C ABI); the autograd op sets
                               it (LSR_REACHED_ONLY=0: not).  Ignored beyond 2^24 Gaussians (no footprint codes).  Hand the
                               same value to every call of a forward and to its lsr_backward.
                               (ABI v10) bit 3, LSR_FWD_FRONT_DONE: lsr_forward_front has already launched the front half of
                               this forward (below); only lsr_forward_nosync / lsr_forward_speculative read it, every other
                               call ignores it.
                               Other bits must be 0 (LSR_EINVAL). */
    int32_t seg_cap_hint;   /* (ABI v9) 0, or the longest tile list (in (Gaussian, tile) pairs) the caller expects — e.g. the
                               `max_tile_pairs` of an earlier call of the same shape plus a margin.  Sizes the per-(view, tile)
                               key segments of the single-pass binning, which are REAL device memory at the end of geom_ws
                               (V * T * capacity * 8 bytes: 268 MB for 16 views of 256 x 256 at the default capacity of 8192,
                               134 MB with a hint of 4096).  A list that outgrows its segment is binned again by the
                               fallback scatter: correct, slower.  Part of the workspace layout: hand the SAME value to
                               lsr_geom_workspace_bytes and to every call that touches the workspaces of a forward. */
} lsr_dims;
#define LSR_FWD_FOR_BACKWARD 1
#define LSR_FWD_CLEARS_GRAD 2
#define LSR_FWD_REACHED_ONLY 4
#define LSR_FWD_FRONT_DONE 8

typedef struct lsr_inputs {
    const float *views;      /* [V][LSR_VIEW_FLOATS] */
    const float *means3D;
    const float *cov3D;
    const float *opacities;
    const float *color;      /* shs or colors_precomp, NULL when color_mode == NONE */
    const float *features;   /* NULL when feat_channels == 0 */
} lsr_inputs;

typedef struct lsr_outputs {
    float *color;    /* [V][3][H][W] or NULL */
    float *feature;  /* [V][C][H][W] or NULL */
    float *mask;     /* [V][H][W]  = 1 - T_final */
    float *depth;    /* [V][H][W]  = sum_i alpha_i T_i z_i */
    int32_t *radii;  /* [V][G] screen radius in pixels, 0 = culled (5th tuple element) */
    void *grad_ws;   /* (ABI v9) forward calls with LSR_FWD_CLEARS_GRAD: the gradient workspace to zero; else ignored (NULL) */
} lsr_outputs;

typedef struct lsr_out_grads { /* any may be NULL (treated as zero) */
    const float *color, *feature, *mask, *depth;
} lsr_out_grads;

typedef struct lsr_in_grads { /* shapes follow the inputs: (G,..) when the stride is 0 (summed
                                 over views) else (V,G,..). Fully overwritten by lsr_backward. */
    float *means3D;   /* required */
    float *cov3D;     /* required */
    float *opacities; /* required */
    float *color;     /* dL/dshs or dL/dcolors_precomp; required iff color_mode != NONE */
    float *features;  /* required iff feat_channels > 0 */
    float *means2D;   /* [V][G][3] NDC-space gradient of the projected mean (x,y,0); optional */
} lsr_in_grads;

/* Debug / test view of the workspaces (byte offsets from the respective workspace base). */
typedef struct lsr_layout {
    /* geom_rec: [V*G][geom_rec_floats] f32 = x_pix y_pix conicA conicB conicC opacity z clampbits payload...
     * geom_bin: [V*G] records of geom_bin_stride bytes: 12 = {u8 rect[4] (minx miny maxx maxy, in tiles); f32 depth;
     *           u8 span[4]} when the tile grid fits byte coordinates, else 16 = {u16 rect[4]; f32 depth; u8 span[4]};
     *           depth 0 = culled; span = footprint of alpha >= 1/255 in 4-pixel cells relative to the rectangle
     * bin_point_list: [P] canonical per-tile lists (depth order, ties by index): tile t of view v owns
     *           [tile_start[v*T+t], tile_start[v*T+t+1])
     * (ABI v8: with key segments — lsr_geom_workspace_bytes includes them, behind everything listed here — bin_keys only
     *           holds the keys of tiles whose lists outgrew their segment; a forward run with LSR_FWD_FOR_BACKWARD narrows the
     *           sub-block bits of bin_half_list to the sub-blocks on which the entry blended a pixel)
     * bin_half_list: [2P] render lists (ABI v6): the tile with canonical list [s, s+n) owns [2s, 2s+2n); the list of
     *           its upper (h = 0: pixel rows 0-7) / lower (h = 1: rows 8-15) half starts at 2s + h*n and has
     *           geom_half_count[2*(v*T+t)+h] entries `index | sub-block bits << 24`: the canonical list restricted to
     *           the entries whose alpha >= 1/255 footprint box reaches the half, in canonical order; bit (4*r + c) =
     *           the entry can reach the half's 4x4-pixel sub-block (c, r).  img_n_contrib counts positions of THESE
     *           lists.  (Sort keys are `depth << 32 | index << 8 | sub-block code`; scenes beyond 2^24 Gaussians use
     *           `depth << 32 | index` and list entries without bits: every entry then reaches every sub-block.) */
    size_t geom_rec, geom_rec_floats, geom_bin, geom_tile_count, geom_tile_start, geom_header;
    size_t bin_keys, bin_point_list;
    size_t img_final_T, img_n_contrib;
    size_t geom_bin_stride;
    size_t bin_half_list, geom_half_count;
    /* (ABI v9) geom_item_flags: [2*V*T] u32, one word per half-tile work item (2*(v*T+t)+h), cleared by every forward;
     *           bit 0 = the forward compositing kernel staged an entry of opacity >= 0.75 for the item: lsr_backward walks
     *           such items back to front (the published recurrence) instead of front to back.  Header word 6
     *           (geom_header + 24) is 1 when the forward's kernel filled the flags in (forwards run with
     *           LSR_FWD_FOR_BACKWARD and all small-batch forwards), else lsr_backward walks every item back to front. */
    size_t geom_item_flags;
} lsr_layout;

int lsr_abi_version(void);
const char *lsr_error_string(int code);
int lsr_last_hip_error(void); /* last hipError_t seen by this thread's lsr call (0 = none) */

/* ---- workspace sizing (host only, no GPU work) ---- */
size_t lsr_geom_workspace_bytes(const lsr_dims *d);
size_t lsr_image_workspace_bytes(const lsr_dims *d);
size_t lsr_binning_workspace_bytes(const lsr_dims *d, int64_t num_pairs, int32_t max_tile_pairs);
size_t lsr_grad_workspace_bytes(const lsr_dims *d);
int lsr_get_layout(const lsr_dims *d, int64_t num_pairs, lsr_layout *out);

/* ---- camera table.  Fills views_out[num_views][LSR_VIEW_FLOATS] on the device from what the
 * reference's render_cuda receives (cuda_splatting.py:56-63): camera-to-world `extrinsics` [V][4][4],
 * normalised `intrinsics` [V][3][3], `near` / `far` [V], background [3] (bg_view_stride 0) or [V][3]
 * (stride 3).  With scale_invariant != 0 the scene scale 1/near is applied to the camera
 * translation and near/far and stored in slot [40] for the kernels to apply to the Gaussians
 * (cuda_splatting.py:75-82).  Replaces get_fov / get_projection_matrix / inverse / matmul
 * (:111-118) — one launch instead of ~40 host-bound PyTorch ops.  Async. */
int lsr_build_views(int32_t num_views, const float *extrinsics, const float *intrinsics, const float *near,
                    const float *far, const float *bg, int32_t bg_view_stride, int32_t scale_invariant,
                    float *views_out, lsr_stream_t stream);

/* ---- one view record straight from the 12-field `GaussianRasterizationSettings` of the reference's
 * per-view call (cuda_splatting.py:132-145): device pointers to `viewmatrix` [16], `projmatrix` [16],
 * `campos` [3], `bg` [3]; tan(fov) by value, or as device scalars when the caller holds tensors (the
 * orthographic path, :260-261 — no host read-back then).  Scene scale 1.  Async, one tiny launch. */
int lsr_pack_view(const float *viewmatrix, const float *projmatrix, const float *campos, const float *bg,
                  float tanfovx, float tanfovy, const float *tanfovx_dev, const float *tanfovy_dev,
                  float *view_out, lsr_stream_t stream);

/* ---- forward, phase 1: per-Gaussian preprocess + per-tile counting + tile offset scan — and, since ABI v8, the BINNING:
 * every (view, tile) owns a fixed-capacity key segment at the end of geom_ws, a workgroup of the preprocess kernel reserves
 * its slots with the (returning) count atomics and writes the sort keys of its Gaussians itself (upstream's
 * duplicateWithKeys, fused; no scatter kernel, no second pass over the binning records).  Dims for which that does not apply
 * (tile grids beyond 255 a side or 1024 tiles per view; LSR_SEGMENTS=0) keep the two-phase binning in phase 2.
 * Writes radii.  Waits once for the device to return the pair count and the longest tile list through
 * the two host pointers (both required): the last workgroup of the preprocess kernel scans the tile counts
 * (ABI v7: no separate scan kernel for calls of up to 4096 (view, tile) pairs) and writes the two numbers plus the
 * call's sequence number into mapped host memory as soon as it has them; the host polls that word (an event behind
 * the kernel is the fallback) — the rest of the scan is still running when the call returns.
 * The view-dependent payload (colour / latent features from SH) is evaluated by the same kernel as the projection
 * when the call's payload is harmonics only and its views share their inputs (one scene or view groups: ABI v7), else by
 * a second launch right behind it — all on `stream` (rounds 2-3 ran that pass on a library-owned side stream; it lost
 * to the in-line launch once the host stopped sleeping on an event, see api.hip). */
int lsr_forward_prepare(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, int32_t *radii,
                        int64_t *num_pairs_host, int32_t *max_tile_pairs_host,
                        lsr_stream_t stream);

/* ---- forward, phase 2: per-tile depth sort, front-to-back compositing (and the binning of whatever phase 1 left: all of it
 * for dims without key segments, else only the tiles whose lists outgrew their segment — `max_tile_pairs` tells). Async.  Must
 * follow the lsr_forward_prepare of the same geom_ws on the same stream, with the two numbers it returned. */
int lsr_forward_render(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, void *bin_ws,
                       void *img_ws, int64_t num_pairs, int32_t max_tile_pairs,
                       const lsr_outputs *out, lsr_stream_t stream);

/* ---- a forward that is given up between lsr_forward_prepare and lsr_forward_render (e.g. the binning workspace
 * could not be allocated).  ABI v6 had to join the library's side stream here before geom_ws could be released; since
 * v7 every launch of a forward is on `stream`, stream-ordered release is safe by itself and this call does nothing.
 * Kept so that v6 callers keep linking. */
int lsr_forward_abandon(lsr_stream_t stream);

/* ---- forward WITHOUT host synchronisation (latency mode; graph-capturable).  The same stages as
 * lsr_forward_prepare + lsr_forward_render, but the pair count never travels to the host (upstream's
 * blocking `num_rendered` read-back, SURVEY.md Appendix A.3 step 3, is what makes the reference's
 * per-view loop at cuda_splatting.py:124-162 serialise host and device):
 *   - bin_ws must hold `pair_capacity` pairs:
 *       lsr_binning_workspace_bytes(d, pair_capacity, INT32_MAX)   (includes the merge scratch);
 *   - `max_tile_hint` = expected longest tile list (e.g. from the previous frame's
 *     lsr_forward_status); it only selects the LDS sort variant, longer lists still sort correctly;
 *   - if the scene produces MORE pairs than `pair_capacity`, the last tile lists are truncated (no
 *     out-of-bounds access), the images are then wrong and the overflow word is set: check it with
 *     lsr_forward_status at the next convenient synchronisation point and re-run with more room;
 *   - a tile list longer than its key segment is NOT an overflow: the fallback scatter for such tiles is always part of
 *     this launch sequence (its workgroups leave at once when the device's longest list fits);
 *   - lsr_backward takes `num_pairs = pair_capacity` for such a forward.
 * No allocation, no host wait, no host-visible write: the launch sequence can be captured in a
 * hipGraph (torch.cuda.graph) and replayed. */
int lsr_forward_nosync(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, void *bin_ws, void *img_ws,
                       int64_t pair_capacity, int32_t max_tile_hint, const lsr_outputs *out,
                       lsr_stream_t stream);

/* ---- forward with a SPECULATIVE workspace size (ABI v8).  The synchronous forward idles the device for ~10 us per call:
 * between the host's read of the pair count in lsr_forward_prepare and the arrival of lsr_forward_render's launches.
 * A caller that renders the same shape again and again (a training loop) knows roughly how many pairs to expect: this call
 * launches the WHOLE forward at once — bin_ws sized for `pair_capacity` pairs exactly as for lsr_forward_nosync,
 * `max_tile_hint` >= 1 the longest tile list the launch structure (sort tiers, fallback scatter) is chosen for — and only
 * then waits for the device's counts the way lsr_forward_prepare does, while the device goes on sorting and compositing.
 *   *overflow_host == 0: the forward is complete and bit-identical to lsr_forward_prepare + lsr_forward_render;
 *                        lsr_backward takes `num_pairs = pair_capacity` for it (the workspace layout).
 *   *overflow_host != 0: the scene produced more pairs than `pair_capacity` or a list longer than `max_tile_hint`; such
 *                        lists were left out (no out-of-bounds access), the outputs are incomplete: run
 *                        lsr_forward_prepare + lsr_forward_render on the same workspaces (they size bin_ws exactly).
 * The counts returned are the true ones either way (what the caller bases its next capacity on). */
int lsr_forward_speculative(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, void *bin_ws, void *img_ws,
                            int64_t pair_capacity, int32_t max_tile_hint, const lsr_outputs *out,
                            int64_t *num_pairs_host, int32_t *max_tile_pairs_host, int32_t *overflow_host,
                            lsr_stream_t stream);

/* ---- the FRONT HALF of a no-sync / speculative forward, launched early (ABI v10).  A caller that still has host work to do
 * before it can make the full call — allocating the outputs, the image and binning workspaces: nine allocations, ~25 us of
 * Python in the autograd op — hands over what the front half needs (dims, inputs, geom_ws, radii, the `pair_capacity` of the
 * call to come) and the device starts on the projection / key emission / tile scan while the host goes on.  The full call
 * then follows on the SAME host thread and stream with the same dims + LSR_FWD_FRONT_DONE, the same geom_ws,
 * `out->radii` == radii and the same pair_capacity, and launches the rest.  Results are those of the one-call form bit for
 * bit (the same launches in the same stream order).  A call with LSR_FWD_FRONT_DONE that does not match the thread's pending
 * front half (other geom_ws / capacity, or none pending) returns LSR_EINVAL and launches nothing.  No allocation, no host
 * wait: graph-capturable like lsr_forward_nosync.  A synchronised call into an idle device: V = 1 0.128 -> 0.120 ms, V = 4
 * 0.201 -> 0.193 (profiles/r06_ab_knobs.md section 12); costs back-to-back callers one more C call (~5 us of host time). */
int lsr_forward_front(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, int32_t *radii, int64_t pair_capacity,
                      lsr_stream_t stream);

/* Pair count, longest tile list and overflow flag (0/1) of the most recent forward that used
 * geom_ws.  Copies 32 bytes to the host and synchronises `stream`. */
int lsr_forward_status(const lsr_dims *d, const void *geom_ws, int64_t *num_pairs_host,
                       int32_t *max_tile_pairs_host, int32_t *overflow_host, lsr_stream_t stream);

/* ---- backward. Needs the three workspaces of the matching forward, unmodified, and the images
 * that forward produced (`fwd`: colour / feature / depth are read wherever the corresponding
 * gradient in `gout` is given; mask and radii are not used).  The compositing gradient walks the
 * tile lists front to back like the forward and gets "everything behind this entry" as
 * (rendered value - prefix), which is why the rendered values are an input.  Async, everything on `stream`.
 * With several scenes in the call (views_per_group) the geometry and SH backward kernels behind the compositing
 * backward still are ONE launch each (the scene is a grid dimension). */
int lsr_backward(const lsr_dims *d, const lsr_inputs *in, const void *geom_ws,
                 const void *bin_ws, const void *img_ws, int64_t num_pairs,
                 const int32_t *radii, const lsr_outputs *fwd, const lsr_out_grads *gout,
                 void *grad_ws, const lsr_in_grads *gin, lsr_stream_t stream);

/* ---- optional measurement hook (bench.py): when enabled, every stage kernel is bracketed by
 * hipEvents on the caller's stream; lsr_profile_read() waits for them, returns the accumulated
 * milliseconds and launch counts per stage since the previous read, and resets the totals.
 * Arrays must hold lsr_profile_num_stages() entries. Not thread-safe; one instance per process,
 * matching the reference's one-rasterizer-per-DDP-process use.
 * on = 0: off; 1: every stage; otherwise a bit mask, bit (s + 1) enables stage s only (two event
 * packets per enabled stage and launch are the whole cost). */
int lsr_profile_enable(int on);
int lsr_profile_num_stages(void);
const char *lsr_profile_stage_name(int stage);
int lsr_profile_read(double *ms_out, int64_t *launches_out);

/* ---- development aid (kernel A/B experiments, tools/): overrides one of the library's LSR_* environment knobs
 * (launch-shape variants, LSR_FOLD_SCAN, LSR_FWD_ROWS, ...: INTEGRATION.md section 4) for the rest of the process.  Knobs
 * select between implementations that produce identical results (LSR_FWD_QUAD: identical transmittance / mask / list
 * prefixes, colour sums in a different order); nothing in the product path calls this. */
int lsr_debug_set_knob(const char *name, int value);

/* ---- arithmetic convention of the projection stage (ABI v7).  0 (default): every float operation of the published
 * preprocess is a separate IEEE operation (what the oracle and the bit-exact index tests assume); 1: products feeding
 * sums are contracted into fused multiply-adds the way a compiler with contraction on (nvcc's default -fmad=true)
 * would build the published source — tools/contraction_census.py quantifies how many radii / tile rectangles / list
 * positions differ between the two.  Process-wide; takes effect at the next forward. */
int lsr_set_projection_contraction(int on);
int lsr_get_projection_contraction(void);

#ifdef __cplusplus
}
#endif
#endif /* LSR_RASTERIZER_H */
