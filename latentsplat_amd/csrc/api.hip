// api.hip — the C ABI declared in include/lsr_rasterizer.h (argument validation, workspace
// layout, stage sequencing).  No torch, no allocation, nothing thrown across the boundary.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lsr_internal.h"

using namespace lsr;

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>

static thread_local int g_last_hip_error = 0;

// ---- per-stage timing with hipEvents recorded on the caller's stream -----------------------
namespace {
struct Prof {
    unsigned on = 0;                                    // 0 off, 1 all stages, else bit (s + 1) = stage s
    bool wants(int stage) const { return on == 1u || ((on >> (stage + 1)) & 1u); }
    std::vector<hipEvent_t> pool;                       // free events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[lsr::kNumStages];
    hipEvent_t open_ev[lsr::kNumStages] = {};
    double ms[lsr::kNumStages] = {};
    int64_t n[lsr::kNumStages] = {};
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
Prof g_prof;
std::mutex g_prof_mu;   // the measurement hook is process-wide state; calls may come from several host threads
}  // namespace
void lsr::prof_begin(int stage, hipStream_t s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof.wants(stage)) return;
    hipEvent_t e = g_prof.get();
    (void)hipEventRecord(e, s);
    g_prof.open_ev[stage] = e;
}
void lsr::prof_end(int stage, hipStream_t s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof.open_ev[stage]) return;
    hipEvent_t e = g_prof.get();
    (void)hipEventRecord(e, s);
    g_prof.pending[stage].push_back({g_prof.open_ev[stage], e});
    g_prof.open_ev[stage] = nullptr;
}


void lsr::note_hip_error(int e) { g_last_hip_error = e; }

// Compute units of the current device, cached per device id (a process may drive several GPUs).
int lsr::device_cus() {
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;   // MI355X
        }
        __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

bool lsr::deterministic_backward() { return env_int("LSR_DETERMINISTIC", 0) != 0; }

// -1: not set yet (the environment default LSR_PROJECTION_CONTRACTION, 0 unless given, applies)
static int g_projection_contraction = -1;
bool lsr::projection_contraction() {
    const int v = __atomic_load_n(&g_projection_contraction, __ATOMIC_RELAXED);
    return (v < 0 ? env_int("LSR_PROJECTION_CONTRACTION", 0) : v) != 0;
}

// Keys per (view, tile) segment of the single-pass binning for a call of these dims; 0 = two-phase binning (lsr_internal.h).
// A pure function of the dims and two knobs, so that every stage of a call (and lsr_geom_workspace_bytes before it) agrees.
//   * byte tile coordinates and at most 1024 tiles per view: the projection kernel's LDS tile histogram (4 views per
//     workgroup) and its 12-byte binning records;
//   * the caller's hint (lsr_dims::seg_cap_hint: the longest list it expects, e.g. from an earlier call of the shape) rounded
//     up to 64 keys, else 8192 (the second sort tier); halved while the segments exceed LSR_SEG_BUDGET_MB (default 512 MB of
//     REAL device memory at the end of the geometry workspace, P * 8 bytes of it ever written) or V * T * capacity reaches
//     2^32 (the kernels index the segments with 32 bits); never more than the scene has Gaussians (a list cannot be longer).
uint32_t lsr::segment_capacity(const lsr_dims &d) {
    if (!env_int("LSR_SEGMENTS", 1) || d.num_gaussians <= 0) return 0u;
    const int64_t T = num_tiles(d);
    if (!narrow_bins(d) || T > 1024) return 0u;
    if (fused_preprocess_sh(d) && !fused_segments_fit(d)) return 0u;   // (its LDS: coefficient rows + two count arrays)
    const int64_t VT = (int64_t)d.num_views * T;
    const int64_t budget = (int64_t)env_int("LSR_SEG_BUDGET_MB", 512) << 20;
    int64_t cap = 8192;
    if (d.seg_cap_hint > 0) cap = std::min<int64_t>(8192, std::max<int64_t>(256, ((int64_t)d.seg_cap_hint + 63) / 64 * 64));
    while (cap > 1024 && (VT * cap * 8 > budget || VT * cap >= ((int64_t)1 << 32))) cap = (cap / 2 + 63) / 64 * 64;
    if (VT * cap * 8 > budget || VT * cap >= ((int64_t)1 << 32)) return 0u;
    const int64_t g64 = ((int64_t)d.num_gaussians + 63) / 64 * 64;
    if (g64 < cap) cap = g64;
    // LSR_SEG_CAP (tests): a smaller capacity than the policy's, to exercise the overflow path on small scenes
    const int64_t forced = env_int("LSR_SEG_CAP", 0);
    if (forced > 0 && forced < cap) cap = (forced + 63) / 64 * 64;
    return (uint32_t)cap;
}

// Development knobs (LSR_FWD_VARIANT, LSR_SH_PLACEMENT, ...): read from the environment once per process and knob,
// then served from a table — no getenv on the launch path.  lsr_debug_set_knob overrides a table entry at run time
// (kernel A/B experiments in one process; not part of the product surface).
namespace {
struct Knob { char name[32]; int value; };
Knob g_knobs[32];
int g_n_knobs = 0;
std::mutex g_knob_mu;
Knob *find_knob(const char *name) {
    for (int i = 0; i < g_n_knobs; ++i)
        if (!strcmp(g_knobs[i].name, name)) return &g_knobs[i];
    return nullptr;
}
}  // namespace
int lsr::env_int(const char *name, int fallback) {
    std::lock_guard<std::mutex> lock(g_knob_mu);
    if (Knob *k = find_knob(name)) return k->value;
    const char *e = getenv(name);
    const int v = e ? atoi(e) : fallback;
    if (g_n_knobs < 32 && strlen(name) < sizeof(g_knobs[0].name)) {
        strcpy(g_knobs[g_n_knobs].name, name);
        g_knobs[g_n_knobs++].value = v;
    }
    return v;
}

static int fail_hip(hipError_t e) {
    g_last_hip_error = (int)e;
    return LSR_ELAUNCH;
}
#define LSR_HIP(call)                                 \
    do {                                              \
        hipError_t e__ = (call);                      \
        if (e__ != hipSuccess) return fail_hip(e__);  \
    } while (0)

// LSR_DEBUG_SYNC=1: announce and synchronise after every stage (the analogue of the upstream
// op's debug=True mode) so a faulting kernel can be identified.
static bool debug_sync() {
    static const bool on = getenv("LSR_DEBUG_SYNC") != nullptr;
    return on;
}
#define LSR_STAGE(name, s, call)                                        \
    do {                                                                \
        if (debug_sync()) { fprintf(stderr, "[lsr] %s ...\n", name); fflush(stderr); } \
        LSR_HIP(call);                                                  \
        if (debug_sync()) { LSR_HIP(hipStreamSynchronize(s)); fprintf(stderr, "[lsr] %s ok\n", name); } \
    } while (0)

static int check_dims(const lsr_dims *d) {
    if (!d) return LSR_ENULL;
    if (d->num_views < 1 || d->num_gaussians < 0 || d->height < 1 || d->width < 1) return LSR_EINVAL;
    if (d->feat_channels < 0 || d->feat_channels > LSR_MAX_FEAT_CHANNELS) return LSR_EINVAL;
    if (d->color_mode < LSR_COLOR_NONE || d->color_mode > LSR_COLOR_PRECOMP) return LSR_EINVAL;
    if (d->color_mode == LSR_COLOR_NONE && d->feat_channels == 0) return LSR_EINVAL;
    if (d->color_mode == LSR_COLOR_SH) {
        if (d->sh_degree < 0 || d->sh_degree > LSR_MAX_SH_DEGREE) return LSR_EINVAL;
        if (d->sh_coeffs < (d->sh_degree + 1) * (d->sh_degree + 1)) return LSR_EINVAL;
    }
    if (tiles_x(*d) > 65535 || tiles_y(*d) > 65535) return LSR_EUNSUPPORTED;
    // work items pack (view*T + tile) into 28 bits (kItemTileMask)
    if ((int64_t)d->num_views * num_tiles(*d) >= (int64_t)1 << 28) return LSR_EUNSUPPORTED;
    // (view, Gaussian) slots are indexed with 31 bits (scenes beyond 2^24 Gaussians run without footprint culling:
    // lsr_internal.h IndexPacking)
    if ((int64_t)d->num_views * d->num_gaussians >= (int64_t)1 << 31) return LSR_EUNSUPPORTED;
    // per-view strides: 0 (shared scene) or exactly one dense (G, ...) array per view
    const int64_t G = d->num_gaussians;
    const int64_t color_elems = d->color_mode == LSR_COLOR_SH ? (int64_t)d->sh_coeffs * 3 : 3;
    if (d->vs_means != 0 && d->vs_means != 3 * G) return LSR_EINVAL;
    if (d->cov_elems != 6 && d->cov_elems != 9) return LSR_EINVAL;
    if (d->vs_cov != 0 && d->vs_cov != (int64_t)d->cov_elems * G) return LSR_EINVAL;
    if (d->vs_opac != 0 && d->vs_opac != G) return LSR_EINVAL;
    if (d->color_mode != LSR_COLOR_NONE && d->vs_color != 0 && d->vs_color != color_elems * G) return LSR_EINVAL;
    if (d->feat_mode != LSR_FEAT_DIRECT && d->feat_mode != LSR_FEAT_SH) return LSR_EINVAL;
    int64_t feat_elems = d->feat_channels;
    if (d->feat_channels > 0 && d->feat_mode == LSR_FEAT_SH) {
        if (d->feat_sh_degree < 0 || d->feat_sh_degree > 2) return LSR_EUNSUPPORTED;
        if (d->feat_sh_coeffs < (d->feat_sh_degree + 1) * (d->feat_sh_degree + 1)) return LSR_EINVAL;
        feat_elems = (int64_t)d->feat_channels * d->feat_sh_coeffs;
        if (feat_elems > 120) return LSR_EUNSUPPORTED;   // LDS budget of the SH kernels (sh.hip)
    }
    if (d->color_mode == LSR_COLOR_SH && d->sh_coeffs * 3 > 120) return LSR_EUNSUPPORTED;
    if (d->feat_channels > 0 && d->vs_feat != 0 && d->vs_feat != feat_elems * G) return LSR_EINVAL;
    if (d->color_sh_convention != LSR_SH_AXES_3DGS && d->color_sh_convention != LSR_SH_AXES_REFERENCE) return LSR_EINVAL;
    if (d->views_per_group < 0) return LSR_EINVAL;
    if (d->forward_flags & ~(LSR_FWD_FOR_BACKWARD | LSR_FWD_CLEARS_GRAD | LSR_FWD_REACHED_ONLY | LSR_FWD_FRONT_DONE)) return LSR_EINVAL;
    if (d->seg_cap_hint < 0) return LSR_EINVAL;
    if (d->views_per_group > 1) {   // view groups: all inputs strided per group
        if (d->num_views % d->views_per_group != 0) return LSR_EINVAL;
        if (d->vs_means == 0 || d->vs_cov == 0 || d->vs_opac == 0) return LSR_EUNSUPPORTED;
        if (d->color_mode != LSR_COLOR_NONE && d->vs_color == 0) return LSR_EUNSUPPORTED;
        if (d->feat_channels > 0 && d->vs_feat == 0) return LSR_EUNSUPPORTED;
    }
    return LSR_OK;
}

// a separate SH payload pass behind k_preprocess (not when the fused projection + SH kernel handles the call)
static bool has_sh_payload(const lsr_dims &d) {
    return d.num_gaussians > 0 && (d.color_mode == LSR_COLOR_SH || (d.feat_channels > 0 && d.feat_mode == LSR_FEAT_SH)) &&
           !fused_preprocess_sh(d);
}
// The SH payload pass of calls the fused kernel does not cover (per-view inputs, direct payload channels next to
// harmonics) runs IN LINE on the caller's stream, right behind k_preprocess.  Rounds 2-3 ran it on a library-owned side
// stream beside the tile scan, the host round trip and the binning chain (forked / joined with events).  Measured in
// round 4 (tools/ab_knobs.py): beside the binning chain the SH pass and the per-tile sort slow each other
// down by more than the overlap hides — configs[3] forward 0.325 ms (side stream) vs 0.313 (in line), configs[4] 0.926 vs
// 0.912 — and with the host polling for the pair count there is no round trip left to fill.  The side stream, its events
// and its mutex are gone; lsr_forward_abandon stays in the ABI as a no-op.
static int sh_forward_inline(const lsr_dims &d, const lsr_inputs &in, char *geom, hipStream_t s) {
    if (!has_sh_payload(d)) return LSR_OK;
    LSR_STAGE("sh_forward", s, launch_sh_forward(d, in, geom, s));   // all view groups in one launch
    return LSR_OK;
}

// The library's one side stream (per host thread and device; created on first use): only ever forked from and joined
// back into the caller's stream with events inside one call, so the caller's stream order — and a hipGraph capture of
// it — covers everything launched there.
namespace {
struct SideStream { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
bool side_stream(SideStream &out) {
    static thread_local SideStream per_dev[64];
    static thread_local bool failed[64] = {};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || failed[dev]) { (void)hipGetLastError(); return false; }
    SideStream &ss = per_dev[dev];
    if (!ss.stream) {
        if (hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            failed[dev] = true;
            ss = SideStream();
            return false;
        }
    }
    out = ss;
    return true;
}
}  // namespace

// Binning + forward compositing on `s`.
// (Round 3 also carried an in-call pipeline that ran the binning of one half of the views beside the compositing of
// the other half on a side stream: bit-identical, never faster — both halves are issue-bound — and deleted in
// round 4.  Callers with independent batches overlap whole calls on two streams instead: INTEGRATION.md.)
static int forward_tail(const lsr_dims &d, const lsr_inputs &in, char *geom, char *bin, char *img, int64_t num_pairs,
                        int32_t max_tile_pairs, const lsr_outputs &out, hipStream_t s, bool device_counts, bool seg,
                        bool speculative = false) {
    // LSR_FWD_CLEARS_GRAD: the backward's gradient workspace is zeroed BESIDE the compositing kernel, on a side stream forked
    // in front of it and joined behind it.  The compositing kernel is bound by instruction issue and leaves register space for
    // one more wave per SIMD (6 x 80 of 512 VGPRs) and three quarters of the HBM bandwidth; the streaming clear (0.05 ms per
    // forward + backward step at 16 views x 300 k when it runs alone in front of the compositing backward) costs it 0.011 ms.
    // Measured (profiles/r06_ab_knobs.md section 5): forward + backward 1.0775 -> 1.056 ms; forked in front of the per-tile
    // sort as well, the sort pays what the clear saves (0.068 -> 0.098 ms: 1.089); the compositing waves storing the zeros
    // themselves, a few KB per batch: 1.058.  The fork / join costs 15-20 us of its own: below ~110 MB of workspace the clear
    // is shorter than that and runs in line behind the compositing kernel (forward + backward, beside vs in line: 2 views x
    // 300 k 0.290 vs 0.283 ms, 4 views 0.3863 vs 0.3845, configs[3] 0.745 vs 0.741; 6 views 0.552 vs 0.559, configs[4] 2.18 vs
    // 2.20: section 5).  LSR_CLEAR_BESIDE = 0 / 1: never / always beside.
    const bool clears = (d.forward_flags & LSR_FWD_CLEARS_GRAD) != 0 && d.num_gaussians > 0;
    SideStream side;
    const int beside_knob = env_int("LSR_CLEAR_BESIDE", -1);
    const bool beside = clears && (beside_knob >= 0 ? beside_knob != 0 : grad_layout(d).fixed >= (size_t)110 * 1000 * 1000) && side_stream(side);
    LSR_STAGE("binning", s, launch_binning(d, geom, bin, num_pairs, max_tile_pairs, out.radii, s, device_counts, seg, speculative));
    bool forked = false;
    if (beside) {
        LSR_HIP(hipEventRecord(side.fork, s));
        LSR_HIP(hipStreamWaitEvent(side.stream, side.fork, 0));
        forked = true;      // from here on the side stream is joined back whatever happens (the workspace is the caller's)
        hipError_t e = launch_clear_grad(d, out.radii, (char *)out.grad_ws, side.stream);
        if (e == hipSuccess) e = hipEventRecord(side.join, side.stream);
        if (e != hipSuccess) { (void)hipStreamSynchronize(side.stream); return fail_hip(e); }
    }
    const hipError_t er = launch_render_forward(d, in, geom, bin, num_pairs, img, out, s);
    if (forked) {
        const hipError_t ej = hipStreamWaitEvent(s, side.join, 0);
        if (ej != hipSuccess) { (void)hipStreamSynchronize(side.stream); return fail_hip(ej); }
    }
    LSR_STAGE("render_forward", s, er);
    if (clears && !forked) LSR_HIP(launch_clear_grad(d, out.radii, (char *)out.grad_ws, s));
    return LSR_OK;
}

static int check_inputs(const lsr_dims *d, const lsr_inputs *in) {
    if (!in || !in->views) return LSR_ENULL;
    if (d->num_gaussians > 0) {
        if (!in->means3D || !in->cov3D || !in->opacities) return LSR_ENULL;
        if (d->color_mode != LSR_COLOR_NONE && !in->color) return LSR_ENULL;
        if (d->feat_channels > 0 && !in->features) return LSR_ENULL;
    }
    return LSR_OK;
}

extern "C" {

int lsr_abi_version(void) { return LSR_ABI_VERSION; }

int lsr_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof.on = (unsigned)on;
    return LSR_OK;
}
int lsr_profile_num_stages(void) { return lsr::kNumStages; }
const char *lsr_profile_stage_name(int stage) {
    static const char *names[lsr::kNumStages] = {"preprocess", "tile_scan", "scatter", "sort_tiles",
                                                  "render_forward", "render_backward", "preprocess_backward",
                                                  "sh_forward", "sh_backward", "adapter_forward", "adapter_backward", "latent_forward", "latent_backward"};
    return (stage >= 0 && stage < lsr::kNumStages) ? names[stage] : "?";
}
int lsr_profile_read(double *ms_out, int64_t *launches_out) {
    if (!ms_out || !launches_out) return LSR_ENULL;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (int st = 0; st < lsr::kNumStages; ++st) {
        for (auto &pr : g_prof.pending[st]) {
            float t = 0.0f;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) {
                g_prof.ms[st] += t;
                g_prof.n[st] += 1;
            }
            g_prof.pool.push_back(pr.first);
            g_prof.pool.push_back(pr.second);
        }
        g_prof.pending[st].clear();
        ms_out[st] = g_prof.ms[st];
        launches_out[st] = g_prof.n[st];
        g_prof.ms[st] = 0.0;
        g_prof.n[st] = 0;
    }
    return LSR_OK;
}
int lsr_last_hip_error(void) { return g_last_hip_error; }

int lsr_set_projection_contraction(int on) {
    __atomic_store_n(&g_projection_contraction, on ? 1 : 0, __ATOMIC_RELAXED);
    return LSR_OK;
}
int lsr_get_projection_contraction(void) { return lsr::projection_contraction() ? 1 : 0; }

int lsr_debug_set_knob(const char *name, int value) {
    if (!name || strlen(name) >= sizeof(g_knobs[0].name)) return LSR_EINVAL;
    std::lock_guard<std::mutex> lock(g_knob_mu);
    if (Knob *k = find_knob(name)) { k->value = value; return LSR_OK; }
    if (g_n_knobs >= 32) return LSR_EINVAL;
    strcpy(g_knobs[g_n_knobs].name, name);
    g_knobs[g_n_knobs++].value = value;
    return LSR_OK;
}

const char *lsr_error_string(int code) {
    switch (code) {
        case LSR_OK: return "ok";
        case LSR_EINVAL: return "invalid dimensions or argument combination";
        case LSR_ENULL: return "required pointer is NULL";
        case LSR_ELAUNCH: return "HIP runtime error (see lsr_last_hip_error)";
        case LSR_ECAPACITY: return "the scene produced more (Gaussian, tile) pairs than the capacity given to lsr_forward_nosync";
        case LSR_EUNSUPPORTED: return "unsupported size";
        default: return "unknown lsr error";
    }
}

size_t lsr_geom_workspace_bytes(const lsr_dims *d) { return check_dims(d) ? 0 : geom_layout(*d).total; }
size_t lsr_image_workspace_bytes(const lsr_dims *d) { return check_dims(d) ? 0 : img_layout(*d).total; }
size_t lsr_binning_workspace_bytes(const lsr_dims *d, int64_t num_pairs, int32_t max_tile_pairs) {
    return check_dims(d) ? 0 : bin_layout(*d, num_pairs, max_tile_pairs).total;
}
size_t lsr_grad_workspace_bytes(const lsr_dims *d) { return check_dims(d) ? 0 : grad_layout(*d).total; }

int lsr_get_layout(const lsr_dims *d, int64_t num_pairs, lsr_layout *out) {
    int rc = check_dims(d);
    if (rc) return rc;
    if (!out) return LSR_ENULL;
    const GeomLayout L = geom_layout(*d);
    const BinLayout B = bin_layout(*d, num_pairs, 0);
    const ImgLayout I = img_layout(*d);
    out->geom_rec = L.rec; out->geom_rec_floats = (size_t)L.rec_floats; out->geom_bin = L.bin;
    out->geom_tile_count = L.tile_count; out->geom_tile_start = L.tile_start; out->geom_header = L.header;
    out->bin_keys = B.keys; out->bin_point_list = B.point_list;
    out->bin_half_list = B.half_list; out->geom_half_count = L.half_count;
    out->img_final_T = I.final_T; out->img_n_contrib = I.n_contrib;
    out->geom_bin_stride = bin_stride(*d);
    out->geom_item_flags = L.item_flags;
    return LSR_OK;
}

int lsr_build_views(int32_t num_views, const float *extrinsics, const float *intrinsics, const float *near,
                    const float *far, const float *bg, int32_t bg_view_stride, int32_t scale_invariant,
                    float *views_out, lsr_stream_t stream) {
    g_last_hip_error = 0;
    if (num_views < 1 || (bg_view_stride != 0 && bg_view_stride != 3)) return LSR_EINVAL;
    if (!extrinsics || !intrinsics || !near || !far || !bg || !views_out) return LSR_ENULL;
    LSR_HIP(launch_build_views(num_views, extrinsics, intrinsics, near, far, bg, bg_view_stride,
                               scale_invariant != 0, views_out, (hipStream_t)stream));
    return LSR_OK;
}

int lsr_pack_view(const float *viewmatrix, const float *projmatrix, const float *campos, const float *bg,
                  float tanfovx, float tanfovy, const float *tanfovx_dev, const float *tanfovy_dev, float *view_out,
                  lsr_stream_t stream) {
    g_last_hip_error = 0;
    if (!viewmatrix || !projmatrix || !campos || !bg || !view_out) return LSR_ENULL;
    LSR_HIP(launch_pack_view(viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, tanfovx_dev, tanfovy_dev, view_out,
                             (hipStream_t)stream));
    return LSR_OK;
}

// The host side of the synchronous forward's only wait: the pair count and the longest list arrive in three mapped,
// pinned host words (pair count, longest list, sequence number of the call) that the scanning workgroup writes as soon
// as it has the totals — before it goes on to the tile offsets and the work items (lsr_tile_scan.h).  The host polls
// the sequence word (a cache line of its own memory until the device's write lands) instead of sleeping on an event:
// the wake-up through the runtime cost more than the scan it was hiding.  The event recorded behind the kernel is
// the safety net: queried every few microseconds, and once it has completed without the sequence word showing up
// (cannot happen unless the launch failed) the header is read back with a copy.  LSR_HOST_POLL=0: wait on the event.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}
static int wait_pair_count(volatile uint32_t *h, uint32_t seq, hipEvent_t ev, bool &have) {
    have = false;
    if (!env_int("LSR_HOST_POLL", 1)) {
        LSR_HIP(hipEventSynchronize(ev));
        have = __atomic_load_n(&h[2], __ATOMIC_ACQUIRE) == seq;
        return LSR_OK;
    }
    // Bounded spin: the words normally arrive within the projection kernel's ~0.1 ms.  When the caller's stream has a
    // backlog in front of this call (a training step with encoder kernels queued ahead) the wait can be long: after
    // ~0.3 ms of spinning the thread yields between polls, and after ~4 ms it sleeps on the event instead of burning
    // a core for the whole backlog.
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n(&h[2], __ATOMIC_ACQUIRE) == seq) { have = true; return LSR_OK; }
        cpu_relax();
        if ((spins & 0x3FFu) == 0u) {
            const hipError_t q = hipEventQuery(ev);
            if (q == hipSuccess) {
                have = __atomic_load_n(&h[2], __ATOMIC_ACQUIRE) == seq;
                return LSR_OK;
            }
            if (q != hipErrorNotReady) return fail_hip(q);
            (void)hipGetLastError();   // hipErrorNotReady is sticky in the last-error slot
            const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (waited > 4000) {
                LSR_HIP(hipEventSynchronize(ev));
                have = __atomic_load_n(&h[2], __ATOMIC_ACQUIRE) == seq;
                return LSR_OK;
            }
            if (waited > 300) sched_yield();
        }
    }
}

// The per-thread mapped host words through which the scanning workgroup reports the pair count and the longest list
// (pinned, device-mapped; the library's only allocation), the event behind the projection kernel (the polling loop's
// safety net) and the sequence number of the thread's calls.
namespace {
struct HostWords {
    uint32_t *host = nullptr, *dev = nullptr;
    hipEvent_t event = nullptr;
    bool mapped() const { return event != nullptr; }
};
uint32_t next_seq() {
    static thread_local uint32_t h_seq = 0;
    if (++h_seq == 0u) h_seq = 1u;                        // 0 is what a fresh buffer holds
    return h_seq;
}
void host_words(HostWords &w) {
    static thread_local uint32_t *h_hdr = nullptr, *h_hdr_dev = nullptr;
    static thread_local hipEvent_t h_events[64] = {};     // one per device this thread has used (events belong to a device)
    static thread_local bool h_tried = false;
    if (!h_tried) {
        h_tried = true;
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess &&
            hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            h_hdr = (uint32_t *)hp; h_hdr_dev = (uint32_t *)dp;
            memset(hp, 0, 64);
        } else {
            (void)hipGetLastError();
        }
    }
    w = HostWords();
    int dev = -1;
    if (h_hdr && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (!h_events[dev] && hipEventCreateWithFlags(&h_events[dev], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            h_events[dev] = nullptr;
        }
        w.event = h_events[dev];
    }
    w.host = h_hdr; w.dev = h_hdr_dev;
}
// the host's only wait of a forward: pair count and longest list of the call with sequence number `seq`
int read_counts(const lsr_dims &d, const char *geom, const HostWords &hw, uint32_t seq, hipStream_t s, uint32_t hdr[2]) {
    bool have = false;
    hdr[0] = hdr[1] = 0;
    if (hw.mapped()) {
        int rc = wait_pair_count(hw.host, seq, hw.event, have);
        if (rc) return rc;
        if (have) { hdr[0] = hw.host[0]; hdr[1] = hw.host[1]; }
    }
    if (!have) {
        LSR_HIP(hipMemcpyAsync(hdr, geom + geom_layout(d).header, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        LSR_HIP(hipStreamSynchronize(s));
    }
    return LSR_OK;
}
// Projection (+ key emission, + SH payload) and the tile scan of one forward.  host: report the counts to the mapped
// host words under `seq`; capacity: pairs the binning workspace holds (UINT32_MAX: sized after the fact).
int launch_front(const lsr_dims &d, const lsr_inputs &in, char *geom, int32_t *radii, const HostWords *hw, uint32_t seq,
                 uint32_t capacity, hipStream_t s) {
    const bool fold = fold_tile_scan(d);
    const bool seg = segment_capacity(d) != 0u;
    const bool mapped = hw && hw->mapped();
    FoldedScan fs{};
    fs.enabled = fold ? 1 : 0;
    fs.host_words = mapped ? hw->dev : nullptr; fs.host_seq = seq; fs.capacity = capacity;
    if (fused_preprocess_sh(d)) LSR_STAGE("preprocess_sh", s, launch_preprocess_sh(d, in, geom, radii, fs, seg, s));
    else LSR_STAGE("preprocess", s, launch_preprocess(d, in, geom, radii, fs, seg, s));
    if (fold && mapped) LSR_HIP(hipEventRecord(hw->event, s));
    int rc = sh_forward_inline(d, in, geom, s);   // view-dependent payload of calls the fused kernel does not cover
    if (rc) return rc;
    if (!fold) {
        LSR_STAGE("tile_scan", s, launch_tile_scan(d, geom, mapped ? hw->dev : nullptr, seq, capacity, s));
        if (mapped) LSR_HIP(hipEventRecord(hw->event, s));
    }
    return LSR_OK;
}
}  // namespace

// lsr_forward_front: the front half launched ahead of the full call (ABI v10).  What the full call has to find again:
namespace {
struct PendingFront { const void *geom = nullptr; const void *radii = nullptr; int64_t capacity = 0; uint32_t seq = 0; hipStream_t stream = nullptr; };
thread_local PendingFront g_front;
// the full call's half of the handshake: true + the sequence number of the pending front half, which is consumed
bool take_front(const void *geom, const void *radii, int64_t capacity, hipStream_t s, uint32_t &seq) {
    const PendingFront f = g_front;
    g_front = PendingFront();
    if (!f.geom || f.geom != geom || f.radii != radii || f.capacity != capacity || f.stream != s) return false;
    seq = f.seq;
    return true;
}
}  // namespace

int lsr_forward_front(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, int32_t *radii, int64_t pair_capacity,
                      lsr_stream_t stream) {
    g_last_hip_error = 0;
    g_front = PendingFront();
    int rc = check_dims(d);
    if (rc) return rc;
    rc = check_inputs(d, in);
    if (rc) return rc;
    if (!geom_ws) return LSR_ENULL;
    if (d->num_gaussians > 0 && !radii) return LSR_ENULL;
    if (pair_capacity < 1 || pair_capacity >= ((int64_t)1 << 32)) return LSR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    HostWords hw;
    host_words(hw);
    const uint32_t seq = next_seq();
    rc = launch_front(*d, *in, (char *)geom_ws, radii, &hw, seq, (uint32_t)pair_capacity, s);
    if (rc) return rc;
    g_front.geom = geom_ws; g_front.radii = radii; g_front.capacity = pair_capacity; g_front.seq = seq; g_front.stream = s;
    return LSR_OK;
}

int lsr_forward_prepare(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, int32_t *radii,
                        int64_t *num_pairs_host, int32_t *max_tile_pairs_host, lsr_stream_t stream) {
    g_last_hip_error = 0;
    int rc = check_dims(d);
    if (rc) return rc;
    rc = check_inputs(d, in);
    if (rc) return rc;
    if (!geom_ws || !num_pairs_host || !max_tile_pairs_host) return LSR_ENULL;
    if (d->num_gaussians > 0 && !radii) return LSR_ENULL;
    hipStream_t s = (hipStream_t)stream;
    char *geom = (char *)geom_ws;
    // Pair count and longest list come back through a 64-byte pinned, device-mapped host buffer (one per host
    // thread, allocated on first use): wait_pair_count above.  Without mapped memory: a device-to-host copy of the
    // header behind the scan.
    // Single-pass binning when the dims allow it (segment_capacity): the projection kernel writes the sort keys itself.
    // A tile list longer than its key segment is no failure: lsr_forward_render then launches the scatter for exactly
    // those tiles (the host reads the longest list here anyway).
    HostWords hw;
    host_words(hw);
    const uint32_t seq = next_seq();
    rc = launch_front(*d, *in, geom, radii, &hw, seq, 0xFFFFFFFFu, s);
    if (rc) return rc;
    uint32_t hdr[2];
    rc = read_counts(*d, geom, hw, seq, s, hdr);
    if (rc) return rc;
    *num_pairs_host = (int64_t)hdr[0];
    *max_tile_pairs_host = (int32_t)hdr[1];
    if (hdr[0] == 0xFFFFFFFFu) return LSR_EUNSUPPORTED;   // more (Gaussian, tile) pairs than the 32-bit offsets address (count saturated)
    return LSR_OK;
}

int lsr_forward_render(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, void *bin_ws,
                       void *img_ws, int64_t num_pairs, int32_t max_tile_pairs,
                       const lsr_outputs *out, lsr_stream_t stream) {
    g_last_hip_error = 0;
    int rc = check_dims(d);
    if (rc) return rc;
    rc = check_inputs(d, in);
    if (rc) return rc;
    if (!geom_ws || !img_ws || !out || !out->mask || !out->depth) return LSR_ENULL;
    if (num_pairs > 0 && !bin_ws) return LSR_ENULL;
    if (d->color_mode != LSR_COLOR_NONE && !out->color) return LSR_ENULL;
    if (d->feat_channels > 0 && !out->feature) return LSR_ENULL;
    if (num_pairs < 0 || max_tile_pairs < 0) return LSR_EINVAL;
    if ((d->forward_flags & LSR_FWD_CLEARS_GRAD) && d->num_gaussians > 0 && !out->grad_ws) return LSR_ENULL;
    hipStream_t s = (hipStream_t)stream;
    // binning + compositing.  Single-pass binning iff lsr_forward_prepare used it (the same pure function of the dims);
    // `max_tile_pairs` tells launch_binning whether any tile outgrew its key segment and needs the fallback scatter
    const bool seg = segment_capacity(*d) != 0u;
    return forward_tail(*d, *in, (char *)geom_ws, (char *)bin_ws, (char *)img_ws, num_pairs, max_tile_pairs, *out, s, false, seg);
}

int lsr_forward_abandon(lsr_stream_t stream) {
    (void)stream;   // ABI v6 joined the library's side stream here; since v7 every launch of a forward is on the caller's stream
    g_last_hip_error = 0;
    return LSR_OK;
}

int lsr_forward_nosync(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, void *bin_ws, void *img_ws,
                       int64_t pair_capacity, int32_t max_tile_hint, const lsr_outputs *out, lsr_stream_t stream) {
    g_last_hip_error = 0;
    int rc = check_dims(d);
    if (rc) return rc;
    rc = check_inputs(d, in);
    if (rc) return rc;
    if (!geom_ws || !bin_ws || !img_ws || !out || !out->mask || !out->depth) return LSR_ENULL;
    if (d->num_gaussians > 0 && !out->radii) return LSR_ENULL;
    if (d->color_mode != LSR_COLOR_NONE && !out->color) return LSR_ENULL;
    if (d->feat_channels > 0 && !out->feature) return LSR_ENULL;
    if (pair_capacity < 1 || pair_capacity >= ((int64_t)1 << 32) || max_tile_hint < 0) return LSR_EINVAL;
    if ((d->forward_flags & LSR_FWD_CLEARS_GRAD) && d->num_gaussians > 0 && !out->grad_ws) return LSR_ENULL;
    hipStream_t s = (hipStream_t)stream;
    char *geom = (char *)geom_ws;
    // the same stage sequence as prepare + render; nothing between the launches waits for the device.  Single-pass
    // binning whenever the dims allow it; the fallback scatter for tiles that outgrow their key segments is always
    // launched here (its workgroups leave at once when the device's longest list fits)
    if (d->forward_flags & LSR_FWD_FRONT_DONE) {       // (launched by lsr_forward_front: the counts also went to the host words, unread)
        uint32_t seq;
        if (!take_front(geom_ws, out->radii, pair_capacity, s, seq)) return LSR_EINVAL;
    } else {
        rc = launch_front(*d, *in, geom, out->radii, nullptr, 0u, (uint32_t)pair_capacity, s);
        if (rc) return rc;
    }
    return forward_tail(*d, *in, geom, (char *)bin_ws, (char *)img_ws, pair_capacity, max_tile_hint, *out, s, true, segment_capacity(*d) != 0u);
}

int lsr_forward_speculative(const lsr_dims *d, const lsr_inputs *in, void *geom_ws, void *bin_ws, void *img_ws,
                            int64_t pair_capacity, int32_t max_tile_hint, const lsr_outputs *out,
                            int64_t *num_pairs_host, int32_t *max_tile_pairs_host, int32_t *overflow_host, lsr_stream_t stream) {
    g_last_hip_error = 0;
    int rc = check_dims(d);
    if (rc) return rc;
    rc = check_inputs(d, in);
    if (rc) return rc;
    if (!geom_ws || !bin_ws || !img_ws || !out || !out->mask || !out->depth || !num_pairs_host || !max_tile_pairs_host || !overflow_host) return LSR_ENULL;
    if (d->num_gaussians > 0 && !out->radii) return LSR_ENULL;
    if (d->color_mode != LSR_COLOR_NONE && !out->color) return LSR_ENULL;
    if (d->feat_channels > 0 && !out->feature) return LSR_ENULL;
    if (pair_capacity < 1 || pair_capacity >= ((int64_t)1 << 32) || max_tile_hint < 1) return LSR_EINVAL;
    if ((d->forward_flags & LSR_FWD_CLEARS_GRAD) && d->num_gaussians > 0 && !out->grad_ws) return LSR_ENULL;
    hipStream_t s = (hipStream_t)stream;
    char *geom = (char *)geom_ws;
    // Everything is launched at once with the launch structure of the SYNCHRONOUS forward for (pair_capacity,
    // max_tile_hint) — sort tiers, fallback scatter — and only then does the host wait for the counts: the device goes on
    // sorting and compositing meanwhile (the synchronous forward idles ~10 us per call between the host's read and the
    // arrival of its next launches).  Lists beyond the hint are left out of the render lists (binning.hip n_limit), offsets
    // beyond the capacity are clamped by the scan: a scene that needs more than it was given renders garbage-free but
    // incomplete images and is reported as overflow.
    HostWords hw;
    host_words(hw);
    uint32_t seq;
    if (d->forward_flags & LSR_FWD_FRONT_DONE) {
        if (!take_front(geom_ws, out->radii, pair_capacity, s, seq)) return LSR_EINVAL;
    } else {
        seq = next_seq();
        rc = launch_front(*d, *in, geom, out->radii, &hw, seq, (uint32_t)pair_capacity, s);
        if (rc) return rc;
    }
    rc = forward_tail(*d, *in, geom, (char *)bin_ws, (char *)img_ws, pair_capacity, max_tile_hint, *out, s, false,
                      segment_capacity(*d) != 0u, true);
    if (rc) return rc;
    uint32_t hdr[2];
    rc = read_counts(*d, geom, hw, seq, s, hdr);
    if (rc) return rc;
    *num_pairs_host = (int64_t)hdr[0];
    *max_tile_pairs_host = (int32_t)hdr[1];
    *overflow_host = ((int64_t)hdr[0] > pair_capacity || hdr[1] > (uint32_t)max_tile_hint || hdr[0] == 0xFFFFFFFFu) ? 1 : 0;
    return LSR_OK;
}

int lsr_forward_status(const lsr_dims *d, const void *geom_ws, int64_t *num_pairs_host, int32_t *max_tile_pairs_host,
                       int32_t *overflow_host, lsr_stream_t stream) {
    g_last_hip_error = 0;
    int rc = check_dims(d);
    if (rc) return rc;
    if (!geom_ws || !num_pairs_host || !max_tile_pairs_host || !overflow_host) return LSR_ENULL;
    uint32_t hdr[8] = {};
    hipStream_t s = (hipStream_t)stream;
    LSR_HIP(hipMemcpyAsync(hdr, (const char *)geom_ws + geom_layout(*d).header, sizeof(hdr), hipMemcpyDeviceToHost, s));
    LSR_HIP(hipStreamSynchronize(s));
    *num_pairs_host = (int64_t)hdr[kHdrPairs];
    *max_tile_pairs_host = (int32_t)hdr[kHdrMaxTile];
    *overflow_host = (int32_t)hdr[kHdrOverflow];
    return LSR_OK;
}

int lsr_backward(const lsr_dims *d, const lsr_inputs *in, const void *geom_ws, const void *bin_ws,
                 const void *img_ws, int64_t num_pairs, const int32_t *radii, const lsr_outputs *fwd,
                 const lsr_out_grads *gout, void *grad_ws, const lsr_in_grads *gin,
                 lsr_stream_t stream) {
    g_last_hip_error = 0;
    int rc = check_dims(d);
    if (rc) return rc;
    rc = check_inputs(d, in);
    if (rc) return rc;
    if (!gout || !gin || !geom_ws || !img_ws || !grad_ws || !fwd) return LSR_ENULL;
    if (d->num_gaussians == 0) return LSR_OK;
    if ((gout->color && d->color_mode != LSR_COLOR_NONE && !fwd->color) || (gout->feature && d->feat_channels > 0 && !fwd->feature) ||
        (gout->depth && !fwd->depth))
        return LSR_ENULL;
    if (!radii || !gin->means3D || !gin->cov3D || !gin->opacities) return LSR_ENULL;
    if (d->color_mode != LSR_COLOR_NONE && !gin->color) return LSR_ENULL;
    if (d->feat_channels > 0 && !gin->features) return LSR_ENULL;
    if (num_pairs > 0 && !bin_ws) return LSR_ENULL;
    hipStream_t s = (hipStream_t)stream;
    // zero the packed gradient records the compositing backward accumulates into (unless the forward did: LSR_FWD_CLEARS_GRAD)
    if (!(d->forward_flags & LSR_FWD_CLEARS_GRAD)) LSR_HIP(launch_clear_grad(*d, radii, (char *)grad_ws, s));
    if (num_pairs > 0)
        LSR_STAGE("render_backward", s, launch_render_backward(*d, *in, (const char *)geom_ws, (const char *)bin_ws, num_pairs,
                                       (const char *)img_ws, *fwd, *gout, (char *)grad_ws, *gin, s));
    // Geometry and SH backward: one launch each, all scenes (view groups) of the call at once (the scene in
    // blockIdx.y).  Both add into the scene's mean gradients, hence in stream order.
    LSR_STAGE("preprocess_backward", s, launch_preprocess_backward(*d, *in, (const char *)geom_ws, radii, (const char *)grad_ws, *gin, s));
    LSR_STAGE("sh_backward", s, launch_sh_backward(*d, *in, (const char *)geom_ws, (const char *)grad_ws, *gin, s));
    return LSR_OK;
}

}  // extern "C"
