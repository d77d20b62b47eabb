// sh.hip — view-dependent payload channels from spherical harmonics, forward and backward.
//
//   colour   : rgb = max(0, 0.5 + sum_k B_k(dir) * shs[k][c])            (kernel axis convention,
//              degree <= 4)                                               lsr_sh.h)
//   features : f_c = 0.5 + sum_k B^ref_k(dir) * feature_sh[c][k]          (reference convention,
//              the latent-feature evaluation the reference does in PyTorch,
//              /root/reference/src/model/decoder/cuda_splatting.py:94-101; fused here for
//              degree <= 2, where B^ref(x,y,z) == B(z,x,y))
//   dir = normalize(mean * scene_scale - campos)
//
// Both kernels are HBM-bound gathers over (G, K*3) / (G, C*Kf) coefficient tensors, so the design
// is about touching every byte once with wide accesses:
//  * a 256-thread workgroup owns 64 consecutive Gaussians for ALL views and BOTH coefficient
//    groups; the groups' blocks are contiguous in memory (64 x ks floats) and are staged by all four
//    waves as flat copies with the direct global->LDS path (no staging registers); lane l reads its
//    row at word stride ks (conflict-free for the odd colour strides); a scene shared by all
//    views (stride 0) is staged once and the waves take different views;
//  * forward: the payload channels of a record are written as whole 16-byte words (the upper half
//    of the 64-byte record line), the "colour channel was clamped at 0" flags go to a dense byte
//    array (1 B per (view, Gaussian)) so the backward never has to re-read the records;
//  * backward: ONE kernel.  Thread (wave = view of a chunk of 4, lane = Gaussian) reads its
//    gradient record once, computes both SH bases, the direction -> mean gradient (needs the staged
//    coefficients) and leaves bases + channel gradients in LDS (re-using the coefficient area);
//    then thread = element of the contiguous 64*ks coefficient-gradient runs sums over the views
//    of the chunk and stores lane-contiguously (no read-modify-write unless a shared scene has
//    more views than one chunk).
// The colour degree is a template parameter (straight-line band code, no dead registers).
#include <algorithm>
#include <type_traits>

#include "lsr_blend.h"
#include "lsr_project.h"
#include "lsr_sh.h"
#include "lsr_tile_scan.h"

namespace lsr {

constexpr int kShThreads = 256;
constexpr int kShWaves = kShThreads / LSR_WAVE;   // = views per chunk of the backward
constexpr int kShBasisC = 27;   // 25 basis values + a zero slot, odd stride
constexpr int kShBasisF = 11;   // 9 + zero slot, odd stride

struct ShParams {
    lsr_dims d;
    lsr_inputs in;
    const float *vis;         // depth word of the (view, Gaussian) bin records: > 0 <=> visible
    int vis_stride;           // in floats (3 narrow, 4 wide records)
    float *rec;               // screen-space records (forward writes the payload slots)
    uint8_t *clamp;           // [V*G] bit c set: colour channel c clamped at 0
    const float *grec;        // backward: packed gradient records
    int RF;
    lsr_in_grads g;
    int has[2];               // group enabled: 0 = colour (G,K,3)|(G,3,K), 1 = features (G,C,Kf)
    int ks[2];                // floats per Gaussian of each group (0 when disabled)
    int offF;                 // LDS word offset of the feature rows (16-byte aligned)
    uint32_t mdiv[2];         // ceil(2^22 / ks): x / ks == (x * mdiv) >> 22 for x < 8192
    uint32_t mdivK, mdivKf;   // same for K (coefficients per colour channel) and Kf
    GroupStrides gs;          // view groups: blockIdx.y = group, every pointer above advances by the group's slices
};
// the launch's parameters as seen by view group blockIdx.y (uniform: scalar arithmetic)
__device__ __forceinline__ ShParams group_params(const ShParams &pk) {
    ShParams p = pk;
    const int64_t g = blockIdx.y;
    p.in.views += g * pk.gs.views; p.in.means3D += g * pk.gs.means;
    if (p.in.cov3D) p.in.cov3D += g * pk.gs.cov;
    if (p.in.opacities) p.in.opacities += g * pk.gs.opac;
    if (p.in.color) p.in.color += g * pk.gs.color;
    if (p.in.features) p.in.features += g * pk.gs.feat;
    p.vis += g * pk.gs.slots * pk.vis_stride; p.rec += g * pk.gs.slots * pk.RF; p.clamp += g * pk.gs.slots;
    if (p.grec) p.grec += g * pk.gs.slots * pk.RF;
    if (p.g.means3D) p.g.means3D += g * pk.gs.means;
    if (p.g.color) p.g.color += g * pk.gs.color;
    if (p.g.features) p.g.features += g * pk.gs.feat;
    return p;
}
__device__ __forceinline__ int udiv_small(int x, uint32_t m) { return (int)(((uint32_t)x * m) >> 22); }

// Stage group `grp` of view v (rows x ks floats, one contiguous 16-byte aligned run) as a flat copy
// into its LDS region with the direct global->LDS path (global_load_lds_dwordx4: no staging VGPRs,
// destination = wave-uniform base + lane*16, i.e. the source order).  Lane l later reads its row at
// word stride ks: conflict-free for odd ks (colour: 3*K with K = 1, 9, 25), a few-way otherwise.
__device__ __forceinline__ void stage_group(float *lds, const ShParams &p, int grp, int v, int g0, int rows, int tid) {
    const int ks = p.ks[grp];
    if (!ks) return;
    const int64_t vs = grp == 0 ? p.d.vs_color : p.d.vs_feat;
    const float *src = (grp == 0 ? p.in.color : p.in.features) + (size_t)v * vs + (size_t)g0 * ks;
    float *dst = lds + (grp == 0 ? 0 : p.offF);
    // the 16-byte global->LDS path needs a 16-byte aligned source run: per-view slices are only
    // aligned when G*ks is a multiple of 4 floats, and a caller's slice (shs[i]) may start anywhere —
    // otherwise the whole run takes the scalar loop below (block-uniform)
    const int n = rows * ks, n4 = (((size_t)src & 15u) == 0) ? n >> 2 : 0;
    const int lane = tid & (LSR_WAVE - 1);
    for (int t4 = tid; t4 - lane < n4; t4 += kShThreads) {
        if (t4 < n4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * (size_t)t4),
                                             (__attribute__((address_space(3))) void *)(dst + 4 * (size_t)(t4 - lane)), 16, 0, 0);
    }
    for (int t = (n4 << 2) + tid; t < n; t += kShThreads) dst[t] = src[t];
}
// all outstanding direct loads have landed in LDS, then the workgroup barrier
__device__ __forceinline__ void staged_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// sum_k bas[k] * co[k * stride] over the bands up to `deg` (<= MAXDEG), ascending k (oracle order)
// (compile-time stride: the LDS reads become one address register + immediate offsets)
template <int MAXDEG, int stride>
__device__ __forceinline__ float sh_dot(const float *bas, const float *co, int deg) {
    float a = bas[0] * co[0];
    if (MAXDEG > 0 && deg > 0) {
#pragma unroll
        for (int k = 1; k < 4; ++k) a += bas[k] * co[k * stride];
        if (MAXDEG > 1 && deg > 1) {
#pragma unroll
            for (int k = 4; k < 9; ++k) a += bas[k] * co[k * stride];
            if (MAXDEG > 2 && deg > 2) {
#pragma unroll
                for (int k = 9; k < 16; ++k) a += bas[k] * co[k * stride];
                if (MAXDEG > 3 && deg > 3) {
#pragma unroll
                    for (int k = 16; k < 25; ++k) a += bas[k] * co[k * stride];
                }
            }
        }
    }
    return a;
}

struct ShDir {
    float dx, dy, dz, len, sc;
};
struct ShPos { float x, y, z; };
typedef const float __attribute__((address_space(4))) *kfloat_ptr;   // constant address space: uniform loads become s_load
__device__ __forceinline__ ShPos sh_position(const ShParams &p, int v, int i) {
    const float *mp = p.in.means3D + (size_t)v * p.d.vs_means + 3 * (size_t)i;
    return ShPos{mp[0], mp[1], mp[2]};
}
// v is wave-uniform in both kernels: scene scale and camera position come through the scalar cache
__device__ __forceinline__ ShDir sh_direction(const ShParams &p, int v, const ShPos &m) {
    const kfloat_ptr vw = (kfloat_ptr)(p.in.views + (size_t)v * LSR_VIEW_FLOATS);
    ShDir r;
    r.sc = vw[40];
    float dx = m.x * r.sc - vw[32], dy = m.y * r.sc - vw[33], dz = m.z * r.sc - vw[34];
    r.len = sqrtf(dx * dx + dy * dy + dz * dz);
    r.dx = dx / r.len; r.dy = dy / r.len; r.dz = dz / r.len;
    return r;
}
__device__ __forceinline__ ShDir sh_direction(const ShParams &p, int v, int i) { return sh_direction(p, v, sh_position(p, v, i)); }

__device__ __forceinline__ bool sh_shared_scene(const ShParams &p) {
    return (!p.has[0] || p.d.vs_color == 0) && (!p.has[1] || p.d.vs_feat == 0);
}

// ------------------------------------------------------------------------------------------
// DEGC: colour SH degree, -1 = colour is not evaluated here.  COFF: payload slot of feature 0.
template <int DEGC, int COFF>
__global__ void __launch_bounds__(kShThreads)
k_sh_fwd(ShParams pk) {
    const ShParams p = group_params(pk);
    extern __shared__ float s_lds[];
    const lsr_dims &d = p.d;
    const int tid = threadIdx.x, lane = tid & (LSR_WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / LSR_WAVE);
    const int G = d.num_gaussians, V = d.num_views;
    const int g0 = blockIdx.x * LSR_WAVE, i = g0 + lane;
    const int rows = G - g0 < LSR_WAVE ? G - g0 : LSR_WAVE;
    const bool active = i < G;
    const int C = d.feat_channels, K = d.sh_coeffs, Kf = d.feat_sh_coeffs, degF = d.feat_sh_degree;
    const bool hasF = p.has[1] != 0;
    const bool cmaj = d.color_sh_channel_major != 0;
    const bool cax = d.color_sh_convention == LSR_SH_AXES_REFERENCE;   // colour basis evaluated at (z,x,y)
    const float *my = s_lds + lane * p.ks[0];
    const float *myF = s_lds + p.offF + lane * p.ks[1];

    // visibility and position are requested by `fetch` (unconditionally, before the wait for the staged
    // coefficients where possible), `compute` consumes them
    struct Fetched { float visf; ShPos pos; };
    auto fetch = [&](int v) {
        const int vl = v < V ? v : V - 1;
        const size_t o = (size_t)vl * G + (active ? i : 0);
        return Fetched{p.vis[o * p.vis_stride], sh_position(p, vl, active ? i : 0)};
    };
    auto compute = [&](int v, const Fetched &f) {
        const size_t o = (size_t)v * G + (active ? i : 0);
        if (!(active && f.visf > 0.0f)) return;
        const ShDir dir = sh_direction(p, v, f.pos);
        float *R = p.rec + o * (size_t)p.RF + 8;
        float basF[9];
        // features use the reference's axis naming: B^ref(x,y,z) = B(z,x,y) up to degree 2
        if (hasF) sh_basis<2>(degF, dir.dz, dir.dx, dir.dy, basF);
        auto feat = [&](int c) -> float {               // feature channel c, 0 beyond the last one (padding)
            if (c >= C) return 0.0f;
            return sh_dot<2, 1>(basF, myF + c * Kf, degF) + 0.5f;
        };
        int c_next = 0;                                 // first feature channel not yet written
        if (COFF == 3) {
            float col[3] = {0.0f, 0.0f, 0.0f};
            if (DEGC >= 0) {
                float basC[25];
                sh_basis<DEGC>(DEGC, cax ? dir.dz : dir.dx, cax ? dir.dx : dir.dy, cax ? dir.dy : dir.dz, basC);   // LSR_SH_AXES_REFERENCE: B(z,x,y)
                uint32_t bits = 0;
                // one channel at a time, each consuming its LDS reads before the next starts (fully
                // interleaved, the 75 reads cost ~60 more live registers and a wave of occupancy)
                float a0, a1, a2;
                if (cmaj) {
                    a0 = sh_dot<DEGC, 1>(basC, my, DEGC) + 0.5f;
                    __builtin_amdgcn_sched_barrier(0);
                    a1 = sh_dot<DEGC, 1>(basC, my + K, DEGC) + 0.5f;
                    __builtin_amdgcn_sched_barrier(0);
                    a2 = sh_dot<DEGC, 1>(basC, my + 2 * K, DEGC) + 0.5f;
                } else {
                    a0 = sh_dot<DEGC, 3>(basC, my, DEGC) + 0.5f;
                    __builtin_amdgcn_sched_barrier(0);
                    a1 = sh_dot<DEGC, 3>(basC, my + 1, DEGC) + 0.5f;
                    __builtin_amdgcn_sched_barrier(0);
                    a2 = sh_dot<DEGC, 3>(basC, my + 2, DEGC) + 0.5f;
                }
                if (a0 < 0.0f) bits |= 1u;
                if (a1 < 0.0f) bits |= 2u;
                if (a2 < 0.0f) bits |= 4u;
                col[0] = a0 > 0.0f ? a0 : 0.0f; col[1] = a1 > 0.0f ? a1 : 0.0f; col[2] = a2 > 0.0f ? a2 : 0.0f;
                p.clamp[o] = (uint8_t)bits;
            }
            if (DEGC >= 0 && hasF) {
                *(float4 *)R = make_float4(col[0], col[1], col[2], feat(0));
            } else if (DEGC >= 0) {
                R[0] = col[0]; R[1] = col[1]; R[2] = col[2];
            } else if (hasF) {
                R[3] = feat(0);
            }
            c_next = 1;
        }
        if (hasF) {
            const int pad = ((COFF + C + 3) & ~3) - COFF;   // feature channels incl. the zero padding
#pragma unroll 1
            for (int c = c_next; c < pad; c += 4)
                *(float4 *)(R + COFF + c) = make_float4(feat(c), feat(c + 1), feat(c + 2), feat(c + 3));
        }
    };

    if (sh_shared_scene(p)) {
        stage_group(s_lds, p, 0, 0, g0, rows, tid);
        stage_group(s_lds, p, 1, 0, g0, rows, tid);
        Fetched f = fetch(wave);
        staged_barrier();
        for (int v = wave; v < V; v += kShWaves) {
            const Fetched nxt = fetch(v + kShWaves);   // next view's reads in flight during this one's arithmetic
            compute(v, f);
            f = nxt;
        }
    } else {
        for (int v = 0; v < V; ++v) {
            __syncthreads();
            stage_group(s_lds, p, 0, v, g0, rows, tid);
            stage_group(s_lds, p, 1, v, g0, rows, tid);
            const Fetched f = fetch(v);
            staged_barrier();
            if (wave == (v & (kShWaves - 1))) compute(v, f);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Stage K1 and the SH payload pass as ONE kernel (round 4), for calls whose payload is evaluated from harmonics only
// (colour SH and / or latent-feature SH: the reference's configs[3] / [4] shape) and whose views share their inputs
// (one scene, or view groups).  The two-kernel form wrote every visible record twice — k_preprocess the whole 64-byte
// line, k_sh_fwd the payload half of it: a PARTIAL line write that the memory system turns into a read-modify-write
// once the line has left the Infinity Cache (16 views x 393 216 records = 403 MB at configs[4]: k_sh_fwd ran at 0.26 of
// the HBM peak there) — and read the means and the visibility words a second time.  Here thread (wave = view, lane =
// Gaussian) projects its Gaussian (lsr_project.h: the same IEEE sequence as k_preprocess, bit for bit), evaluates the
// harmonics from the coefficient rows its workgroup staged in LDS and stores the finished record, one whole line.
// A workgroup walks `chunks` consecutive 64-Gaussian chunks so that its LDS-privatised tile histogram is flushed once
// per 512 Gaussians (as in k_preprocess); the last workgroup to finish runs the tile scan (lsr_tile_scan.h).
struct PreShArgs {
    char *binrec; int narrow;
    int32_t *radii;
    uint32_t *tile_count, *header;
    FoldedScan fs;
    int chunks;         // 64-Gaussian chunks per workgroup
    int lds_hist;       // tile histogram of the group's views privatised in LDS (behind the coefficient rows)
    int hist_off;       // its offset in the dynamic LDS allocation, in words
    int views_total;    // views of the whole call (the scan's count array)
    SegOut seg;         // single-pass binning (SEG instances): the key segments; the coefficient area doubles as the bucket array
};

#ifndef LSR_PSH_WAVES
#define LSR_PSH_WAVES 1
#endif
template <int DEGC, int COFF, bool FMA, bool SEG>
__global__ void __launch_bounds__(kShThreads, LSR_PSH_WAVES)
k_preprocess_sh(ShParams pk, PreShArgs a) {
    const ShParams p = group_params(pk);
    extern __shared__ float s_lds[];
    __shared__ float2 s_focal[64];
    const lsr_dims &d = p.d;
    const int tid = threadIdx.x, lane = tid & (LSR_WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / LSR_WAVE);
    const int G = d.num_gaussians, V = d.num_views;        // the GROUP's views (all views of a call without groups)
    const int gx = (d.width + LSR_TILE - 1) / LSR_TILE, gy = (d.height + LSR_TILE - 1) / LSR_TILE, T = gx * gy;
    const int C = d.feat_channels, K = d.sh_coeffs, Kf = d.feat_sh_coeffs, degF = d.feat_sh_degree;
    const bool hasF = p.has[1] != 0;
    const bool cmaj = d.color_sh_channel_major != 0;
    const bool cax = d.color_sh_convention == LSR_SH_AXES_REFERENCE;
    const int ce = d.cov_elems;
    const int RF = p.RF;
    const bool skip_none = reached_only(d);
    const size_t slot0 = (size_t)blockIdx.y * pk.gs.slots;             // first (view, Gaussian) slot of this group
    const int view0 = blockIdx.y * (pk.gs.slots ? V : 0);               // first view of this group in the call
    int32_t *radii = a.radii + slot0;
    char *binrec = a.binrec + slot0 * (a.narrow ? sizeof(BinRec) : sizeof(BinRecWide));
    uint32_t *tile_count = a.tile_count + (size_t)view0 * T;
    uint32_t *s_hist = (uint32_t *)s_lds + a.hist_off;
    const float *my = s_lds + lane * p.ks[0];
    const float *myF = s_lds + p.offF + lane * p.ks[1];
    // per-view constants that cost an IEEE division each: once per workgroup
    for (int v = tid; v < V && v < 64; v += kShThreads) {
        const float *vw = p.in.views + (size_t)v * LSR_VIEW_FLOATS;
        s_focal[v] = make_float2(d.width / (2.0f * vw[35]), d.height / (2.0f * vw[36]));
    }
    if (a.lds_hist)
        for (int t = tid; t < V * T; t += kShThreads) s_hist[t] = 0;

    for (int c = 0; c < a.chunks; ++c) {
        const int g0 = (blockIdx.x * a.chunks + c) * LSR_WAVE;
        if (g0 >= G) break;                                   // block-uniform
        const int i = g0 + lane;
        const int rows = G - g0 < LSR_WAVE ? G - g0 : LSR_WAVE;
        const bool active = i < G;
        const size_t ii = active ? (size_t)i : 0;
        __syncthreads();                                      // the previous chunk is done with the coefficient rows (first pass: s_focal, histogram cleared)
        stage_group(s_lds, p, 0, 0, g0, rows, tid);
        stage_group(s_lds, p, 1, 0, g0, rows, tid);
        // the Gaussian, once for all views of this lane's waves (the four waves read the same 64 rows: cache hits)
        const float *mp = p.in.means3D + 3 * ii;
        const float q0 = mp[0], q1 = mp[1], q2 = mp[2];
        const float *c6 = p.in.cov3D + (size_t)ce * ii;
        const float r0 = c6[0], r1 = c6[1], r2 = c6[2];
        const float r3 = c6[ce == 9 ? 4 : 3], r4 = c6[ce == 9 ? 5 : 4], r5 = c6[ce == 9 ? 8 : 5];
        const float opacity = p.in.opacities[ii];
        staged_barrier();
        for (int v = wave; v < V; v += kShWaves) {
            const kfloat_ptr vw = (kfloat_ptr)(p.in.views + (size_t)v * LSR_VIEW_FLOATS);
            float vm[16], pm[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { vm[k] = vw[k]; pm[k] = vw[16 + k]; }
            const float limx = 1.3f * vw[35], limy = 1.3f * vw[36];
            const float2 focal = s_focal[v < 64 ? v : 0];
            const float focal_x = v < 64 ? focal.x : d.width / (2.0f * vw[35]), focal_y = v < 64 ? focal.y : d.height / (2.0f * vw[36]);
            const float scale = vw[40], scale2 = scale * scale;
            const float p0 = q0 * scale, p1 = q1 * scale, p2 = q2 * scale;
            const Projected pj = project_gaussian<FMA>(vm, pm, limx, limy, focal_x, focal_y, p0, p1, p2, r0 * scale2, r1 * scale2, r2 * scale2,
                                                       r3 * scale2, r4 * scale2, r5 * scale2, d.width, d.height, gx, gy, active);
            const bool ok = pj.ok;
            const size_t o = (size_t)v * G + ii;
            const uint32_t span = ok ? footprint_cells(pj.px, pj.py, pj.conic_a, pj.conic_b, pj.conic_c, opacity, pj.rminx, pj.rminy) : kSpanNone;
            if (ok) {
                uint32_t *hist = a.lds_hist ? s_hist + v * T : tile_count + (size_t)v * T;
                int hx0 = pj.rminx, hy0 = pj.rminy, hx1 = pj.rmaxx, hy1 = pj.rmaxy;
                if (skip_none) reached_rect(span, hx0, hy0, hx1, hy1);      // LSR_FWD_REACHED_ONLY
                for (int y = hy0; y < hy1; ++y)
                    for (int x = hx0; x < hx1; ++x) atomicAdd(&hist[y * gx + x], 1u);
            }
            if (active) {
                radii[o] = ok ? (int32_t)pj.radius : 0;
                const float out_depth = ok ? pj.tz : 0.0f;
                if (a.narrow) {
                    BinRec br;
                    br.rect = ok ? ((uint32_t)pj.rminx | ((uint32_t)pj.rminy << 8) | ((uint32_t)pj.rmaxx << 16) | ((uint32_t)pj.rmaxy << 24)) : 0u;
                    br.depth = out_depth; br.span = span;
                    ((BinRec *)binrec)[o] = br;
                } else {
                    BinRecWide br;
                    br.rect = ok ? make_ushort4((unsigned short)pj.rminx, (unsigned short)pj.rminy, (unsigned short)pj.rmaxx, (unsigned short)pj.rmaxy)
                                 : make_ushort4(0, 0, 0, 0);
                    br.depth = out_depth; br.span = span;
                    ((BinRecWide *)binrec)[o] = br;
                }
            }
            if (!ok) continue;
            // ---- payload from the harmonics (the arithmetic of k_sh_fwd, operation for operation) ----
            const ShDir dir = sh_direction(p, v, ShPos{q0, q1, q2});
            float basF[9];
            if (hasF) sh_basis<2>(degF, dir.dz, dir.dx, dir.dy, basF);
            auto feat = [&](int ch) -> float {
                if (ch >= C) return 0.0f;
                return sh_dot<2, 1>(basF, myF + ch * Kf, degF) + 0.5f;
            };
            float col[3] = {0.0f, 0.0f, 0.0f};
            if (DEGC >= 0) {
                float basC[25];
                sh_basis<DEGC>(DEGC, cax ? dir.dz : dir.dx, cax ? dir.dx : dir.dy, cax ? dir.dy : dir.dz, basC);
                uint32_t bits = 0;
                float a0, a1, a2;
                if (cmaj) {
                    a0 = sh_dot<DEGC, 1>(basC, my, DEGC) + 0.5f;
                    __builtin_amdgcn_sched_barrier(0);
                    a1 = sh_dot<DEGC, 1>(basC, my + K, DEGC) + 0.5f;
                    __builtin_amdgcn_sched_barrier(0);
                    a2 = sh_dot<DEGC, 1>(basC, my + 2 * K, DEGC) + 0.5f;
                } else {
                    a0 = sh_dot<DEGC, 3>(basC, my, DEGC) + 0.5f;
                    __builtin_amdgcn_sched_barrier(0);
                    a1 = sh_dot<DEGC, 3>(basC, my + 1, DEGC) + 0.5f;
                    __builtin_amdgcn_sched_barrier(0);
                    a2 = sh_dot<DEGC, 3>(basC, my + 2, DEGC) + 0.5f;
                }
                if (a0 < 0.0f) bits |= 1u;
                if (a1 < 0.0f) bits |= 2u;
                if (a2 < 0.0f) bits |= 4u;
                col[0] = a0 > 0.0f ? a0 : 0.0f; col[1] = a1 > 0.0f ? a1 : 0.0f; col[2] = a2 > 0.0f ? a2 : 0.0f;
                p.clamp[o] = (uint8_t)bits;
            }
            float4 *R = (float4 *)(p.rec + o * (size_t)RF);
            R[0] = make_float4(pj.px, pj.py, pj.conic_a, pj.conic_b);
            R[1] = make_float4(pj.conic_c, opacity, pj.tz, 0.0f);
            // payload slots 8.. : rgb (COFF = 3) then the feature channels, zero padded to the record
            auto slot = [&](int ch) -> float { return ch < COFF ? col[ch < 3 ? ch : 0] : (hasF ? feat(ch - COFF) : 0.0f); };
#pragma unroll 1
            for (int j = 2; j < RF / 4; ++j) {
                const int ch = 4 * (j - 2);
                R[j] = make_float4(slot(ch), slot(ch + 1), slot(ch + 2), slot(ch + 3));
            }
        }
    }
    if (a.lds_hist && !SEG) {
        __syncthreads();
        for (int t = tid; t < V * T; t += kShThreads) {
            const uint32_t cnt = s_hist[t];
            if (cnt) atomicAdd(&tile_count[t], cnt);
        }
    }
    if (SEG) {
        // ---- single-pass binning (round 5), as in k_preprocess: reserve this workgroup's slots in the tiles' key segments
        // with the returning form of the count atomics, then emit the keys of its (view, Gaussian) items view by view
        // through an LDS bucket pass (the coefficient rows are dead by now: their area holds the buckets) ----
        __syncthreads();
        uint32_t *s_first = s_hist + V * T;
        for (int t0 = tid; t0 < V * T; t0 += 4 * kShThreads) {
            uint32_t c[4], first[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int t = t0 + k * kShThreads; c[k] = t < V * T ? s_hist[t] : 0u; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = t0 + k * kShThreads;
                if (c[k]) first[k] = __hip_atomic_fetch_add(&tile_count[t], c[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int t = t0 + k * kShThreads; if (t < V * T) s_first[t] = first[k]; }
        }
        __syncthreads();
        const uint32_t cap = a.seg.cap;
        uint32_t *s_delta = (uint32_t *)s_lds;
        const int kbase = (T * 4 + 7) & ~7;
        uint64_t *s_key = (uint64_t *)((char *)s_lds + kbase);
        const uint32_t buf = (uint32_t)(((size_t)a.hist_off * 4 - kbase) / 12);
        uint32_t *s_pos = (uint32_t *)((char *)s_lds + kbase + (size_t)buf * 8);
        __shared__ uint32_t s_scanw[kShWaves];
        const int tpt = (T + kShThreads - 1) / kShThreads;
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            uint32_t *cur = s_hist + v * T;
            uint32_t cnt[4], mine = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = tid * tpt + k;
                cnt[k] = (k < tpt && t < T) ? cur[t] : 0u;
                mine += cnt[k];
            }
            uint32_t n_v;
            uint32_t off = block_exclusive_scan<kShThreads>(mine, s_scanw, n_v);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = tid * tpt + k;
                if (k < tpt && t < T) { cur[t] = off; s_delta[t] = s_first[v * T + t] - off; off += cnt[k]; }
            }
            __syncthreads();
            const uint32_t seg0 = (uint32_t)(view0 + v) * (uint32_t)T;
            // the workgroup's Gaussians of this view: wave w takes chunks w, w + 4 (records written by whichever wave
            // projected the view: visible since the barrier above)
            uint3 br[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = min(wave + kShWaves * k, a.chunks - 1);
                const int i = min((blockIdx.x * a.chunks + c) * LSR_WAVE + lane, G - 1);
                br[k] = *(const uint3 *)(binrec + ((size_t)v * G + (size_t)i) * sizeof(BinRec));
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = wave + kShWaves * k;
                const int i = (blockIdx.x * a.chunks + c) * LSR_WAVE + lane;
                const uint32_t rc = (c < a.chunks && i < G) ? br[k].x : 0u;
                const int x0 = rc & 0xff, y0 = (rc >> 8) & 0xff, x1 = (rc >> 16) & 0xff, y1 = rc >> 24;
                const uint64_t key = ((uint64_t)br[k].y << 32) | ((uint32_t)i << a.seg.key_shift);
                const uint32_t sp = br[k].z;
                int ex0 = x0, ey0 = y0, ex1 = x1, ey1 = y1;
                if (skip_none) reached_rect(sp, ex0, ey0, ex1, ey1);
                for (int y = ey0; y < ey1; ++y)
                    for (int x = ex0; x < ex1; ++x) {
                        const int t = y * gx + x;
                        const uint32_t code = a.seg.key_shift ? span_code(sp, x - x0, y - y0) : 0u;
                        const uint32_t slot = atomicAdd(&cur[t], 1u);
                        const uint32_t pos = (seg0 + (uint32_t)t) * cap + min(slot + s_delta[t], cap - 1u);
                        if (slot < buf) { s_key[slot] = key | code; s_pos[slot] = pos; }
                        else a.seg.keys[pos] = key | code;
                    }
            }
            __syncthreads();
            const uint32_t nflush = min(n_v, buf);
            for (uint32_t j = tid; j < nflush; j += kShThreads) a.seg.keys[s_pos[j]] = s_key[j];
            __syncthreads();
        }
    }
    // ---- the tile scan in the last workgroup to arrive (as in k_preprocess; the counts pass through LDS) ----
    if (a.fs.enabled) {
        __shared__ uint32_t s_last;
        // (the scan's 4 KB of counters live in the dynamic allocation behind the staged counts, not in static LDS: static
        // bytes count against every workgroup of the launch, and 4.7 KB of them were the difference between three and four
        // workgroups per CU at the configs[3] / [4] payload)
        TileScanShared<kShThreads> &s_scan = *(TileScanShared<kShThreads> *)((char *)s_lds + (size_t)kFoldTiles * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const uint32_t arrived = __hip_atomic_fetch_add(&a.header[kHdrPreDone], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = arrived == gridDim.x * gridDim.y - 1u;
        }
        __syncthreads();
        if (s_last) {
            uint32_t *s_counts = (uint32_t *)s_lds;
            const int N = a.views_total * T;
            for (int i0 = tid; i0 < N; i0 += 4 * kShThreads) {
                uint32_t cc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    cc[k] = __hip_atomic_load(&a.tile_count[min(i0 + k * kShThreads, N - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + k * kShThreads < N) s_counts[i0 + k * kShThreads] = cc[k];
            }
            __syncthreads();
            tile_scan_block<kShThreads, 0, false>(s_counts, a.fs.tile_start, a.header, HostMirror{a.fs.host_words, a.fs.host_seq},
                                                  a.fs.tile_order, N, a.fs.capacity, s_scan);
        }
    }
}

// ------------------------------------------------------------------------------------------
template <int DEGC, int COFF>
__global__ void __launch_bounds__(kShThreads, 4)
k_sh_bwd(ShParams pk) {
    const ShParams p = group_params(pk);
    extern __shared__ float s_lds[];
    const lsr_dims &d = p.d;
    const int tid = threadIdx.x, lane = tid & (LSR_WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / LSR_WAVE);
    const int G = d.num_gaussians, V = d.num_views;
    const int g0 = blockIdx.x * LSR_WAVE, i = g0 + lane;
    const int rows = G - g0 < LSR_WAVE ? G - g0 : LSR_WAVE;
    const bool active = i < G;
    const int C = d.feat_channels, K = d.sh_coeffs, Kf = d.feat_sh_coeffs, degF = d.feat_sh_degree;
    constexpr int nbC = DEGC >= 0 ? (DEGC + 1) * (DEGC + 1) : 0;
    const int nbF = (degF + 1) * (degF + 1);
    const bool hasF = p.has[1] != 0;
    const bool cmaj = d.color_sh_channel_major != 0;
    const bool cax = d.color_sh_convention == LSR_SH_AXES_REFERENCE;   // colour basis evaluated at (z,x,y)
    const bool shared = sh_shared_scene(p);
    const int chunk = shared ? kShWaves : 1;            // per-view coefficient outputs: one view at a time
    const int ntot = COFF + C, cs = ntot | 1;
    // LDS: [ coefficient rows | re-used after the direction pass: basC, then (after the colour element pass)
    // basF in the same place ] [ gch ].  One basis array at a time keeps the block at ~36 KB for the
    // configs[3] payload (colour degree 4 + 4 x degree-2 features): four blocks per CU instead of three.
    const int area = max(p.offF + LSR_WAVE * p.ks[1], kShWaves * LSR_WAVE * kShBasisC);
    float *s_basC = s_lds, *s_basF = s_lds;
    float *s_gch = s_lds + area;
    const float *my = s_lds + lane * p.ks[0];
    const float *myF = s_lds + p.offF + lane * p.ks[1];
    float acc[3] = {0.0f, 0.0f, 0.0f};   // mean gradient summed over this thread's views (shared means)

    for (int v0 = 0; v0 < V; v0 += chunk) {
        const int nv = V - v0 < chunk ? V - v0 : chunk;
        __syncthreads();   // previous chunk's element pass is done with the LDS arrays
        stage_group(s_lds, p, 0, shared ? 0 : v0, g0, rows, tid);
        stage_group(s_lds, p, 1, shared ? 0 : v0, g0, rows, tid);
        // ---- pass 1: thread = (view v0 + wave, Gaussian lane) ----
        // Everything this thread reads from global memory is requested HERE, unconditionally (clamped view /
        // Gaussian), next to the coefficient staging: visibility, clamp bits, position and the channel
        // gradients of its record.  (Loaded inside the `vis` branch and the per-channel loops they were ~10
        // serial round trips per block — the kernel ran at 2.4 TB/s with the VALU 43 % busy.)
        const int v = v0 + wave;
        const int vl = v < V ? v : V - 1;
        const size_t o = (size_t)vl * G + (active ? i : 0);
        const float visf = p.vis[o * p.vis_stride];
        const uint32_t bits = DEGC >= 0 ? (uint32_t)p.clamp[o] : 0u;
        const ShPos pos = sh_position(p, vl, active ? i : 0);
        const float *gr = p.grec + o * (size_t)p.RF + 8;
        const bool pay8 = ntot <= 8;          // the whole payload half of a 64-byte record as two 16-byte loads
        float4 gq0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), gq1 = gq0;
        if (pay8) { gq0 = *(const float4 *)gr; gq1 = *(const float4 *)(gr + 4); }
        float *mg = s_gch + (wave * LSR_WAVE + lane) * cs;
        staged_barrier();
        const bool vis = wave < nv && active && visf > 0.0f;
        if (pay8) {   // channel gradients (clamped colour channels zeroed) go to this thread's LDS row first;
                      // the rolled channel loops below read them back by index
            const float gv[8] = {gq0.x, gq0.y, gq0.z, gq0.w, gq1.x, gq1.y, gq1.z, gq1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < ntot) mg[j] = (vis && !(DEGC >= 0 && j < 3 && (bits >> j & 1u))) ? gv[j] : 0.0f;
        }
        if (vis) {
            const ShDir dir = sh_direction(p, v, pos);
            float ddx = 0.0f, ddy = 0.0f, ddz = 0.0f;   // dL / d(unit direction)
            if (DEGC >= 0) {
                float sg[nbC > 0 ? nbC : 1];
#pragma unroll
                for (int k = 0; k < nbC; ++k) sg[k] = 0.0f;
#pragma unroll 1
                for (int c = 0; c < 3; ++c) {   // not unrolled: 25 coefficient reads in flight, not 75
                    float gc;
                    if (pay8) gc = mg[c];
                    else { gc = (bits >> c & 1u) ? 0.0f : gr[c]; mg[c] = gc; }
                    if (cmaj) {
                        const float *co = my + c * K;
#pragma unroll
                        for (int k = 0; k < nbC; ++k) sg[k] = __builtin_fmaf(co[k], gc, sg[k]);
                    } else {
                        const float *co = my + c;
#pragma unroll
                        for (int k = 0; k < nbC; ++k) sg[k] = __builtin_fmaf(co[3 * k], gc, sg[k]);
                    }
                }
                float dbas[nbC > 0 ? nbC : 1][3];
                sh_basis_grad<DEGC>(DEGC, cax ? dir.dz : dir.dx, cax ? dir.dx : dir.dy, cax ? dir.dy : dir.dz, dbas);
                float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
#pragma unroll
                for (int k = 0; k < nbC; ++k) { c0 += dbas[k][0] * sg[k]; c1 += dbas[k][1] * sg[k]; c2 += dbas[k][2] * sg[k]; }
                // gradient w.r.t. the basis arguments -> w.r.t. the direction (arguments (dz, dx, dy) when cax)
                ddx += cax ? c1 : c0; ddy += cax ? c2 : c1; ddz += cax ? c0 : c2;
            } else if (COFF == 3) {
                mg[0] = mg[1] = mg[2] = 0.0f;
            }
            if (hasF) {
                float sg[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) sg[k] = 0.0f;
#pragma unroll 1
                for (int c = 0; c < C; ++c) {
                    float gc;
                    if (pay8) gc = mg[COFF + c];
                    else { gc = gr[COFF + c]; mg[COFF + c] = gc; }
                    const float *co = myF + c * Kf;
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        if (k < nbF) sg[k] = __builtin_fmaf(co[k], gc, sg[k]);   // (unconditional clamped reads: 11 more spilled registers, 0.157 -> 0.164 ms)
                }
                float dbas[9][3];
                sh_basis_grad<2>(degF, dir.dz, dir.dx, dir.dy, dbas);
                float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    if (k < nbF) { e0 += dbas[k][0] * sg[k]; e1 += dbas[k][1] * sg[k]; e2 += dbas[k][2] * sg[k]; }
                ddx += e1; ddy += e2; ddz += e0;   // basis arguments were (dz, dx, dy)
            } else {
                for (int c = 0; c < C; ++c) mg[COFF + c] = 0.0f;
            }
            // through the normalisation and the scene scale to the mean
            const float dot = ddx * dir.dx + ddy * dir.dy + ddz * dir.dz;
            const float f = dir.sc / dir.len;
            const float gm[3] = {(ddx - dir.dx * dot) * f, (ddy - dir.dy * dot) * f, (ddz - dir.dz * dot) * f};
            if (d.vs_means != 0) {
                float *o3 = p.g.means3D + (size_t)v * d.vs_means + 3 * (size_t)i;
                o3[0] += gm[0]; o3[1] += gm[1]; o3[2] += gm[2];
            } else {
                acc[0] += gm[0]; acc[1] += gm[1]; acc[2] += gm[2];
            }
        } else {
            for (int j = 0; j < ntot; ++j) mg[j] = 0.0f;
        }
        __syncthreads();   // every thread is done reading the coefficient rows
        {   // the SH bases go where the coefficients were (recomputed here rather than kept live
            // through the direction pass: registers, not VALU, limit this kernel's occupancy)
            ShDir dir = {0.0f, 0.0f, 1.0f, 1.0f, 1.0f};
            if (vis) dir = sh_direction(p, v, pos);
            if (DEGC >= 0) {
                float basC[nbC > 0 ? nbC : 1];
                sh_basis<DEGC>(DEGC, cax ? dir.dz : dir.dx, cax ? dir.dx : dir.dy, cax ? dir.dy : dir.dz, basC);   // LSR_SH_AXES_REFERENCE: B(z,x,y)
                float *mb = s_basC + (wave * LSR_WAVE + lane) * kShBasisC;
#pragma unroll
                for (int k = 0; k < 26; ++k) mb[k] = (vis && k < nbC) ? basC[k < nbC ? k : 0] : 0.0f;
            }
        }
        __syncthreads();
        // ---- pass 2: thread = element of the contiguous coefficient-gradient runs ----
        const bool rmw = shared && v0 > 0;
        if (DEGC >= 0) {
            const int ks = p.ks[0];
            float *dst = p.g.color + (d.vs_color != 0 ? (size_t)v0 * d.vs_color : 0) + (size_t)g0 * ks;
            // Two instances of the loop: with the read-modify-write load in it the compiler puts
            // `s_waitcnt vmcnt(0)` in front of every store, and on gfx9 that also waits for the PREVIOUS
            // store — 28 serial store round trips per thread.  Views beyond nv and culled Gaussians have
            // zero rows in both LDS arrays, so all four view terms are added unconditionally.
            // Thread <-> ONE coefficient (k, c) of every per-th Gaussian (per = 256 / ks Gaussians per sweep; lane-consecutive
            // addresses as in a flat t += 256 loop): the element's index arithmetic — two divisions by runtime constants, the
            // band select, two LDS addresses: 24 vector instructions per element, seven of them quarter-rate v_mul_lo_u32,
            // around four multiply-adds in the flat loop's ISA — is paid once per thread.  The thread index goes through an
            // empty asm so that the compiler cannot hoist that arithmetic out of the view-chunk loop, where its results stay
            // live across pass 1 (48 more bytes of scratch per lane: measured 7 % slower than the flat loop; as written 1.5 %
            // faster — this kernel is bound by its chain of barrier-separated phases, not by its instruction count:
            // profiles/r05_ab_knobs.md section 10).  ks > 256 (more than 28 latent channels at degree 2) keeps the flat loop.
            auto colour_pass = [&](auto RMW) {
                if (ks <= kShThreads) {
                    int t0 = tid;
                    asm volatile("" : "+v"(t0));
                    const int per = udiv_small(kShThreads, p.mdiv[0]);
                    if (t0 >= per * ks) return;
                    const int gq = udiv_small(t0, p.mdiv[0]), rem = t0 - gq * ks;
                    int k, c;
                    if (cmaj) { c = udiv_small(rem, p.mdivK); k = rem - c * K; }
                    else { k = (rem * 0xAAABu) >> 17; c = rem - 3 * k; }
                    const int kb = k < nbC ? k : 25;   // coefficients beyond the evaluated bands: zero slot
                    const float *pb = s_basC + gq * kShBasisC + kb, *pg = s_gch + gq * cs + c;
                    float *o = dst + t0;
                    const int db = per * kShBasisC, dg = per * cs, dd = per * ks;
                    for (int g = gq; g < rows; g += per) {
                        float a = decltype(RMW)::value ? *o : 0.0f;
#pragma unroll
                        for (int vi = 0; vi < kShWaves; ++vi) a = __builtin_fmaf(pb[vi * LSR_WAVE * kShBasisC], pg[vi * LSR_WAVE * cs], a);
                        *o = a;
                        pb += db; pg += dg; o += dd;
                    }
                    return;
                }
                for (int t = tid; t < rows * ks; t += kShThreads) {
                    const int g = udiv_small(t, p.mdiv[0]), rem = t - g * ks;
                    int k, c;
                    if (cmaj) { c = udiv_small(rem, p.mdivK); k = rem - c * K; }
                    else { k = (rem * 0xAAABu) >> 17; c = rem - 3 * k; }
                    const int kb = k < nbC ? k : 25;
                    float a = decltype(RMW)::value ? dst[t] : 0.0f;
                    const float *pb = s_basC + g * kShBasisC + kb, *pg = s_gch + g * cs + c;
#pragma unroll
                    for (int vi = 0; vi < kShWaves; ++vi) a = __builtin_fmaf(pb[vi * LSR_WAVE * kShBasisC], pg[vi * LSR_WAVE * cs], a);
                    dst[t] = a;
                }
            };
            if (rmw) colour_pass(std::true_type{}); else colour_pass(std::false_type{});
        }
        if (hasF) {
            if (DEGC >= 0) __syncthreads();   // the colour element pass is done with the basis area
            {
                ShDir dir = {0.0f, 0.0f, 1.0f, 1.0f, 1.0f};
                if (vis) dir = sh_direction(p, v, pos);
                float basF[9];
                sh_basis<2>(degF, dir.dz, dir.dx, dir.dy, basF);
                float *mb = s_basF + (wave * LSR_WAVE + lane) * kShBasisF;
#pragma unroll
                for (int k = 0; k < 9; ++k) mb[k] = (vis && k < nbF) ? basF[k] : 0.0f;
                mb[9] = 0.0f;
            }
            __syncthreads();
            const int ks = p.ks[1];
            float *dst = p.g.features + (d.vs_feat != 0 ? (size_t)v0 * d.vs_feat : 0) + (size_t)g0 * ks;
            auto feature_pass = [&](auto RMW) {
                if (ks <= kShThreads) {   // one coefficient (c, k) of every per-th Gaussian per thread (see the colour pass)
                    int t0 = tid;
                    asm volatile("" : "+v"(t0));
                    const int per = udiv_small(kShThreads, p.mdiv[1]);
                    if (t0 >= per * ks) return;
                    const int gq = udiv_small(t0, p.mdiv[1]), rem = t0 - gq * ks;
                    const int c = udiv_small(rem, p.mdivKf), k = rem - c * Kf;
                    const int kb = k < nbF ? k : 9;
                    const float *pb = s_basF + gq * kShBasisF + kb, *pg = s_gch + gq * cs + COFF + c;
                    float *o = dst + t0;
                    const int db = per * kShBasisF, dg = per * cs, dd = per * ks;
                    for (int g = gq; g < rows; g += per) {
                        float a = decltype(RMW)::value ? *o : 0.0f;
#pragma unroll
                        for (int vi = 0; vi < kShWaves; ++vi) a = __builtin_fmaf(pb[vi * LSR_WAVE * kShBasisF], pg[vi * LSR_WAVE * cs], a);
                        *o = a;
                        pb += db; pg += dg; o += dd;
                    }
                    return;
                }
                for (int t = tid; t < rows * ks; t += kShThreads) {
                    const int g = udiv_small(t, p.mdiv[1]), rem = t - g * ks;
                    const int c = udiv_small(rem, p.mdivKf), k = rem - c * Kf;
                    const int kb = k < nbF ? k : 9;
                    float a = decltype(RMW)::value ? dst[t] : 0.0f;
                    const float *pb = s_basF + g * kShBasisF + kb, *pg = s_gch + g * cs + COFF + c;
#pragma unroll
                    for (int vi = 0; vi < kShWaves; ++vi) a = __builtin_fmaf(pb[vi * LSR_WAVE * kShBasisF], pg[vi * LSR_WAVE * cs], a);
                    dst[t] = a;
                }
            };
            if (rmw) feature_pass(std::true_type{}); else feature_pass(std::false_type{});
        }
    }
    if (d.vs_means == 0) {
        __syncthreads();   // the last element pass is done with the LDS arrays: the partial sums go to their start
        float (*s_part)[LSR_WAVE][3] = (float (*)[LSR_WAVE][3])s_lds;
#pragma unroll
        for (int a = 0; a < 3; ++a) s_part[wave][lane][a] = acc[a];
        __syncthreads();
        if (wave == 0 && active) {
            float *o3 = p.g.means3D + 3 * (size_t)i;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float t = 0.0f;
#pragma unroll
                for (int w = 0; w < kShWaves; ++w) t += s_part[w][lane][a];
                o3[a] += t;
            }
        }
    }
}

static bool group_enabled(const lsr_dims &d, int group) {
    return group == 0 ? d.color_mode == LSR_COLOR_SH : (d.feat_channels > 0 && d.feat_mode == LSR_FEAT_SH);
}

static uint32_t mdiv_of(int x) { return x > 0 ? (uint32_t)(((1u << 22) + x - 1) / x) : 0u; }

// d: the whole call's dims.  The kernels see ONE group's dims (group_dims: its views, shared inputs) and find the
// group's slices through ShParams::gs.
static ShParams make_params(const lsr_dims &d, const lsr_inputs &in, const GeomLayout &L, const char *geom) {
    ShParams p;
    p.d = group_dims(d); p.in = in;
    p.gs = group_strides(d);
    p.vis_stride = (int)(bin_stride(d) / 4);
    p.vis = (const float *)(geom + L.bin) + (narrow_bins(d) ? 1 : 2);
    p.rec = (float *)const_cast<char *>(geom + L.rec);
    p.clamp = (uint8_t *)const_cast<char *>(geom + L.sh_clamp);
    p.grec = nullptr; p.RF = L.rec_floats; p.g = lsr_in_grads{};
    p.has[0] = group_enabled(d, 0); p.has[1] = group_enabled(d, 1);
    p.ks[0] = p.has[0] ? d.sh_coeffs * 3 : 0;
    p.ks[1] = p.has[1] ? d.feat_sh_coeffs * d.feat_channels : 0;
    p.offF = (LSR_WAVE * p.ks[0] + 3) & ~3;
    for (int g = 0; g < 2; ++g) p.mdiv[g] = mdiv_of(p.ks[g]);
    p.mdivK = mdiv_of(d.sh_coeffs); p.mdivKf = mdiv_of(d.feat_sh_coeffs);
    return p;
}

// colour degree / feature offset -> template instance
#define LSR_SH_DISPATCH(KERNEL, degc, coff, ...)                                          \
    do {                                                                                   \
        if (degc == 4) hipLaunchKernelGGL((KERNEL<4, 3>), __VA_ARGS__);                    \
        else if (degc == 3) hipLaunchKernelGGL((KERNEL<3, 3>), __VA_ARGS__);               \
        else if (degc == 2) hipLaunchKernelGGL((KERNEL<2, 3>), __VA_ARGS__);               \
        else if (degc == 1) hipLaunchKernelGGL((KERNEL<1, 3>), __VA_ARGS__);               \
        else if (degc == 0) hipLaunchKernelGGL((KERNEL<0, 3>), __VA_ARGS__);               \
        else if (coff == 3) hipLaunchKernelGGL((KERNEL<-1, 3>), __VA_ARGS__);              \
        else hipLaunchKernelGGL((KERNEL<-1, 0>), __VA_ARGS__);                             \
    } while (0)

hipError_t launch_sh_forward(const lsr_dims &d, const lsr_inputs &in, char *geom, hipStream_t s) {
    if (d.num_gaussians == 0) return hipSuccess;
    if (!group_enabled(d, 0) && !group_enabled(d, 1)) return hipSuccess;
    const GeomLayout L = geom_layout(d);
    const ShParams p = make_params(d, in, L, geom);
    const int degc = p.has[0] ? d.sh_degree : -1, coff = d.color_mode != LSR_COLOR_NONE ? 3 : 0;
    const dim3 grid((d.num_gaussians + LSR_WAVE - 1) / LSR_WAVE, num_view_groups(d)), block(kShThreads);
    const size_t shm = ((size_t)p.offF + (size_t)LSR_WAVE * p.ks[1]) * 4;
    prof_begin(kStShFwd, s);
    LSR_SH_DISPATCH(k_sh_fwd, degc, coff, grid, block, shm, s, p);
    prof_end(kStShFwd, s);
    return hipGetLastError();
}

bool fused_preprocess_sh(const lsr_dims &d) {
    if (d.num_gaussians == 0 || !env_int("LSR_FUSE_SH", 1)) return false;
    const bool color_sh = d.color_mode == LSR_COLOR_SH, feat_sh = d.feat_channels > 0 && d.feat_mode == LSR_FEAT_SH;
    if (!color_sh && !feat_sh) return false;
    if (d.color_mode == LSR_COLOR_PRECOMP || (d.feat_channels > 0 && !feat_sh)) return false;   // a payload channel that is not a harmonic
    // the views of a workgroup read one input slice: a shared scene or view groups
    const bool shared = d.vs_means == 0 && d.vs_cov == 0 && d.vs_opac == 0 && (!color_sh || d.vs_color == 0) && (!feat_sh || d.vs_feat == 0);
    return shared || d.views_per_group > 1;
}

// words of the fused kernel's coefficient area (what make_params / launch_preprocess_sh lay out), from the dims alone
static size_t coef_words_of(const lsr_dims &d) {
    const int ks0 = group_enabled(d, 0) ? d.sh_coeffs * 3 : 0, ks1 = group_enabled(d, 1) ? d.feat_sh_coeffs * d.feat_channels : 0;
    const size_t offF = ((size_t)LSR_WAVE * ks0 + 3) & ~(size_t)3;
    return offF + (size_t)LSR_WAVE * ks1;
}
// single-pass binning in the fused kernel: coefficient area (at least 16 KB: it doubles as the bucket array) + the tile
// histogram + the first-slot array of the group's views within the 64 KB a launch gets without a function attribute
bool fused_segments_fit(const lsr_dims &d) {
    const size_t hist_off = std::max<size_t>((coef_words_of(d) + 3) & ~(size_t)3, 4096);
    const size_t VgT = (size_t)group_dims(d).num_views * (size_t)num_tiles(d);
    return (hist_off + 2 * VgT) * 4 <= 65536;
}

hipError_t launch_preprocess_sh(const lsr_dims &d, const lsr_inputs &in, char *geom, int32_t *radii, const FoldedScan &fs_in, bool seg_mode, hipStream_t s) {
    const GeomLayout L = geom_layout(d);
    {
        hipError_t e = launch_clear(geom + L.header, L.tile_start - L.header, s);   // header + tile_count + tile_cursor
        if (e != hipSuccess) return e;
    }
    const ShParams p = make_params(d, in, L, geom);
    const int degc = p.has[0] ? d.sh_degree : -1;
    const int T = (int)num_tiles(d), groups = num_view_groups(d), Vg = group_dims(d).num_views;
    PreShArgs a;
    a.binrec = geom + L.bin; a.narrow = narrow_bins(d) ? 1 : 0; a.radii = radii;
    a.tile_count = (uint32_t *)(geom + L.tile_count); a.header = (uint32_t *)(geom + L.header);
    a.fs = fs_in;
    a.fs.tile_start = (uint32_t *)(geom + L.tile_start); a.fs.tile_order = (uint32_t *)(geom + L.tile_order);
    a.views_total = d.num_views;
    // chunks per workgroup: one round of resident workgroups (three per CU, the number round 4 sized this for; four fit since
    // round 5) when the scene is large enough, at most 8
    const int64_t resident = (int64_t)device_cus() * 3 / groups;
    const int64_t nchunks = (d.num_gaussians + LSR_WAVE - 1) / LSR_WAVE;
    a.chunks = (int)std::max<int64_t>(1, std::min<int64_t>(8, (nchunks + std::max<int64_t>(resident, 1) - 1) / std::max<int64_t>(resident, 1)));
    const size_t coef_words = (size_t)p.offF + (size_t)LSR_WAVE * p.ks[1];
    a.hist_off = (int)((coef_words + 3) & ~(size_t)3);
    // the privatised tile histogram only while the whole dynamic allocation stays within the 64 KB a launch gets
    // without a function attribute (degree-4 colour + 13 latent channels at degree 2 take 50 KB of coefficient rows:
    // with 8192 counters behind them the launch would ask for 82 KB); larger calls count with global atomics
    a.lds_hist = ((size_t)Vg * T <= 8192 && ((size_t)a.hist_off + (size_t)Vg * T) * 4 <= 65536) ? 1 : 0;
    // single-pass binning (the caller asks for it only when segment_capacity(d) > 0: byte tile coordinates, T <= 1024): the
    // kernel then also emits the sort keys; its bucket array lives in the coefficient area, which therefore is at least 16 KB,
    // and the histogram is followed by the array of first slots — up to 8 chunks of 64 Gaussians per workgroup (two per wave)
    a.seg = SegOut{nullptr, 0u, index_packing(d).key_shift, 0u};
    if (seg_mode) {
        a.hist_off = std::max(a.hist_off, 4096);
        if (L.seg_cap == 0 || !narrow_bins(d) || ((size_t)a.hist_off + 2 * (size_t)Vg * T) * 4 > 65536) return hipErrorInvalidValue;
        a.lds_hist = 1;
        a.seg.keys = (uint64_t *)(geom + L.seg_keys); a.seg.cap = L.seg_cap;
    }
    size_t shm = ((size_t)a.hist_off + (a.lds_hist ? (size_t)Vg * T * (seg_mode ? 2 : 1) : 0)) * 4;
    if (a.fs.enabled) shm = std::max<size_t>(shm, (size_t)kFoldTiles * 4 + sizeof(TileScanShared<kShThreads>));   // the scan stages the counts at the start of the allocation, its counters behind them
    const dim3 grid((unsigned)((nchunks + a.chunks - 1) / a.chunks), groups), block(kShThreads);
    const bool fma = projection_contraction();
    prof_begin(kStPreprocess, s);
#define LSR_PSH2(DC, CO, FM) do { if (seg_mode) hipLaunchKernelGGL((k_preprocess_sh<DC, CO, FM, true>), grid, block, shm, s, p, a); \
                                  else hipLaunchKernelGGL((k_preprocess_sh<DC, CO, FM, false>), grid, block, shm, s, p, a); } while (0)
#define LSR_PSH(DC, CO) do { if (fma) LSR_PSH2(DC, CO, true); else LSR_PSH2(DC, CO, false); } while (0)
    if (degc == 4) LSR_PSH(4, 3); else if (degc == 3) LSR_PSH(3, 3); else if (degc == 2) LSR_PSH(2, 3);
    else if (degc == 1) LSR_PSH(1, 3); else if (degc == 0) LSR_PSH(0, 3); else LSR_PSH(-1, 0);
#undef LSR_PSH
#undef LSR_PSH2
    prof_end(kStPreprocess, s);
    return hipGetLastError();
}

template <int DEGC, int COFF>
static void allow_big_lds(size_t shm) {
    // function attributes are per device: set on every launch that needs it (a process may drive several GPUs)
    if (shm > 65536)
        (void)hipFuncSetAttribute((const void *)k_sh_bwd<DEGC, COFF>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
}

hipError_t launch_sh_backward(const lsr_dims &d, const lsr_inputs &in, const char *geom, const char *grad,
                              const lsr_in_grads &gin, hipStream_t s) {
    if (d.num_gaussians == 0) return hipSuccess;
    if (!group_enabled(d, 0) && !group_enabled(d, 1)) return hipSuccess;
    const GeomLayout L = geom_layout(d);
    const GradLayout R = grad_layout(d);
    ShParams p = make_params(d, in, L, geom);
    p.grec = (const float *)(grad + R.rec); p.g = gin;
    const int degc = p.has[0] ? d.sh_degree : -1, coff = d.color_mode != LSR_COLOR_NONE ? 3 : 0;
    const dim3 grid((d.num_gaussians + LSR_WAVE - 1) / LSR_WAVE, num_view_groups(d)), block(kShThreads);
    const int cs = (coff + d.feat_channels) | 1;
    const int area = std::max(p.offF + LSR_WAVE * p.ks[1], kShWaves * LSR_WAVE * kShBasisC);
    const size_t shm = ((size_t)area + (size_t)kShWaves * LSR_WAVE * cs) * 4;
    if (shm > 65536) {   // many direct channels next to an SH group
        if (degc == 4) allow_big_lds<4, 3>(shm); else if (degc == 3) allow_big_lds<3, 3>(shm);
        else if (degc == 2) allow_big_lds<2, 3>(shm); else if (degc == 1) allow_big_lds<1, 3>(shm);
        else if (degc == 0) allow_big_lds<0, 3>(shm); else if (coff == 3) allow_big_lds<-1, 3>(shm);
        else allow_big_lds<-1, 0>(shm);
    }
    prof_begin(kStShBwd, s);
    LSR_SH_DISPATCH(k_sh_bwd, degc, coff, grid, block, shm, s, p);
    prof_end(kStShBwd, s);
    return hipGetLastError();
}

}  // namespace lsr
