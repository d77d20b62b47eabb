// preprocess.hip — stage K1: per-(view, Gaussian) projection, EWA covariance, conic, radius,
// tile rectangle, colour from SH, plus the per-tile pair histogram (LDS-privatised).
// View-dependent payload channels (colour from SH, latent features from SH) are filled in
// afterwards by sh.hip for the Gaussians that survive culling.
//
// Build this file with -ffp-contract=off: radius / rectangle / depth bits feed the bit-exact
// tile lists, so every float operation must be the single IEEE operation written here (the CPU
// oracle performs the identical sequence).  Spec: SURVEY.md Appendix A.1-A.3.
#include <algorithm>

#include "lsr_blend.h"
#include "lsr_project.h"
#include "lsr_tile_scan.h"

namespace lsr {

typedef const float __attribute__((address_space(4))) *kfloat_ptr;   // constant address space: uniform loads become s_load

constexpr int kPreThreads = 256;
static_assert(kPreThreads == kPreThreadsScan, "the folded tile scan is sized for this workgroup");
constexpr int kPreItems = 8;  // (Gaussian, view) items per thread: 2048 per block, one histogram flush each

// VB = views per block.  Views that read the same input slice (a shared scene, or the views of one
// group) are processed by ONE block: a thread loads its Gaussian once and loops over the block's VB
// views, so a shared scene is read V / VB times instead of V times (269 of the kernel's
// 578 MB were per-view re-reads of the scene from L2/MALL).  Items per thread shrink by the same
// factor, so the grid keeps its size.
//
// SEG (round 5, single-pass binning: lsr_internal.h segment_capacity): the workgroup also EMITS the sort keys of its
// items.  When all its items are projected, the LDS histogram holds its pair count per (view, tile); the one global
// atomic per non-empty counter that used to add the count to the tile's total now RETURNS the old total — the
// workgroup's first slot in the tile's fixed-capacity key segment — and a second pass over the workgroup's own binning
// records (12 bytes each, written a few microseconds earlier by the same thread: L2 hits, and a thread only ever reads
// what it wrote itself) places `depth << 32 | index << 8 | sub-block code` at segment[base + LDS cursor++].  This is
// k_scatter's body without its launch, without the machine-wide re-read of the records from HBM and without the wait
// for the tile scan in front of it; the order of the keys inside a segment is whatever the atomics produce, as before
// (k_sort_tiles sorts distinct 64-bit keys).
template <int COLOR_MODE, bool LDS_HIST, int VB, bool FMA, bool SEG>
__global__ void __launch_bounds__(kPreThreads, 6)   // six waves per SIMD (80 VGPRs): the kernel streams, occupancy hides its latencies
k_preprocess(lsr_dims d, lsr_inputs in, float *__restrict__ rec, int RF, char *__restrict__ binrec, int narrow,
             int32_t *__restrict__ radii, uint32_t *__restrict__ tile_count, uint32_t *header, FoldedScan fs, int kItems, SegOut seg) {
    static_assert(!SEG || LDS_HIST, "the key emission reserves its slots from the LDS histogram");
    extern __shared__ uint32_t s_hist[];   // [VB][T] pair counts
    // 64-byte records are staged here and stored by the whole block as one contiguous run
    // (lane-contiguous 16-byte stores) instead of 4 strided partial-line stores per thread.
    // Layout [k][thread] with a row stride of kPreThreads + 4 float4: the per-thread stores (fixed k,
    // consecutive threads) and the run-order loads (4 consecutive lanes = the 4 words of one
    // record) are both bank-conflict free.
    constexpr int kRecRow = kPreThreads + 4;
    __shared__ float4 s_rec[kRecRow * 4];
    static_assert(sizeof(float4) * kRecRow * 4 >= sizeof(uint32_t) * kFoldTiles, "the folded tile scan stages its counts here");
    const bool staged = RF == 16;
    const int v0 = blockIdx.y * VB;
    const int G = d.num_gaussians;
    const int gx = (d.width + LSR_TILE - 1) / LSR_TILE, gy = (d.height + LSR_TILE - 1) / LSR_TILE;
    const int T = gx * gy;
    // per-view constants that cost an IEEE division each: once per block, not once per (Gaussian, view)
    __shared__ float2 s_focal[VB];
    if (threadIdx.x < VB && v0 + (int)threadIdx.x < d.num_views) {
        const float *vw = in.views + (size_t)(v0 + threadIdx.x) * LSR_VIEW_FLOATS;
        s_focal[threadIdx.x] = make_float2(d.width / (2.0f * vw[35]), d.height / (2.0f * vw[36]));
    }
    if (LDS_HIST)
        for (int t = threadIdx.x; t < VB * T; t += kPreThreads) s_hist[t] = 0;
    __syncthreads();
    const int ce = d.cov_elems;
    const size_t sl = (size_t)input_slice(d, v0);   // the input slice all views of this block read
    const float *means = in.means3D + sl * d.vs_means;
    const float *covs = in.cov3D + sl * d.vs_cov;
    const float *opac = in.opacities + sl * d.vs_opac;
    constexpr int coff = COLOR_MODE == LSR_COLOR_NONE ? 0 : 3;
    const bool direct_feat = d.feat_mode == LSR_FEAT_DIRECT;
    const bool skip_none = reached_only(d);

    // kItems Gaussians per thread (at most kPreItems / VB: 2048 (Gaussian, view) items per block and histogram flush;
    // fewer when the call is small, so that a single view still fills the machine: launch_preprocess)
    const int base = blockIdx.x * (kPreThreads * kItems);
#pragma unroll 1
    for (int it = 0; it < kItems; ++it) {
        const int chunk0 = base + it * kPreThreads;
        if (chunk0 >= G) break;   // block-uniform
        const int i = chunk0 + threadIdx.x;
        const bool in_range = i < G;
        const size_t ii = in_range ? (size_t)i : 0;
        // ---- the Gaussian, once for all views of the block ----
        const float q0 = means[3 * ii], q1 = means[3 * ii + 1], q2 = means[3 * ii + 2];
        const float *c6 = covs + (size_t)ce * ii;
        const float r0 = c6[0], r1 = c6[1], r2 = c6[2];
        const float r3 = c6[ce == 9 ? 4 : 3], r4 = c6[ce == 9 ? 5 : 4], r5 = c6[ce == 9 ? 8 : 5];
        const float opacity = opac[ii];
        float pay_in[3] = {0.0f, 0.0f, 0.0f};
        if (COLOR_MODE == LSR_COLOR_PRECOMP) {
            const float *cp = in.color + sl * d.vs_color + 3 * ii;
            pay_in[0] = cp[0]; pay_in[1] = cp[1]; pay_in[2] = cp[2];
        }
        const float *fp = in.features + sl * d.vs_feat + ii * d.feat_channels;
        // payload slots 8.. : rgb (if any; SH colour is filled in by sh.hip) then the feature channels,
        // zero padded.  View independent, so the two words of a 64-byte record are built once per Gaussian.
        auto payload = [&](int c4) {
            float w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * c4 + k;
                w[k] = c < coff ? pay_in[c < 3 ? c : 0]
                                : ((direct_feat && c - coff < d.feat_channels) ? fp[c - coff] : 0.0f);
            }
            return make_float4(w[0], w[1], w[2], w[3]);
        };
        float4 pay0 = make_float4(0, 0, 0, 0), pay1 = make_float4(0, 0, 0, 0);
        if (staged) { pay0 = payload(0); pay1 = payload(1); }
        // The wait for these loads belongs HERE, once per Gaussian: left to the compiler it lands inside the
        // view loop as `s_waitcnt vmcnt(0)`, which on gfx9 also waits for the previous view's record stores.
        asm volatile("" :: "v"(q0), "v"(q1), "v"(q2), "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(opacity),
                     "v"(pay0.x), "v"(pay0.y), "v"(pay0.z), "v"(pay0.w), "v"(pay1.x), "v"(pay1.y), "v"(pay1.z), "v"(pay1.w));

#pragma unroll 1
        for (int vb = 0; vb < VB; ++vb) {
            const int v = v0 + vb;
            if (v >= d.num_views) break;   // block-uniform
            // the camera goes through the scalar cache into SGPRs (constant address space: the table was
            // written by an earlier launch), not through 44 broadcast vector loads per lane
            const kfloat_ptr vw = (kfloat_ptr)(in.views + (size_t)v * LSR_VIEW_FLOATS);
            float vm[16], pm[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { vm[k] = vw[k]; pm[k] = vw[16 + k]; }
            const float tanfovx = vw[35], tanfovy = vw[36];
            const float focal_x = s_focal[vb].x, focal_y = s_focal[vb].y;   // = width / (2 tan), height / (2 tan)
            const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
            const float scale = vw[40], scale2 = scale * scale;   // scene scale (1/near), applied like the reference does
            uint32_t *tc = tile_count + (size_t)v * T;
            uint32_t *hist = s_hist + vb * T;
            const size_t o = (size_t)v * G + ii;
            const float p0 = q0 * scale, p1 = q1 * scale, p2 = q2 * scale;
            const float s0 = r0 * scale2, s1 = r1 * scale2, s2 = r2 * scale2;
            const float s3 = r3 * scale2, s4 = r4 * scale2, s5 = r5 * scale2;
            const Projected pj = project_gaussian<FMA>(vm, pm, limx, limy, focal_x, focal_y, p0, p1, p2, s0, s1, s2, s3, s4, s5,
                                                       d.width, d.height, gx, gy, in_range);
            const bool ok = pj.ok;
            const float px = pj.px, py = pj.py, conic_a = pj.conic_a, conic_b = pj.conic_b, conic_c = pj.conic_c, tz = pj.tz;
            const float my_radius = pj.radius;
            const int rminx = pj.rminx, rminy = pj.rminy, rmaxx = pj.rmaxx, rmaxy = pj.rmaxy;

            const float4 rr0 = make_float4(px, py, conic_a, conic_b);
            const float4 rr1 = make_float4(conic_c, opacity, ok ? tz : 0.0f, 0.0f);   // view z 0 marks a culled record
            // footprint span for the half-tile render lists (k_scatter / k_sort_tiles); not part of the bit-exact contract
            const uint32_t span = ok ? footprint_cells(px, py, conic_a, conic_b, conic_c, opacity, rminx, rminy) : kSpanNone;
            if (ok) {
                if (!staged) {
                    float4 *R = (float4 *)(rec + o * (size_t)RF);
                    R[0] = rr0; R[1] = rr1;
                    for (int c4 = 0; c4 < (RF - 8) / 4; ++c4) R[2 + c4] = payload(c4);
                }
                // per-tile pair counts (also the compositing kernels' scheduling key: a finer work
                // estimate — quadrants reached per entry — was measured to schedule no better)
                // (LSR_FWD_REACHED_ONLY: only the tiles of the rectangle the footprint box reaches are pairs)
                int hx0 = rminx, hy0 = rminy, hx1 = rmaxx, hy1 = rmaxy;
                if (skip_none) reached_rect(span, hx0, hy0, hx1, hy1);
                for (int y = hy0; y < hy1; ++y)
                    for (int x = hx0; x < hx1; ++x) {
                        if (LDS_HIST) atomicAdd(&hist[y * gx + x], 1u);
                        else atomicAdd(&tc[y * gx + x], 1u);
                    }
            }
            if (in_range) {
                radii[o] = ok ? (int32_t)my_radius : 0;
                const float out_depth = ok ? tz : 0.0f;
                if (narrow) {
                    BinRec br;
                    br.rect = ok ? ((uint32_t)rminx | ((uint32_t)rminy << 8) | ((uint32_t)rmaxx << 16) | ((uint32_t)rmaxy << 24)) : 0u;
                    br.depth = out_depth; br.span = span;
                    ((BinRec *)binrec)[o] = br;
                } else {
                    BinRecWide br;
                    br.rect = ok ? make_ushort4((unsigned short)rminx, (unsigned short)rminy, (unsigned short)rmaxx, (unsigned short)rmaxy)
                                 : make_ushort4(0, 0, 0, 0);
                    br.depth = out_depth; br.span = span;
                    ((BinRecWide *)binrec)[o] = br;
                }
            }
            if (staged) {
                s_rec[0 * kRecRow + threadIdx.x] = rr0;
                s_rec[1 * kRecRow + threadIdx.x] = rr1;
                s_rec[2 * kRecRow + threadIdx.x] = pay0;
                s_rec[3 * kRecRow + threadIdx.x] = pay1;
                __syncthreads();
                float4 *dst = (float4 *)(rec + ((size_t)v * G + chunk0) * 16);
                const int nrec = G - chunk0 < kPreThreads ? G - chunk0 : kPreThreads;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = k * kPreThreads + threadIdx.x;      // float4 index inside the run
                    // culled Gaussians (view z == 0 in their staged record) are never read: skip them
                    if (idx < nrec * 4 && s_rec[kRecRow + (idx >> 2)].z > 0.0f) dst[idx] = s_rec[(idx & 3) * kRecRow + (idx >> 2)];
                }
                __syncthreads();
            }
        }
    }
    if (LDS_HIST && !SEG) {
        __syncthreads();
        for (int t = threadIdx.x; t < VB * T; t += kPreThreads) {
            const int v = v0 + t / T;
            const uint32_t c = s_hist[t];
            if (c && v < d.num_views) atomicAdd(&tile_count[(size_t)v * T + (t % T)], c);
        }
    }
    if (SEG) {
        // ---- reserve: count -> first slot of this workgroup in the tile's segment ----
        __syncthreads();
        uint32_t *s_first = s_hist + VB * T;      // (the dynamic allocation holds two arrays of VB * T words in this instance)
        // (four counters per thread and round, their returning atomics in flight together: one round trip to the
        // memory-side atomic unit per round instead of one per counter.  Only non-empty counters issue an atomic — adds of
        // zero to a clamped address put thousands of same-address atomics of EVERY workgroup on one word: measured 5.5 ms
        // instead of 0.15 for a one-view-per-workgroup launch)
        for (int t0 = threadIdx.x; t0 < VB * T; t0 += 4 * kPreThreads) {
            uint32_t c[4], first[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = t0 + k * kPreThreads;
                c[k] = t < VB * T ? s_hist[t] : 0u;
                if (c[k] && v0 + t / T >= d.num_views) c[k] = 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = t0 + k * kPreThreads;
                if (c[k]) first[k] = __hip_atomic_fetch_add(&tile_count[(size_t)(v0 + t / T) * T + (t % T)], c[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = t0 + k * kPreThreads;
                if (t < VB * T) s_first[t] = first[k];
            }
        }
        __syncthreads();
        const uint32_t cap = seg.cap;
        // ---- emit, one view of the workgroup at a time, THROUGH LDS: the keys are first bucketed by tile in the (now idle)
        // record staging array — a key's slot is its tile's local offset (exclusive scan of the workgroup's counts) plus its
        // arrival rank — together with their final positions, and then leave as one linear pass over the slots: every
        // lane stores, and the lanes of a tile's run hit consecutive addresses.  Written straight from the item loop (the
        // first version of this pass) the same keys were 7.4 M lane-scattered 8-byte stores at 31 % lane utilisation: 0.053
        // of the kernel's 0.155 ms (ablations in profiles/r05_ab_knobs.md); a thread still reads back only the binning
        // records it wrote itself. ----
        constexpr int KI = kPreItems / VB;                               // most Gaussians per thread
        uint32_t *s_delta = (uint32_t *)s_rec;                           // [T] first slot in the segment minus local offset
        const int kbase = (T * 4 + 7) & ~7;
        uint64_t *s_key = (uint64_t *)((char *)s_rec + kbase);
        const uint32_t buf = (uint32_t)((sizeof(s_rec) - kbase) / 12);
        uint32_t *s_pos = (uint32_t *)((char *)s_rec + kbase + (size_t)buf * 8);
        __shared__ uint32_t s_scanw[kPreThreads / LSR_WAVE];
        const int tpt = (T + kPreThreads - 1) / kPreThreads;             // tiles per thread of the scan (<= 4: T <= 1024)
#pragma unroll 1
        for (int vb = 0; vb < VB; ++vb) {
            const int v = v0 + vb;
            if (v >= d.num_views || (seg.ablate & 2u)) break;             // block-uniform
            uint32_t *cur = s_hist + vb * T;
            // local offsets of this view's tiles: a contiguous chunk of tiles per thread
            uint32_t cnt[4], mine = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = (int)threadIdx.x * tpt + k;
                cnt[k] = (k < tpt && t < T) ? cur[t] : 0u;
                mine += cnt[k];
            }
            uint32_t n_v;
            uint32_t off = block_exclusive_scan<kPreThreads>(mine, s_scanw, n_v);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = (int)threadIdx.x * tpt + k;
                if (k < tpt && t < T) { cur[t] = off; s_delta[t] = s_first[vb * T + t] - off; off += cnt[k]; }
            }
            __syncthreads();
            const uint32_t seg0 = (uint32_t)v * (uint32_t)T;
            // this view's records of the thread's Gaussians: unconditional loads at clamped addresses, all in flight together
            uint3 br[KI];
#pragma unroll
            for (int it = 0; it < KI; ++it) {
                const int i = min(base + min(it, kItems - 1) * kPreThreads + (int)threadIdx.x, G - 1);
                br[it] = *(const uint3 *)(binrec + ((size_t)v * G + (size_t)i) * sizeof(BinRec));
            }
#pragma unroll
            for (int it = 0; it < KI; ++it) {
                const int i = base + it * kPreThreads + (int)threadIdx.x;
                const uint32_t rc = (it < kItems && i < G) ? br[it].x : 0u;     // (a culled record holds an empty rectangle)
                const int x0 = rc & 0xff, y0 = (rc >> 8) & 0xff, x1 = (rc >> 16) & 0xff, y1 = rc >> 24;
                const uint64_t key = ((uint64_t)br[it].y << 32) | ((uint32_t)i << seg.key_shift);
                const uint32_t sp = br[it].z;
                int ex0 = x0, ey0 = y0, ex1 = x1, ey1 = y1;
                if (skip_none) reached_rect(sp, ex0, ey0, ex1, ey1);      // (the pairs pass 1 counted)
                for (int y = ey0; y < ey1; ++y)
                    for (int x = ex0; x < ex1; ++x) {
                        const int t = y * gx + x;
                        const uint32_t code = seg.key_shift ? span_code(sp, x - x0, y - y0) : 0u;
                        const uint32_t slot = atomicAdd(&cur[t], 1u);
                        // position in the tile's segment, CLAMPED (not tested): the surplus keys of an overfull segment land
                        // on its last slot; such a tile is binned again by the fallback scatter
                        const uint32_t pos = (seg0 + (uint32_t)t) * cap + min(slot + s_delta[t], cap - 1u);
                        if (slot < buf) { s_key[slot] = key | code; s_pos[slot] = pos; }
                        else seg.keys[pos] = key | code;              // (more pairs in one view of this workgroup than the array holds)
                    }
            }
            __syncthreads();
            const uint32_t nflush = min(n_v, buf);
            if (!(seg.ablate & 1u))
                for (uint32_t j = threadIdx.x; j < nflush; j += kPreThreads) seg.keys[s_pos[j]] = s_key[j];
            __syncthreads();
        }
    }
    // ---- the tile scan, folded in (round 4; it used to be a kernel of its own between this one and k_scatter): the LAST
    // workgroup to arrive scans the tile counts, writes the tile offsets, the header, the compositing work items and
    // the two numbers the synchronous forward's host is waiting for (lsr_tile_scan.h).
    // The counts are only ever touched by agent-scope atomics, which are performed past the (mutually incoherent)
    // per-XCD L2s: a block waits until its own count updates have been acknowledged (vmcnt) and only then arrives at the
    // counter, and the last block reads the counts with agent-scope atomic loads.  No release fence: that would write
    // back every record line the block has just left dirty in L2 (measured in round 2: the forward went from 0.56 to
    // 0.83 ms per step). ----
    if (fs.enabled) {
        __shared__ uint32_t s_last;
        __shared__ TileScanShared<kPreThreads> s_scan;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t arrived = __hip_atomic_fetch_add(&header[kHdrPreDone], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = arrived == gridDim.x * gridDim.y - 1u;
        }
        __syncthreads();
        if (s_last) {
            // the counts go through LDS (the record staging array is free by now): sixteen of them per thread in
            // registers cost the whole kernel a wave per SIMD (85 instead of 77 VGPRs)
            uint32_t *s_counts = (uint32_t *)s_rec;
            const int N = d.num_views * T;
            for (int i0 = threadIdx.x; i0 < N; i0 += 4 * kPreThreads) {     // four coalesced loads in flight per thread
                uint32_t c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    c[k] = __hip_atomic_load(&tile_count[min(i0 + k * kPreThreads, N - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + k * kPreThreads < N) s_counts[i0 + k * kPreThreads] = c[k];
            }
            __syncthreads();
            tile_scan_block<kPreThreads, 0, false>(s_counts, fs.tile_start, header, HostMirror{fs.host_words, fs.host_seq},
                                                   fs.tile_order, N, fs.capacity, s_scan);
        }
    }
}

__global__ void __launch_bounds__(256) k_clear16(uint4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
// Zero `bytes` (a multiple of 16) at a 16-byte aligned device address.
hipError_t launch_clear(void *ptr, size_t bytes, hipStream_t s) {
    const size_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    const size_t blocks = (n16 + 255) / 256;
    hipLaunchKernelGGL(k_clear16, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, (uint4 *)ptr, n16);
    return hipGetLastError();
}

hipError_t launch_preprocess(const lsr_dims &d, const lsr_inputs &in, char *geom, int32_t *radii, const FoldedScan &fs_in,
                             bool seg_mode, hipStream_t s) {
    const GeomLayout L = geom_layout(d);
    const int T = (int)num_tiles(d);
    // zero header + tile_count + tile_cursor (adjacent).  A kernel rather than hipMemsetAsync: the
    // stage sequence is captured into hipGraphs (latency mode), where back-to-back replays of a
    // captured memset node were observed to leave the counters uncleared on ROCm 7.2.
    {
        hipError_t e = launch_clear(geom + L.header, L.tile_start - L.header, s);   // both offsets are 256-byte aligned
        if (e != hipSuccess) return e;
    }
    if (d.num_gaussians == 0) return hipSuccess;
    // views per block: 4 (or 2) when that many consecutive views read the same input slice
    const bool shared = d.vs_means == 0 && d.vs_cov == 0 && d.vs_opac == 0 && (d.color_mode != LSR_COLOR_PRECOMP || d.vs_color == 0) &&
                        (d.feat_channels == 0 || d.vs_feat == 0);
    const int span = shared ? d.num_views : (d.views_per_group > 1 ? d.views_per_group : 1);   // views per input slice
    int vb = (span % 4 == 0 || (shared && span >= 4)) ? 4 : ((span % 2 == 0 || (shared && span >= 2)) ? 2 : 1);
    if (const int forced_vb = env_int("LSR_PRE_VB", 0)) vb = std::min(vb, forced_vb >= 4 ? 4 : (forced_vb >= 2 ? 2 : 1));   // (development knob)
    // Gaussians per thread: kPreItems / vb (one LDS histogram flush per 2048 items) unless that leaves fewer than two
    // workgroups per CU — a single 300 k view was 147 workgroups of 8 sequential Gaussians per thread on 256 CUs
    const int64_t yblocks = (d.num_views + vb - 1) / vb;
    const int64_t fill = ((int64_t)d.num_gaussians * yblocks + (int64_t)kPreThreads * 2 * device_cus() - 1) / ((int64_t)kPreThreads * 2 * device_cus());
    // (development knob: (Gaussian, view) items per thread and histogram flush; the key emission of the single-pass binning
    // keeps a thread's binning records in registers: never more than kPreItems there)
    const int pre_items = seg_mode ? std::min(kPreItems, std::max(1, env_int("LSR_PRE_ITEMS", kPreItems))) : std::max(1, env_int("LSR_PRE_ITEMS", kPreItems));
    const int items = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(1, pre_items / vb), fill));
    dim3 grid((d.num_gaussians + kPreThreads * items - 1) / (kPreThreads * items), (unsigned)yblocks);
    float *rec = (float *)(geom + L.rec);
    char *binrec = geom + L.bin;
    const int narrow = narrow_bins(d) ? 1 : 0;
    const int RF = L.rec_floats;
    uint32_t *tc = (uint32_t *)(geom + L.tile_count);
    FoldedScan fs = fs_in;
    fs.tile_start = (uint32_t *)(geom + L.tile_start); fs.tile_order = (uint32_t *)(geom + L.tile_order);
    const bool lds = (size_t)T * vb <= 4096;
    const size_t shm = lds ? (size_t)T * vb * 4 * (seg_mode ? 2 : 1) : 0;   // (single-pass binning: counts and first slots)
    const bool fma = projection_contraction();
    // single-pass binning: the caller asks for it only when segment_capacity(d) > 0, which implies byte tile coordinates
    // and T <= 1024 (the LDS histogram of up to 4 views)
    SegOut so{nullptr, 0u, index_packing(d).key_shift, (uint32_t)env_int("LSR_SEG_ABLATE", 0)};
    if (seg_mode) {
        if (!lds || !narrow || L.seg_cap == 0) return hipErrorInvalidValue;
        so.keys = (uint64_t *)(geom + L.seg_keys); so.cap = L.seg_cap;
    }
#define LSR_PRE4(CM, LH, VBV, FM, SG) hipLaunchKernelGGL((k_preprocess<CM, LH, VBV, FM, SG>), grid, dim3(kPreThreads), shm, s, d, in, rec, RF, binrec, narrow, radii, tc, (uint32_t *)(geom + L.header), fs, items, so)
#define LSR_PRE3(CM, LH, VBV, FM) do { if (LH && seg_mode) LSR_PRE4(CM, LH, VBV, FM, LH); else LSR_PRE4(CM, LH, VBV, FM, false); } while (0)
#define LSR_PRE2(CM, LH, VBV) do { if (fma) LSR_PRE3(CM, LH, VBV, true); else LSR_PRE3(CM, LH, VBV, false); } while (0)
#define LSR_PRE(CM)                                                                              \
    do {                                                                                         \
        if (lds) { if (vb == 4) LSR_PRE2(CM, true, 4); else if (vb == 2) LSR_PRE2(CM, true, 2); else LSR_PRE2(CM, true, 1); } \
        else { if (vb == 4) LSR_PRE2(CM, false, 4); else if (vb == 2) LSR_PRE2(CM, false, 2); else LSR_PRE2(CM, false, 1); } \
    } while (0)
    prof_begin(kStPreprocess, s);
    if (d.color_mode == LSR_COLOR_SH) LSR_PRE(LSR_COLOR_SH);
    else if (d.color_mode == LSR_COLOR_PRECOMP) LSR_PRE(LSR_COLOR_PRECOMP);
    else LSR_PRE(LSR_COLOR_NONE);
#undef LSR_PRE
#undef LSR_PRE2
#undef LSR_PRE3
#undef LSR_PRE4
    prof_end(kStPreprocess, s);
    return hipGetLastError();
}

}  // namespace lsr
