// binning.hip — stages K2-K5 re-designed for CDNA4: instead of one global 64-bit radix sort over
// all (tile, depth) keys, pairs are binned straight into per-tile segments and every tile list
// is depth-sorted inside LDS by its own workgroup.
//
//   k_tile_scan : exclusive scan of the per-tile pair counts (V*T entries) -> tile_start,
//                 total pair count and the longest list (header[0], header[1]); also emits the
//                 compositing work items — two (view, tile, half) items per tile — ordered
//                 longest-list-first (counting sort): the work queue order of both compositing
//                 kernels (longest-processing-time-first balancing).  Since round 4 this work is done by the
//                 last workgroup of k_preprocess (lsr_tile_scan.h); the kernel here remains for calls whose
//                 V*T counts do not fit that workgroup's registers.
//   k_scatter   : every visible Gaussian writes (depth_bits<<32 | index) into each tile segment
//                 it overlaps.  Slots come from a two-level reservation: LDS counters per block,
//                 then ONE global atomic per (block, tile).
//   k_sort_tiles: per-tile sort of the 64-bit keys in LDS (bucket sort with parallel in-bucket ranking, bitonic
//                 fallback).  Ascending (depth bits, index)
//                 == the published stable sort by depth with ties in emission (= index) order,
//                 so lists are bit-exact whatever order the scatter produced.
#include "lsr_blend.h"
#include "lsr_tile_scan.h"
#include <algorithm>
#include <mutex>
#ifdef LSR_ENABLE_TRACE
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

namespace lsr {

// ------------------------------------------------------------------------------------------
constexpr int kScanThreads = 1024;

constexpr int kScanRegs = 8;   // counts per thread kept in registers (4 at 1024 threads x 4096 tiles)

__global__ void __launch_bounds__(kScanThreads)
k_tile_scan(const uint32_t *__restrict__ count, uint32_t *__restrict__ start, uint32_t *header,
            uint32_t *host_words, uint32_t host_seq, uint32_t *__restrict__ order, int N, uint32_t capacity) {
    __shared__ TileScanShared<kScanThreads> s_scan;
    tile_scan_block<kScanThreads, kScanRegs, false>(count, start, header, HostMirror{host_words, host_seq}, order, N, capacity, s_scan);
}

hipError_t launch_tile_scan(const lsr_dims &d, char *geom, uint32_t *host_words, uint32_t host_seq, uint32_t pair_capacity, hipStream_t s) {
    const GeomLayout L = geom_layout(d);
    const int N = d.num_views * (int)num_tiles(d);
    prof_begin(kStTileScan, s);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(kScanThreads), 0, s,
                       (const uint32_t *)(geom + L.tile_count), (uint32_t *)(geom + L.tile_start),
                       (uint32_t *)(geom + L.header), host_words, host_seq, (uint32_t *)(geom + L.tile_order), N, pair_capacity);
    prof_end(kStTileScan, s);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
constexpr int kScatThreads = 256;
constexpr int kScatItems = 12;   // most (view, Gaussian) items of one thread; the launcher picks the count

// over_cap != 0 (single-pass binning, round 5): the projection kernel has already written the keys of every tile whose
// list fits its fixed-capacity segment; this launch only places the pairs of the tiles that do NOT fit
// (tile_count > over_cap) into their exact segments of the binning workspace — the device-side fallback for overfull
// tiles.  The synchronous forward launches it when the host has seen such a list, the no-sync forward always (every
// workgroup leaves at once when the header's longest list fits).
template <bool LDS_RESERVE, bool NARROW>
__global__ void __launch_bounds__(kScatThreads, NARROW ? 8 : 6)   // (the wide-coordinate instance, images beyond 4080 px a side, keeps two words per rectangle: 80 registers instead of a spill)
k_scatter(int G, int gx, int T, const char *__restrict__ binrec,
          const uint32_t *__restrict__ tile_start, uint32_t *__restrict__ tile_cursor,
          uint64_t *__restrict__ keys, uint32_t capacity, uint32_t chunks, int items, uint32_t key_shift, unsigned long long *trace,
          const uint32_t *__restrict__ tile_count, const uint32_t *__restrict__ header, uint32_t over_cap, int skip_none) {
    // (skip_none: LSR_FWD_REACHED_ONLY — pairs whose footprint box misses the tile were not counted by the projection kernel)
    if (over_cap && header[kHdrMaxTile] <= over_cap) return;   // (uniform) no overfull tile in this call
#ifdef LSR_ENABLE_TRACE
#define LSR_STAMP(k) do { if (trace && threadIdx.x == 0) trace[8 * (size_t)blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define LSR_STAMP(k) do { } while (0)
#endif
    LSR_STAMP(0);
    extern __shared__ uint32_t s_mem[];  // [T] counts, [T] bases
    uint32_t *s_cnt = s_mem, *s_base = s_mem + T;
    const uint32_t unit = blockIdx.x;   // view-major (view, chunk)
    const int v = (int)(unit / chunks);
    const size_t vo = (size_t)v * G;
    const uint32_t *ts = tile_start + (size_t)v * T;
    uint32_t *cur = tile_cursor + (size_t)v * T;
    const uint32_t *tcv = tile_count + (size_t)v * T;
    const int base = (int)(unit % chunks) * (kScatThreads * items);
    // rectangles stay PACKED in registers between the two passes (narrow records: one word, wide: x0 | y0 << 16,
    // x1 | y1 << 16); the empty asm makes each pass unpack its own copy instead of keeping four coordinates
    // per item live
    uint32_t lo16[kScatItems], hi16[NARROW ? 1 : kScatItems], span[kScatItems];
    float dep[kScatItems];
    // unconditional loads at clamped addresses (all in flight together; a load under a per-lane condition
    // makes the compiler wait for each one), invalid items become empty rectangles afterwards
#pragma unroll
    for (int it = 0; it < kScatItems; ++it) {
        const int i = base + min(it, items - 1) * kScatThreads + (int)threadIdx.x;
        const bool valid = it < items && i < G;
        if (NARROW) {
            const uint3 br = *(const uint3 *)(binrec + (vo + min(i, G - 1)) * sizeof(BinRec));      // rectangle, depth bits, footprint span
            lo16[it] = valid ? br.x : 0u; dep[it] = __uint_as_float(br.y); span[it] = br.z;
        } else {
            const uint4 br = *(const uint4 *)(binrec + (vo + min(i, G - 1)) * sizeof(BinRecWide));
            lo16[it] = br.x; hi16[NARROW ? 0 : it] = valid ? br.y : 0u; dep[it] = __uint_as_float(br.z); span[it] = br.w;
        }
    }
    auto unpack = [&](int it, int &x0, int &y0, int &x1, int &y1) {
        uint32_t a = lo16[it], b = NARROW ? 0u : hi16[NARROW ? 0 : it];
        asm volatile("" : "+v"(a), "+v"(b));
        if (NARROW) { x0 = a & 0xff; y0 = (a >> 8) & 0xff; x1 = (a >> 16) & 0xff; y1 = a >> 24; }
        else { x0 = a & 0xffff; y0 = a >> 16; x1 = b & 0xffff; y1 = b >> 16; }
    };
    if (LDS_RESERVE) {
        for (int t = threadIdx.x; t < T; t += kScatThreads) s_cnt[t] = 0;
        __syncthreads();
        LSR_STAMP(1);
#pragma unroll
        for (int it = 0; it < kScatItems; ++it) {
            int x0, y0, x1, y1;
            unpack(it, x0, y0, x1, y1);
            if (skip_none) reached_rect(span[it], x0, y0, x1, y1);
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    if (over_cap && tcv[y * gx + x] <= over_cap) continue;
                    atomicAdd(&s_cnt[y * gx + x], 1u);
                }
        }
        __syncthreads();
        LSR_STAMP(2);
        for (int t = threadIdx.x; t < T; t += kScatThreads) {
            const uint32_t c = s_cnt[t];
            if (c) s_base[t] = ts[t] + atomicAdd(&cur[t], c);
            s_cnt[t] = 0;
        }
        __syncthreads();
        LSR_STAMP(3);
    }
    uint32_t tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));   // recomputed per item below rather than kept (or spilled) across the passes
#pragma unroll
    for (int it = 0; it < kScatItems; ++it) {
        const uint32_t i = (uint32_t)(base + it * kScatThreads) + tid2;
        int x0, y0, x1, y1;
        unpack(it, x0, y0, x1, y1);
        if (x1 <= x0 || y1 <= y0) continue;
        // key = depth bits << 32 | index << 8 | sub-block code of THIS tile (lsr_internal.h)
        const uint64_t key = ((uint64_t)__float_as_uint(dep[it]) << 32) | (i << key_shift);
        const uint32_t sp = span[it];
        int ex0 = x0, ey0 = y0, ex1 = x1, ey1 = y1;
        if (skip_none) reached_rect(sp, ex0, ey0, ex1, ey1);
        for (int y = ey0; y < ey1; ++y)
            for (int x = ex0; x < ex1; ++x) {
                const int t = y * gx + x;
                if (over_cap && tcv[t] <= over_cap) continue;
                const uint32_t code = key_shift ? span_code(sp, x - x0, y - y0) : 0u;   // (no room for a code beside a 32-bit index)
                uint32_t pos;
                if (LDS_RESERVE) pos = s_base[t] + atomicAdd(&s_cnt[t], 1u);
                else pos = ts[t] + atomicAdd(&cur[t], 1u);
                // The position is CLAMPED to the workspace (one v_min; a store under a per-lane bounds test cost 17 % of
                // this kernel): in the no-sync forward tile_scan clamps the offsets to the capacity, in the synchronous one
                // the capacity is the host's early pair count — if that ever came out short, the surplus pairs land
                // on the last slot and the sort flags the overflow; never an out-of-bounds write.
                keys[min(pos, capacity - 1u)] = key | code;
            }
    }
    LSR_STAMP(4);
#ifdef LSR_ENABLE_TRACE
    __builtin_amdgcn_s_waitcnt(0);   // stores drained
    __syncthreads();
    LSR_STAMP(5);
    if (trace && threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trace[8 * (size_t)blockIdx.x + 7] = ((unsigned long long)(xcc & 0xff) << 32) | hwid;
    }
#endif
#undef LSR_STAMP
}

// ------------------------------------------------------------------------------------------
constexpr int kSortThreads = 512;      // workgroup of the later tiers and of the global merge path

// Where k_sort_tiles puts the two half-tile render lists of a tile (BinLayout::half_list, GeomLayout::half_count).
struct HalfOut {
    uint32_t *half_list, *half_count;
    uint32_t *header;         // geometry-workspace header (overflow flag)
    uint32_t capacity;        // pairs the binning workspace holds
    // Tiles whose list is longer than the first launch's variant handles are appended here by that launch ((view, tile)
    // indices; the scatter's cursor array is free by then) and picked up by PERSISTENT launches of the larger
    // variants: a few hundred workgroups looping over a list instead of one (large-LDS) workgroup per tile that
    // finds nothing to do — the empty second-tier launch over 4096 tiles cost 93 us.
    // Two lists share the array from its two ends (a tile is in at most one, so they cannot meet):
    //   class 0, from the front: first-tier capacity < n <= kSortTier2   -> k_sort_tiles<kSortTier2, 512, true>
    //   class 1, from the back :  n > kSortTier2                         -> k_sort_tiles<kSortLdsMax, 512, true>, and for
    //                                                                       n > kSortLdsMax the global merge path
    uint32_t *long_list;
    uint32_t long_cap;        // entries of the array (tiles of the call)
    uint32_t *long_count[2];  // header words
    IndexPacking ip;          // how keys and list entries carry the Gaussian index (lsr_internal.h)
    uint32_t seg_cap;         // single-pass binning: a list of up to seg_cap keys sits in seg_keys[vt * seg_cap ...] (0: two-phase binning)
    const uint64_t *seg_keys;
    uint32_t n_limit;         // speculative forward: the longest list the launch structure was chosen for (the caller's hint); a
                              // longer list is left unsorted with EMPTY render lists — the host sees the true longest list
                              // and runs the call again with exact sizes (UINT32_MAX otherwise)
};
constexpr int kSortTier2 = 8192;
__device__ __forceinline__ uint32_t long_tile(const HalfOut &ho, int cls, uint32_t i) {
    return ho.long_list[cls ? ho.long_cap - 1u - i : i];
}
__device__ __forceinline__ uint32_t key_index(const HalfOut &ho, uint32_t low_word) { return low_word >> ho.ip.key_shift; }
// 16-bit sub-block mask of a sorted key's low word; list entry of a half
__device__ __forceinline__ uint32_t key_mask(const HalfOut &ho, uint32_t low_word) { return ho.ip.key_shift ? code_mask(low_word & 0xFFu) : 0xFFFFu; }
__device__ __forceinline__ uint32_t list_entry(const HalfOut &ho, uint32_t idx, uint32_t bits) { return idx | ((bits << kListBitsShift) & ~ho.ip.index_mask); }
constexpr int kEmitTab = 130;   // uint64 words of LDS scratch emit_half_lists needs

// Block-wide (THREADS threads): walks the tile's depth-sorted list in order and appends every entry to
// the list of each half of the tile (pixel rows 0-7 / 8-15) its footprint can reach (order preserved), as
// `index | 8 sub-block bits << 24`; the sub-block mask comes from the code in the low key byte.
// low_word(p) = low key word of sorted position p.  A pass covers 64 wave-chunks of 64 positions: per chunk
// and half a ballot count (two 32-bit fields of one u64), one wave scans the 64 chunk totals, then every
// entry's place is chunk offset + lanes below it in the ballot.
// hdst = half_list + 2 * tile_start: the list of half h starts at hdst + h * n.
template <int THREADS, class LowWord>
__device__ __forceinline__ void emit_half_lists(const HalfOut &ho, uint32_t n, uint32_t *hdst, uint32_t *hcnt, uint64_t *s_tab, LowWord low_word) {
    const int tid = threadIdx.x, lane = tid & (LSR_WAVE - 1), wid = tid / LSR_WAVE;
    constexpr int kWaves = THREADS / LSR_WAVE;
    uint32_t run0 = 0u, run1 = 0u;
    for (uint32_t base = 0; base < n; base += LSR_WAVE * LSR_WAVE) {
        // (nothing but the running lengths lives across the barriers: the second phase reads the words again — the
        // kernel's register budget decides how many of its workgroups share a CU)
        for (int c = wid; c < LSR_WAVE; c += kWaves) {                  // chunk c: positions base + 64 c .. + 63
            const uint32_t p = base + (uint32_t)(c * LSR_WAVE + lane);
            uint64_t packed = 0;
            if (p - lane < n) {                                        // wave-uniform: chunks beyond the list only write their zero
                const uint32_t w = low_word(min(p, n - 1));            // unconditional at a clamped position
                const uint32_t m16 = p < n ? key_mask(ho, w) : 0u;
                packed = (uint64_t)__builtin_popcountll(__ballot((m16 & 0x00FFu) != 0u)) |
                         ((uint64_t)__builtin_popcountll(__ballot((m16 & 0xFF00u) != 0u)) << 32);
            }
            if (lane == 0) s_tab[c] = packed;
        }
        __syncthreads();
        if (wid == 0) {
            const uint64_t v = s_tab[lane];
            uint64_t incl = v;
#pragma unroll
            for (int off = 1; off < LSR_WAVE; off <<= 1) { const uint64_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
            s_tab[LSR_WAVE + lane] = incl - v;
            if (lane == LSR_WAVE - 1) s_tab[2 * LSR_WAVE] = incl;
        }
        __syncthreads();
        const uint64_t tot = s_tab[2 * LSR_WAVE];
        for (int c = wid; c < LSR_WAVE; c += kWaves) {
            const uint32_t p = base + (uint32_t)(c * LSR_WAVE + lane);
            if (p - lane >= n) break;                                  // wave-uniform
            const uint32_t w = low_word(min(p, n - 1));
            const uint32_t idx = key_index(ho, w), m16 = p < n ? key_mask(ho, w) : 0u;
            const uint64_t off = s_tab[LSR_WAVE + c];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t bits = half_bits(m16, h);
                const uint64_t bal = __ballot(bits != 0u);
                if (bits) {
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    hdst[(size_t)h * n + (h ? run1 : run0) + (uint32_t)(off >> (32 * h)) + below] = list_entry(ho, idx, bits);
                }
            }
        }
        run0 += (uint32_t)tot; run1 += (uint32_t)(tot >> 32);
        __syncthreads();   // s_tab is reused by the next pass
    }
    if (tid < 2) hcnt[tid] = tid == 0 ? run0 : run1;
}

// One workgroup of THREADS (= 512) threads per (tile, view); list length n <= CAP, keys in registers (CAP / THREADS per
// thread), sorted inside LDS.
//
// Order-preserving bucket sort: bucket = (key - min) >> shift is monotone in the key, so a histogram +
// scan puts every key into its final neighbourhood; inside a bucket (about one key on average) a
// key's place is the number of smaller members.  Because the low 32 key bits are the (distinct)
// Gaussian indices, even identical depths spread over the buckets; only a heavy cluster plus a far
// outlier can overfill a bucket, and such tiles take the bitonic network below (same LDS array).
// Both paths produce the same total order of distinct 64-bit keys, i.e. the bit-exact published
// ordering.
//
// The list is read from memory once into registers; the returning histogram atomic is the key's arrival rank
// in its bucket, so after the scan every key is placed with a plain LDS write, and its final position is found
// by EVERY KEY IN PARALLEL (count the smaller members of its bucket, four per step).  The round-1 version
// finished buckets with a serial insertion sort, four buckets per thread: a chain of dependent LDS round trips
// that was half of the kernel (phase trace: 5.1 of 10.2 us per workgroup).  All LDS scratch lives in the dynamic
// allocation (the reduction scratch aliases the key array before keys are placed).
//
// Round 4, measured and rejected (tools/ab_knobs.py): smaller workgroups for shorter lists (128 / 256 / 384
// threads for up to 1024 / 2048 / 3072 keys, always eight keys per thread, i.e. more TILES in flight per CU: 0.0675 vs
// 0.0597 ms on a scene whose longest list is 2 741 keys — the 384-thread variant loses 13 %); persistent workgroups that
// request the next tile's keys while the current tile's lists are written and count the half-list entries while the
// sorted words are placed (no counting pass, two barriers fewer): 0.074 ms at 80 VGPRs with spills, 0.086 at 96, against
// 0.065 for this kernel — on gfx9 the wait for the prefetched keys also waits for the list stores issued behind them.
constexpr uint32_t kBucketOverflow = 48;   // longest bucket the in-bucket pass may get

// tier: 0 = first launch of a call (also owns the empty lists), 1 = a later tier (its tiles come from a long list)
template <int CAP, int THREADS>
__device__ __forceinline__ void sort_tile(uint64_t *s_keys, size_t vt, int tier, const uint32_t *__restrict__ tile_start,
                                          const uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list,
                                          const HalfOut &ho, unsigned long long *trace) {
#ifdef LSR_ENABLE_TRACE
#define LSR_STAMP(k) do { if (trace && threadIdx.x == 0) trace[8 * (size_t)blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define LSR_STAMP(k) do { } while (0)
#endif
    LSR_STAMP(0);
    static_assert(CAP % THREADS == 0, "whole keys per thread");
    constexpr int NB = CAP < 2048 ? CAP : 2048;          // buckets
    constexpr int kWaves = THREADS / LSR_WAVE;
    uint32_t *s_cnt = (uint32_t *)(s_keys + CAP);         // [NB] histogram -> bucket offsets
    uint64_t *s_red = s_keys;                             // [2 * kWaves] range reduction  } alias the key array:
    uint32_t *s_wsum = (uint32_t *)(s_keys + 2 * kWaves); // [kWaves] scan partials        } dead before the
    uint32_t *s_flag = s_wsum + kWaves;                   // bucket overflow               } first key is placed
    const int tid = threadIdx.x, lane = tid & (LSR_WAVE - 1), wid = tid / LSR_WAVE;
    const uint32_t start = tile_start[vt], n = tile_start[vt + 1] - start;
    uint32_t *hcnt = ho.half_count + 2 * vt, *hdst = ho.half_list + 2 * (size_t)start;
    if (n == 0) { if (tid < 2) hcnt[tid] = 0; return; }
    if ((uint64_t)start + n > ho.capacity) {      // a list beyond the workspace (synchronous forward: the host's count was short): flag, render nothing
        if (tid < 2) hcnt[tid] = 0;
        if (tid == 0) ho.header[kHdrOverflow] = 1u;
        return;
    }
    if (n > ho.n_limit) {         // speculative forward, hint too small: nothing of this tile may reach the compositing kernels
        if (tid < 2) hcnt[tid] = 0;
        return;
    }
    if (n > (uint32_t)CAP) {        // for a later tier: a larger LDS variant, or (beyond the largest) the global merge path
        if (tid == 0 && tier == 0) {
            const int cls = n > (uint32_t)kSortTier2 ? 1 : 0;
            const uint32_t i = atomicAdd(ho.long_count[cls], 1u);
            ho.long_list[cls ? ho.long_cap - 1u - i : i] = (uint32_t)vt;
        }
        return;
    }
    // single-pass binning: a list that fits its fixed-capacity segment was written there by the projection kernel; an
    // overfull tile's keys were placed at its exact offsets of the binning workspace by the fallback scatter
    const uint64_t *src = (ho.seg_cap && n <= ho.seg_cap) ? ho.seg_keys + vt * (size_t)ho.seg_cap : keys + start;
    if (n == 1) {
        if (tid == 0) {
            const uint32_t w = (uint32_t)src[0], idx = key_index(ho, w);
            point_list[start] = idx;
            const uint32_t m = key_mask(ho, w);
            for (int h = 0; h < 2; ++h) {
                const uint32_t bits = half_bits(m, h);
                if (bits) hdst[h] = list_entry(ho, idx, bits);
                hcnt[h] = bits ? 1u : 0u;
            }
        }
        return;
    }

    constexpr int PERK = CAP / THREADS;   // keys per thread, register-resident (the 16 384-key variant: 32 keys, 256 VGPRs)
    uint64_t kreg[PERK];
#pragma unroll
    for (int q = 0; q < PERK; ++q) {
        const uint32_t i = tid + q * THREADS;
        const uint64_t k = src[min(i, n - 1)];   // unconditional (clamped) so that all loads are in flight together
        kreg[q] = i < n ? k : ~0ull;
    }
    // ---- key range ----
    uint64_t kmin = ~0ull, kmax = 0ull;
#pragma unroll
    for (int q = 0; q < PERK; ++q) {
        const uint64_t k = kreg[q];
        kmin = k < kmin ? k : kmin;
        if (tid + q * THREADS < n) kmax = k > kmax ? k : kmax;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t a = __shfl_xor(kmin, off), b = __shfl_xor(kmax, off);
        kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) { s_red[2 * wid] = kmin; s_red[2 * wid + 1] = kmax; }
    for (int b = tid; b < NB; b += THREADS) s_cnt[b] = 0;
    if (tid == 0) *s_flag = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        kmin = s_red[2 * w] < kmin ? s_red[2 * w] : kmin;
        kmax = s_red[2 * w + 1] > kmax ? s_red[2 * w + 1] : kmax;
    }
    LSR_STAMP(1);
    const uint64_t range = kmax - kmin;
    const int bits = range ? 64 - __builtin_clzll(range) : 0;      // range < 2^bits
    constexpr int LOGNB = NB == 4096 ? 12 : (NB == 2048 ? 11 : 10);
    const int shift = bits > LOGNB ? bits - LOGNB : 0;

    // ---- histogram (the returning atomic is the key's rank inside its bucket), scan ----
    uint32_t rnk[PERK];
#pragma unroll
    for (int q = 0; q < PERK; ++q)   // (under the validity test: padding lanes on one dummy word serialise — measured 0.048 -> 0.077 ms)
        if (tid + q * THREADS < n) rnk[q] = atomicAdd(&s_cnt[(uint32_t)((kreg[q] - kmin) >> shift)], 1u);
    __syncthreads();
    LSR_STAMP(2);
    constexpr int PER = (NB + THREADS - 1) / THREADS;   // buckets per thread of the scan (the last threads may own fewer)
    uint32_t loc[PER], sum = 0, mx = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        loc[q] = (NB % THREADS == 0 || tid * PER + q < NB) ? s_cnt[min(tid * PER + q, NB - 1)] : 0u;
        sum += loc[q]; mx = loc[q] > mx ? loc[q] : mx;
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < LSR_WAVE; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == LSR_WAVE - 1) s_wsum[wid] = incl;
    if (mx > kBucketOverflow) *s_flag = 1;
    __syncthreads();
    uint32_t run = incl - sum;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) run += w < wid ? s_wsum[w] : 0u;
    const bool overflow = *s_flag != 0;
    if (!overflow) {
#pragma unroll
        for (int q = 0; q < PER; ++q) { if (NB % THREADS == 0 || tid * PER + q < NB) s_cnt[tid * PER + q] = run; run += loc[q]; }   // exclusive starts
        __syncthreads();   // also: every thread is done with the aliased scratch
        LSR_STAMP(3);
        // ---- place: start of the bucket + arrival rank ----
#pragma unroll
        for (int q = 0; q < PERK; ++q)
            if (tid + q * THREADS < n) s_keys[s_cnt[(uint32_t)((kreg[q] - kmin) >> shift)] + rnk[q]] = kreg[q];
        __syncthreads();
        LSR_STAMP(4);
        // ---- final position of every key: bucket start + number of smaller members ----
        uint32_t dst[PERK];
#pragma unroll
        for (int q = 0; q < PERK; ++q) {
            dst[q] = 0;
            if (tid + q * THREADS < n) {
                const uint64_t key = kreg[q];
                const uint32_t b = (uint32_t)((key - kmin) >> shift);
                const uint32_t lo = s_cnt[b], hi = b + 1 < (uint32_t)NB ? s_cnt[b + 1] : n;
                uint32_t below = 0;
                for (uint32_t j = lo; j < hi; j += 4) {
                    const uint32_t last = hi - 1;
                    const uint64_t m0 = s_keys[j], m1 = s_keys[min(j + 1, last)], m2 = s_keys[min(j + 2, last)], m3 = s_keys[min(j + 3, last)];
                    below += (m0 < key) + (j + 1 < hi && m1 < key) + (j + 2 < hi && m2 < key) + (j + 3 < hi && m3 < key);
                }
                dst[q] = lo + below;
            }
        }
        __syncthreads();   // all reads of the key array done: its storage becomes the sorted list of low key words
        uint32_t *s_out = (uint32_t *)s_keys;
#pragma unroll
        for (int q = 0; q < PERK; ++q)
            if (tid + q * THREADS < n) s_out[dst[q]] = (uint32_t)kreg[q];
        __syncthreads();
        LSR_STAMP(5);
        for (uint32_t i = tid; i < n; i += THREADS) point_list[start + i] = key_index(ho, s_out[i]);
        // half-tile render lists (the bucket offsets in s_cnt are dead: scratch of the emitter)
        emit_half_lists<THREADS>(ho, n, hdst, hcnt, (uint64_t *)s_cnt, [&](uint32_t p) { return s_out[p]; });
        LSR_STAMP(6);
    } else {
        // ---- bitonic network over the padded list ----
        uint32_t npad = 2;
        while (npad < n) npad <<= 1;
        __syncthreads();
        if (npad <= (uint32_t)CAP) {
#pragma unroll
            for (int q = 0; q < PERK; ++q)
                if (tid + q * THREADS < npad) s_keys[tid + q * THREADS] = kreg[q];   // padding = ~0
        } else {
            // (CAP = 3072: a list of 2049 .. 3072 keys pads to 4096 > CAP; the key array and the bucket counters behind it
            // hold 3072 + 1024 keys' worth of bytes — exactly the padded list)
            static_assert(CAP * 8 + NB * 4 >= (CAP <= 2048 ? CAP : (CAP <= 4096 ? 4096 : CAP)) * 8, "bitonic padding fits the allocation");
#pragma unroll
            for (int q = 0; q < PERK; ++q) s_keys[tid + q * THREADS] = kreg[q];
            for (uint32_t i = CAP + tid; i < npad; i += THREADS) s_keys[i] = ~0ull;
        }
        __syncthreads();
        for (uint32_t k = 2; k <= npad; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < (npad >> 1); t += THREADS) {
                    // t-th compare-exchange of this stage: partner indices differ in bit j
                    const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const uint32_t hi = lo | j;
                    const bool up = (lo & k) == 0;
                    const uint64_t a = s_keys[lo], b = s_keys[hi];
                    if ((a > b) == up) { s_keys[lo] = b; s_keys[hi] = a; }
                }
                __syncthreads();
            }
        }
        // (the emitter's scratch, the counter array, only overlaps padding keys: it reads positions < n <= CAP)
        for (uint32_t i = tid; i < n; i += THREADS) point_list[start + i] = key_index(ho, (uint32_t)s_keys[i]);
        emit_half_lists<THREADS>(ho, n, hdst, hcnt, (uint64_t *)s_cnt, [&](uint32_t p) { return (uint32_t)s_keys[p]; });
    }
#ifdef LSR_ENABLE_TRACE
    if (trace && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trace[8 * (size_t)blockIdx.x + 7] = ((unsigned long long)n << 40) | ((unsigned long long)(xcc & 0xff) << 32) | hwid;
    }
#endif
#undef LSR_STAMP
}

// PERSISTENT = false: one workgroup per (view, tile) (first tier), in the costliest-first order of the work items when
// `order` is given (longest-processing-time first: the launch runs several rounds of resident workgroups and should
// not end on its longest lists); true: workgroups looping over the previous tier's long list.  (Two kernels rather
// than one with a runtime switch: with both paths in one body the first-tier variant needed 112 instead of 80 VGPRs
// and lost a resident workgroup per CU.)
template <int CAP, int THREADS, bool PERSISTENT>
__global__ void __launch_bounds__(THREADS, CAP <= 4096 ? 6 : (CAP <= 8192 ? 4 : 2))   // waves per SIMD: the register budget follows
k_sort_tiles(int cls, const uint32_t *__restrict__ order, const uint32_t *__restrict__ tile_start, const uint64_t *__restrict__ keys,
             uint32_t *__restrict__ point_list, HalfOut ho, unsigned long long *trace) {
    extern __shared__ uint64_t s_keys[];                  // [CAP] keys grouped by bucket / sorted, then [NB] u32 counters
    if (!PERSISTENT) {
        const size_t vt = order ? (size_t)(order[2 * (size_t)blockIdx.x] & kItemTileMask) : (size_t)blockIdx.x;
        sort_tile<CAP, THREADS>(s_keys, vt, 0, tile_start, keys, point_list, ho, trace);
    } else {
        const uint32_t count = *ho.long_count[cls];       // persistent launches: the class of long tiles this launch owns
        for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
            sort_tile<CAP, THREADS>(s_keys, long_tile(ho, cls, i), 1, tile_start, keys, point_list, ho, trace);
            __syncthreads();   // the LDS arrays are reused by the next tile
        }
    }
}

// Global-memory path for lists longer than the LDS capacity: bottom-up merge sort by one
// workgroup per oversized tile (rank-by-binary-search merges, ping-pong between keys and tmp).
// Rare (needs > kSortLdsMax Gaussians over one 16x16 tile); correctness path, not tuned.
__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t *a, uint32_t n, uint64_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
// one oversized tile; s_tab: kEmitTab words of LDS scratch
__device__ __forceinline__ void sort_tile_global(size_t vt, const uint32_t *__restrict__ tile_start, uint64_t *keys, uint64_t *tmp,
                                                 uint32_t *point_list, const HalfOut &ho, uint64_t *s_tab) {
    const uint32_t start = tile_start[vt], n = tile_start[vt + 1] - start;
    uint64_t *src = keys + start, *dst = tmp + start;
    // width-1 runs are trivially sorted; keys are unique (index in the low word) so ranks are exact
    for (uint32_t w = 1; w < n; w <<= 1) {
        for (uint32_t i = threadIdx.x; i < n; i += kSortThreads) {
            const uint32_t pair = i / (2 * w), a0 = pair * 2 * w;
            const uint32_t a1 = min(a0 + w, n), b1 = min(a0 + 2 * w, n);
            const uint64_t x = src[i];
            uint32_t pos;
            if (i < a1) pos = a0 + (i - a0) + lower_bound_u64(src + a1, b1 - a1, x);
            else pos = a0 + (i - a1) + lower_bound_u64(src + a0, a1 - a0, x);
            dst[pos] = x;
        }
        __threadfence_block();
        __syncthreads();
        uint64_t *t = src; src = dst; dst = t;
    }
    for (uint32_t i = threadIdx.x; i < n; i += kSortThreads)
        point_list[start + i] = key_index(ho, (uint32_t)src[i]);
    const uint64_t *sorted = src;
    emit_half_lists<kSortThreads>(ho, n, ho.half_list + 2 * (size_t)start, ho.half_count + 2 * vt, s_tab, [&](uint32_t p) { return (uint32_t)sorted[p]; });
}
__global__ void __launch_bounds__(kSortThreads)
k_sort_tiles_global(const uint32_t *__restrict__ tile_start, uint64_t *keys, uint64_t *tmp, uint32_t *point_list, HalfOut ho) {
    __shared__ uint64_t s_tab[kEmitTab];
    const uint32_t count = *ho.long_count[1];
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {   // persistent over the tiles beyond the LDS sort
        const size_t vt = long_tile(ho, 1, it);
        if (tile_start[vt + 1] - tile_start[vt] <= (uint32_t)kSortLdsMax) continue;   // the class's shorter lists belong to the largest LDS variant
        sort_tile_global(vt, tile_start, keys, tmp, point_list, ho, s_tab);
        __syncthreads();
    }
}
// The no-sync forward cannot know whether any list exceeds the first tier: instead of three mostly empty launches (the
// two persistent LDS tiers and the merge path) it issues ONE, whose workgroups walk both long-tile lists with the
// largest LDS variant and the merge path behind it.
__global__ void __launch_bounds__(kSortThreads, 2)
k_sort_tiles_long(const uint32_t *__restrict__ tile_start, uint64_t *keys, uint64_t *tmp, uint32_t *point_list, HalfOut ho) {
    extern __shared__ uint64_t s_keys[];
    const uint32_t c0 = *ho.long_count[0], c1 = *ho.long_count[1];
    for (uint32_t it = blockIdx.x; it < c0 + c1; it += gridDim.x) {
        const size_t vt = it < c0 ? long_tile(ho, 0, it) : long_tile(ho, 1, it - c0);
        if (tile_start[vt + 1] - tile_start[vt] <= (uint32_t)kSortLdsMax) sort_tile<kSortLdsMax, kSortThreads>(s_keys, vt, 1, tile_start, keys, point_list, ho, nullptr);
        else sort_tile_global(vt, tile_start, keys, tmp, point_list, ho, s_keys);   // (its scratch: the start of the same allocation)
        __syncthreads();
    }
}

// Dynamic LDS of a sort variant, and (once per process, device and variant) the function attribute the larger ones need.
template <int CAP>
constexpr size_t sort_lds_bytes() { return (size_t)CAP * 8 + (size_t)(CAP < 2048 ? CAP : 2048) * 4; }
template <int CAP, int THREADS, bool PERS>
static void sort_launch(dim3 grid, hipStream_t s, int cls, const uint32_t *order, const uint32_t *ts, const uint64_t *keys,
                        uint32_t *plist, const HalfOut &ho, unsigned long long *trace) {
    constexpr size_t lds = sort_lds_bytes<CAP>();
    if (lds > 48 * 1024) {
        static uint64_t done = 0;     // one bit per device id
        static std::mutex mu;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || dev >= 64 || !((done >> dev) & 1ull)) {
            (void)hipFuncSetAttribute((const void *)k_sort_tiles<CAP, THREADS, PERS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (dev >= 0 && dev < 64) done |= 1ull << dev;
        }
    }
    hipLaunchKernelGGL((k_sort_tiles<CAP, THREADS, PERS>), grid, dim3(THREADS), lds, s, cls, order, ts, keys, plist, ho, trace);
}

hipError_t launch_binning(const lsr_dims &d, char *geom, char *bin, int64_t num_pairs,
                          int32_t max_tile_pairs, const int32_t *radii, hipStream_t s, bool device_counts, bool seg, bool speculative) {
    (void)radii;
    const GeomLayout L = geom_layout(d);
    if (num_pairs <= 0 || d.num_gaussians == 0)   // nothing to sort: every half-tile render list is empty
        return launch_clear(geom + L.half_count, align_up((size_t)d.num_views * (size_t)num_tiles(d) * 8, 16), s);
    // no-sync forward: the merge scratch is always part of the layout (the longest list is unknown)
    const BinLayout B = bin_layout(d, num_pairs, device_counts ? kSortLdsMax + 1 : max_tile_pairs);
    const int T = (int)num_tiles(d), gx = tiles_x(d);
    // single-pass binning: the projection kernel has already written the keys into the tile segments of the geometry
    // workspace — no scatter, and the keys area of the binning workspace stays untouched
    if (seg && L.seg_cap == 0) return hipErrorInvalidValue;
    uint64_t *keys = (uint64_t *)(bin + B.keys);
    uint32_t *plist = (uint32_t *)(bin + B.point_list);
    const uint32_t *ts = (const uint32_t *)(geom + L.tile_start);
    HalfOut ho;
    ho.half_list = (uint32_t *)(bin + B.half_list); ho.half_count = (uint32_t *)(geom + L.half_count);
    ho.header = (uint32_t *)(geom + L.header);
    ho.capacity = (uint32_t)(num_pairs < 0xFFFFFFFFll ? num_pairs : 0xFFFFFFFFll);
    // long-tile lists: the two ends of the scatter cursor array (free once k_scatter is done; a list holds at most
    // the call's tiles), counters in the header (cleared per forward)
    ho.long_list = (uint32_t *)(geom + L.tile_cursor);
    ho.long_cap = (uint32_t)d.num_views * (uint32_t)T;
    ho.long_count[0] = ho.header + kHdrLongTiles; ho.long_count[1] = ho.long_count[0] + 1;
    ho.ip = index_packing(d);
    ho.seg_cap = seg ? L.seg_cap : 0u;
    ho.seg_keys = (const uint64_t *)(geom + L.seg_keys);
    ho.n_limit = speculative ? (uint32_t)max_tile_pairs : 0xFFFFFFFFu;
    // the scatter: everything (two-phase binning), or only the tiles whose lists outgrew their segments — launched when
    // the host knows of such a list, or cannot know (no-sync forward: the workgroups leave at once when there is none)
    const uint32_t over_cap = seg ? L.seg_cap : 0u;
    if (!seg || device_counts || (uint32_t)max_tile_pairs > L.seg_cap) {
        const bool lds = T <= 8192;
        // Items per thread: the launch should be ONE round of resident workgroups (8 per CU, fewer when
        // the per-tile counters of a large image take the LDS) — with a fixed 2048 Gaussians per block the
        // 16-view headline needed 9.2 blocks per CU and ran a second, 15 %-full round (phase trace).
        const int64_t resident = (int64_t)device_cus() * (lds ? std::max<int64_t>(1, std::min<int64_t>(8, (160 * 1024) / ((int64_t)T * 8 + 64))) : 8);
        const int64_t work = (int64_t)d.num_views * d.num_gaussians;
        const int64_t rounds = (work + resident * kScatThreads * kScatItems - 1) / (resident * kScatThreads * kScatItems);
        int items = (int)((work + resident * kScatThreads * rounds - 1) / (resident * kScatThreads * rounds));
        // ... but never more than ~160 workgroups per view: every workgroup reserves its share of a tile's segment with ONE
        // global atomic per tile, and the workgroups of a view hit the same T cursor words — same-address atomics serialise
        // at ~60 ns each.  With one Gaussian per thread a single 300 k view ran 1 172 workgroups and spent 55 of its 66 us
        // queueing on those words (round 4: V = 1 scatter 0.066 -> see DESIGN.md §4; V = 4 0.027).
        const int min_items = (int)((d.num_gaussians + (int64_t)kScatThreads * 160 - 1) / ((int64_t)kScatThreads * 160));
        items = std::max(1, std::min(kScatItems, std::max(items, min_items)));
        const uint32_t chunks = (uint32_t)((d.num_gaussians + kScatThreads * items - 1) / (kScatThreads * items));
        dim3 grid(chunks * (uint32_t)d.num_views);
        unsigned long long *strace = nullptr;
#ifdef LSR_ENABLE_TRACE
        const char *strace_path = getenv("LSR_TRACE_SCATTER");
        if (strace_path) {
            (void)hipMalloc((void **)&strace, (size_t)grid.x * 64);
            (void)hipMemsetAsync(strace, 0, (size_t)grid.x * 64, s);
        }
#endif
        prof_begin(kStScatter, s);
        const uint32_t capacity = (uint32_t)(num_pairs < 0xFFFFFFFFll ? num_pairs : 0xFFFFFFFFll);
#define LSR_SCAT2(LDSR, NRW, SHM)                                                                         \
    hipLaunchKernelGGL((k_scatter<LDSR, NRW>), grid, dim3(kScatThreads), SHM, s, d.num_gaussians, gx, T, \
                       (const char *)(geom + L.bin), ts, (uint32_t *)(geom + L.tile_cursor), keys, capacity, chunks, items, index_packing(d).key_shift, strace, \
                       (const uint32_t *)(geom + L.tile_count), (const uint32_t *)(geom + L.header), over_cap, reached_only(d) ? 1 : 0)
#define LSR_SCAT(LDSR, SHM) do { if (narrow_bins(d)) LSR_SCAT2(LDSR, true, SHM); else LSR_SCAT2(LDSR, false, SHM); } while (0)
        if (lds) LSR_SCAT(true, (size_t)T * 8);
        else LSR_SCAT(false, 0);
#undef LSR_SCAT
#undef LSR_SCAT2
        prof_end(kStScatter, s);
#ifdef LSR_ENABLE_TRACE
        if (strace_path) {
            std::vector<unsigned long long> host((size_t)grid.x * 8);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(host.data(), strace, host.size() * 8, hipMemcpyDeviceToHost);
            if (FILE *f = fopen(strace_path, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
            (void)hipFree(strace);
        }
#endif
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    {
        dim3 grid((uint32_t)T * (uint32_t)d.num_views);
        unsigned long long *trace = nullptr;
#ifdef LSR_ENABLE_TRACE
        const char *trace_path = getenv("LSR_TRACE_SORT");
        if (trace_path) {
            (void)hipMalloc((void **)&trace, (size_t)grid.x * 64);
            (void)hipMemsetAsync(trace, 0, (size_t)grid.x * 64, s);
        }
#endif
        // Tiers: (1) one 512-thread workgroup per tile, in costliest-first order, with the variant that fits the longest
        // list the host knows of (1024 / 2048 / 4096 keys); longer lists are appended to one of two lists by that launch
        // and sorted by persistent launches of (2) the 8192-key variant (two workgroups per CU), (3) the 16 384-key variant
        // (one) and (4) the global merge path.  The later tiers are
        // launched when the host knows they are needed (synchronous forward) or cannot know (no-sync forward: a few
        // hundred workgroups each that find an empty list; its first tier is always the 4096-key variant, whatever the
        // caller's hint says, so that only lists beyond 4096 keys take the later tiers).
        prof_begin(kStSort, s);
        const uint32_t *order = env_int("LSR_SORT_LPT", 1) ? (const uint32_t *)(geom + L.tile_order) : nullptr;
        const int32_t longest = device_counts ? std::max<int32_t>(max_tile_pairs, 4096) : max_tile_pairs;
        int32_t tier1;     // capacity of the first tier
        if (longest <= 1024) { tier1 = 1024; sort_launch<1024, 512, false>(grid, s, 0, order, ts, keys, plist, ho, trace); }
        else if (longest <= 2048) { tier1 = 2048; sort_launch<2048, 512, false>(grid, s, 0, order, ts, keys, plist, ho, trace); }
        else { tier1 = 4096; sort_launch<4096, 512, false>(grid, s, 0, order, ts, keys, plist, ho, trace); }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const uint32_t cus = (uint32_t)device_cus();
        const dim3 pgrid(cus);
        if (device_counts) {   // one launch for whatever the first tier left over (usually nothing)
            constexpr size_t lds = sort_lds_bytes<kSortLdsMax>();
            static uint64_t attr_done = 0;     // one bit per device id
            static std::mutex attr_mu;
            int dev = 0;
            (void)hipGetDevice(&dev);
            {
                std::lock_guard<std::mutex> lock(attr_mu);
                if (dev < 0 || dev >= 64 || !((attr_done >> dev) & 1ull)) {
                    (void)hipFuncSetAttribute((const void *)k_sort_tiles_long, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (dev >= 0 && dev < 64) attr_done |= 1ull << dev;
                }
            }
            hipLaunchKernelGGL(k_sort_tiles_long, pgrid, dim3(kSortThreads), lds, s, ts, keys, (uint64_t *)(bin + B.tmp), plist, ho);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        } else {
        if (max_tile_pairs > tier1) {   // class 0: two 72 KB workgroups per CU
            sort_launch<kSortTier2, 512, true>(dim3(2 * cus), s, 0, nullptr, ts, keys, plist, ho, trace);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        if (max_tile_pairs > kSortTier2) {   // class 1 up to the LDS capacity
            sort_launch<kSortLdsMax, 512, true>(pgrid, s, 1, nullptr, ts, keys, plist, ho, trace);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        if (max_tile_pairs > kSortLdsMax) {
            hipLaunchKernelGGL(k_sort_tiles_global, pgrid, dim3(kSortThreads), 0, s, ts, keys, (uint64_t *)(bin + B.tmp), plist, ho);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        }
        prof_end(kStSort, s);
#ifdef LSR_ENABLE_TRACE
        if (trace_path) {
            std::vector<unsigned long long> host((size_t)grid.x * 8);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(host.data(), trace, host.size() * 8, hipMemcpyDeviceToHost);
            if (FILE *f = fopen(trace_path, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
            (void)hipFree(trace);
        }
#endif
    }
    return hipSuccess;
}

}  // namespace lsr
