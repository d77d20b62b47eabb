// lsr_tile_scan.h — what ONE workgroup does with the per-tile pair counts of a forward call (stage K2-K3 of the
// published pipeline: "prefix sum of tiles touched"):
//   * exclusive scan of the V*T counts -> tile_start (clamped to the pair capacity), total pair count, longest list,
//     overflow flag (header words), the two numbers the synchronous forward's host waits for (mapped host words);
//   * the compositing work items — two (view, tile, half) items per tile — in longest-list-first order (counting
//     sort on THREADS cost classes): the work-queue order of both compositing kernels and of the per-tile sort.
// Used by the LAST workgroup of k_preprocess (THREADS = its 256 threads, the counts read with agent-scope atomic loads:
// round 4 — the stage used to be a kernel of its own on the critical path of every forward) and by k_tile_scan, the
// stand-alone kernel for calls with more tiles than that workgroup handles from registers.
#pragma once
#include "lsr_internal.h"

namespace lsr {

constexpr int kScanClasses = 1024;   // cost classes of the work-item order (whatever the workgroup size: with the folded
                                     // instance's 256 threads = 256 classes the bench scene's lists fell into 43 classes
                                     // instead of 172 and the compositing backward, whose first 4096 items are assigned
                                     // statically in this order, ran 3 % longer)
template <int THREADS>
struct TileScanShared {
    uint32_t wave[THREADS / LSR_WAVE];
    uint32_t mx[THREADS / LSR_WAVE];
    uint64_t tot64[THREADS / LSR_WAVE];
    uint32_t cls[kScanClasses];      // counting-sort classes: histogram -> running offsets
};

// What the host of a synchronous forward reads from its mapped words: [0] pair count (saturated), [1] longest list,
// [2] sequence number of the call (written LAST, system scope: the host may poll it while the kernel is still running).
struct HostMirror {
    uint32_t *words;     // device pointer to the mapped host words, or nullptr
    uint32_t seq;
};

template <bool ATOMIC>
__device__ __forceinline__ uint32_t scan_load(const uint32_t *p) {
    // ATOMIC: the counts were produced by agent-scope atomics of OTHER workgroups of the same kernel; they are read past
    // the (mutually incoherent) per-XCD L2s the same way
    if (ATOMIC) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// Block-wide exclusive prefix sum of one value per thread: wave scans in registers + one LDS hop (two barriers).
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    constexpr int kWaves = THREADS / LSR_WAVE;
    const int lane = threadIdx.x & (LSR_WAVE - 1), wid = threadIdx.x / LSR_WAVE;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < LSR_WAVE; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == LSR_WAVE - 1) s_wave[wid] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t x = s_wave[w];
        tot += x;
        base += w < wid ? x : 0u;
    }
    __syncthreads();   // s_wave may be reused
    total = tot;
    return base + incl - v;
}

// REGS = counts a thread keeps in registers; longer chunks (and REGS = 0: the folded instance, whose counts sit in
// LDS) are re-read by every pass.
template <int THREADS, int REGS, bool ATOMIC>
__device__ __forceinline__ void tile_scan_block(const uint32_t *count, uint32_t *__restrict__ start, uint32_t *header,
                                                HostMirror hm, uint32_t *__restrict__ order, int N, uint32_t capacity,
                                                TileScanShared<THREADS> &sh) {
    constexpr int kWaves = THREADS / LSR_WAVE;
    const int tid = threadIdx.x, lane = tid & (LSR_WAVE - 1), wid = tid / LSR_WAVE;
    const int per = (N + THREADS - 1) / THREADS;
    const int lo = tid * per, hi = min(N, lo + per);
    const bool in_regs = REGS > 0 && per <= REGS; // block-uniform
    // chunks of whole, aligned 16-byte groups (the folded instance: 4096 counts in LDS, 16 per thread): every pass reads
    // four counts per instruction and the reads of a pass are independent of each other
    const bool vec4 = !ATOMIC && !in_regs && (per & 3) == 0 && per <= 16 && (N % per) == 0 && ((size_t)count & 15) == 0;
    uint32_t creg[REGS > 0 ? REGS : 1];
    if (in_regs) {
        // unconditional loads, all in flight together, from ONE base address with immediate offsets (sixteen clamped
        // addresses cost the folded instance 30 registers and k_preprocess a wave per SIMD): a chunk that starts
        // beyond the counts reads the REGS words behind them instead — the count array is followed by at least 64
        // readable words of the same workspace (GeomLayout: tile_cursor) — and every value is masked below
        const uint32_t *src = count + min(lo, N);
#pragma unroll
        for (int q = 0; q < REGS; ++q) {
            const uint32_t c = scan_load<ATOMIC>(src + q);
            creg[q] = (q < per && lo + q < hi) ? c : 0u;
        }
    }
    uint32_t sum = 0, mx = 0;
    uint64_t sum64 = 0;
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < REGS; ++q) { sum += creg[q]; mx = max(mx, creg[q]); }
        sum64 = sum;                              // at most REGS counts of < 2^28 each per thread
    } else if (vec4) {
        for (int i = lo; i < hi; i += 4) {
            const uint4 c = *(const uint4 *)(count + i);
            sum += c.x + c.y + c.z + c.w; mx = max(max(mx, c.x), max(max(c.y, c.z), c.w));
        }
        sum64 = sum;                              // a chunk of < 2^4 counts of < 2^28 each
    } else {
        for (int i = lo; i < hi; ++i) { const uint32_t c = scan_load<ATOMIC>(&count[i]); sum += c; sum64 += c; mx = max(mx, c); }
    }
    // one combined pass: exclusive scan of the chunk sums, the longest list, the 64-bit total (two barriers)
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < LSR_WAVE; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, off)); sum64 += __shfl_xor(sum64, off); }
    if (lane == LSR_WAVE - 1) sh.wave[wid] = incl;
    if (lane == 0) { sh.mx[wid] = mx; sh.tot64[wid] = sum64; }
    __syncthreads();
    uint32_t wbase = 0, total = 0, maxc = 0;
    sum64 = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t x = sh.wave[w];
        total += x;
        wbase += w < wid ? x : 0u;
        maxc = max(maxc, sh.mx[w]);
        sum64 += sh.tot64[w];
    }
    __syncthreads();   // sh.wave is reused below
    // Offsets are 32-bit: a call whose pair count does not fit (possible in principle: every Gaussian can touch every
    // tile of every view) is reported as an overflow with the count saturated, never silently wrapped.
    const bool wrapped = sum64 > 0xFFFFFFFFull;
    if (wrapped) total = 0xFFFFFFFFu;
    // The two numbers the host of a synchronous forward is waiting for go out FIRST, straight into its mapped, pinned
    // memory; the sequence word behind a system-scope release lets the host poll for them while this workgroup goes on.
    if (hm.words && tid == 0) {
        hm.words[0] = total; hm.words[1] = maxc;
        __hip_atomic_store(&hm.words[2], hm.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    uint32_t run = wbase + incl - sum;            // exclusive prefix of this thread's chunk
    // Offsets are clamped to the capacity of the binning workspace: with exact sizing (capacity = UINT32_MAX) nothing
    // changes; in the no-sync forward a scene that produces more pairs than the caller provided for gets its last lists
    // truncated (never an out-of-bounds write) and the overflow word set — the caller must then discard the result and
    // retry with more room.
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < REGS; ++q) { if (lo + q < hi) start[lo + q] = min(run, capacity); run += creg[q]; }
    } else if (vec4) {
        for (int i = lo; i < hi; i += 4) {
            const uint4 c = *(const uint4 *)(count + i);
            uint4 o;
            o.x = min(run, capacity); run += c.x; o.y = min(run, capacity); run += c.y;
            o.z = min(run, capacity); run += c.z; o.w = min(run, capacity); run += c.w;
            *(uint4 *)(start + i) = o;
        }
    } else {
        for (int i = lo; i < hi; ++i) { start[i] = min(run, capacity); run += scan_load<ATOMIC>(&count[i]); }
    }
    if (tid == THREADS - 1) {
        start[N] = min(total, capacity); header[kHdrPairs] = total; header[kHdrMaxTile] = maxc;
        header[kHdrOverflow] = (total > capacity || wrapped) ? 1u : 0u;
    }
    // ---- work items, costliest first: counting sort on THREADS classes of the per-tile cost (= canonical list
    // length; the two half-tile items of a tile stay together).  The exact half list lengths only exist after
    // k_sort_tiles; the order is a scheduling hint (longest-processing-time first for the work queues), never a
    // correctness matter. ----
    // class = THREADS - 1 - floor(w THREADS / (maxc + 1)), in float (a software 64-bit division per count was a third of
    // this stage; the classes only steer scheduling, and both passes below evaluate the same expression)
    const float cls_scale = (float)kScanClasses / ((float)maxc + 1.0f);
    auto cls = [&](uint32_t w) -> uint32_t { return kScanClasses - 1 - min((uint32_t)((float)w * cls_scale), (uint32_t)(kScanClasses - 1)); };
    if (tid == 0) header[kHdrNumItems] = 2u * (uint32_t)N;
    constexpr int CPT = kScanClasses / THREADS;   // classes per thread of the offset scan
    static_assert(kScanClasses % THREADS == 0, "whole classes per thread");
#pragma unroll
    for (int k = 0; k < CPT; ++k) sh.cls[tid * CPT + k] = 0;
    __syncthreads();
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < REGS; ++q) if (lo + q < hi) atomicAdd(&sh.cls[cls(creg[q])], 2u);
    } else if (vec4) {
        for (int i = lo; i < hi; i += 4) {
            const uint4 c = *(const uint4 *)(count + i);
            atomicAdd(&sh.cls[cls(c.x)], 2u); atomicAdd(&sh.cls[cls(c.y)], 2u); atomicAdd(&sh.cls[cls(c.z)], 2u); atomicAdd(&sh.cls[cls(c.w)], 2u);
        }
    } else {
        for (int i = lo; i < hi; ++i) atomicAdd(&sh.cls[cls(scan_load<ATOMIC>(&count[i]))], 2u);
    }
    __syncthreads();
    uint32_t cnt[CPT], mine = 0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) { cnt[k] = sh.cls[tid * CPT + k]; mine += cnt[k]; }
    uint32_t num_items;
    uint32_t first = block_exclusive_scan<THREADS>(mine, sh.wave, num_items);   // exclusive start of this thread's classes
#pragma unroll
    for (int k = 0; k < CPT; ++k) { sh.cls[tid * CPT + k] = first; first += cnt[k]; }
    __syncthreads();
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < REGS; ++q)
            if (lo + q < hi) {
                const uint32_t at = atomicAdd(&sh.cls[cls(creg[q])], 2u);
                *(uint2 *)(order + at) = make_uint2((uint32_t)(lo + q), (uint32_t)(lo + q) | (1u << kItemHalfShift));
            }
    } else if (vec4) {
        for (int i = lo; i < hi; i += 4) {
            const uint4 c = *(const uint4 *)(count + i);
            const uint32_t cc[4] = {c.x, c.y, c.z, c.w};
            uint32_t at[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) at[k] = atomicAdd(&sh.cls[cls(cc[k])], 2u);
#pragma unroll
            for (int k = 0; k < 4; ++k) *(uint2 *)(order + at[k]) = make_uint2((uint32_t)(i + k), (uint32_t)(i + k) | (1u << kItemHalfShift));
        }
    } else {
        for (int i = lo; i < hi; ++i) {
            const uint32_t at = atomicAdd(&sh.cls[cls(scan_load<ATOMIC>(&count[i]))], 2u);
            *(uint2 *)(order + at) = make_uint2((uint32_t)i, (uint32_t)i | (1u << kItemHalfShift));
        }
    }
}

}  // namespace lsr
