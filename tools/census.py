#!/usr/bin/env python
"""CPU census of the compositing work at the bench scene (one view): how many lock-step wave
iterations different work decompositions need per (Gaussian, tile) pair.  Development aid for
DESIGN.md §4 (uses the CPU oracle for the canonical tile lists; never part of the product path).

    python tools/census.py [--gaussians 300000] [--size 256]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def span_mask4(lo, hi):
    c0 = np.maximum(0, np.ceil((lo - 3.0) * 0.25)).astype(np.int64)
    c1 = np.minimum(3, np.floor(hi * 0.25)).astype(np.int64)
    ok = c0 <= c1
    m = ((2 << np.clip(c1, 0, 3)) - (1 << np.clip(c0, 0, 3))) & 0xF
    return np.where(ok, m, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=300_000)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    from tests import util
    from latentsplat_amd.synthetic import make_scene
    S = args.size
    sc = make_scene(args.gaussians, image_size=S, views=1, color_sh_degree=None, feature_channels=4, seed=1234)
    bi = util.boundary_inputs(sc, S, S)
    o = util.oracle_forward(bi, 0)
    xy, co = o["xy"].astype(np.float64), o["conic_opacity"].astype(np.float64)
    plist, ranges = o["point_list"].astype(np.int64), o["ranges"].astype(np.int64)
    gx = (S + 15) // 16
    T = ranges.shape[0]
    P = len(plist)
    tile_of = np.zeros(P, np.int64)
    for t in range(T):
        tile_of[ranges[t, 0]:ranges[t, 1]] = t
    g = plist
    A, B, Cc, op = co[g, 0], co[g, 1], co[g, 2], co[g, 3]
    x, y = xy[g, 0], xy[g, 1]
    tx0, ty0 = (tile_of % gx) * 16.0, (tile_of // gx) * 16.0
    det = A * Cc - B * B
    tau = np.log(np.maximum(255.0 * op, 1e-30)) * 1.0001 + 1e-4
    s = 2.0 * np.maximum(tau, 0) / det
    ex = np.sqrt(s * Cc) * 1.001 + 0.05
    ey = np.sqrt(s * A) * 1.001 + 0.05
    x0, x1, y0, y1 = x - ex - tx0, x + ex - tx0, y - ey - ty0, y + ey - ty0
    live = (op >= 1 / 255.0) & (x0 <= 15) & (x1 >= 0) & (y0 <= 15) & (y1 >= 0)
    cm = span_mask4(np.clip(x0, -8, 24), np.clip(x1, -8, 24))
    rm = span_mask4(np.clip(y0, -8, 24), np.clip(y1, -8, 24))
    m16 = np.zeros(P, np.int64)
    for r in range(4):
        m16 |= np.where((rm >> r) & 1, cm << (4 * r), 0)
    m16 = np.where(live, m16, 0)
    dead = (m16 == 0)
    print(f"pairs {P}, dead {dead.mean():.3f}, sub-blocks per pair {np.mean([bin(v).count('1') for v in m16[:50000]]):.2f}")
    qsb = [0x33 << (8 * (q >> 1) + 2 * (q & 1)) for q in range(4)]
    qhit = np.stack([(m16 & qm) != 0 for qm in qsb], 1)           # (P,4)
    print(f"quadrants per pair {qhit.sum(1).mean():.3f}; per live pair {qhit.sum(1)[~dead].mean():.3f}")
    sb_of_q = [[q0, q0 + 1, q0 + 4, q0 + 5] for q0 in (0, 2, 8, 10)]
    Bn = args.batch
    # (a) current: batches of Bn canonical entries, 4 rounds, iterations = max over the round's 4 sub-block lists
    it_cur = 0
    # (b) quadrant lists (dense, dead and foreign entries removed), batches of Bn entries of THAT list
    it_quad = 0
    batches_cur = batches_quad = 0
    quad_len = np.zeros((T, 4), np.int64)
    quad_iters = np.zeros((T, 4), np.int64)
    tile_iters = np.zeros(T, np.int64)
    for t in range(T):
        s0, s1 = ranges[t]
        m = m16[s0:s1]
        n = len(m)
        for b0 in range(0, n, Bn):
            mb = m[b0:b0 + Bn]
            batches_cur += 1
            for q in range(4):
                c = max(int(((mb >> sb) & 1).sum()) for sb in sb_of_q[q])
                it_cur += c
                tile_iters[t] += c
        for q in range(4):
            mq = m[(m & qsb[q]) != 0]
            quad_len[t, q] = len(mq)
            for b0 in range(0, len(mq), Bn):
                mb = mq[b0:b0 + Bn]
                batches_quad += 1
                c = max(int(((mb >> sb) & 1).sum()) for sb in sb_of_q[q])
                it_quad += c
                quad_iters[t, q] += c
    # (c) half-tile lists (rows 0-7 / 8-15), 8 sub-block lists per wave iteration (two pixels per lane, 8-lane groups)
    it_half = batches_half = ent_half = 0
    half_iters = np.zeros((T, 2), np.int64)
    for t in range(T):
        s0, s1 = ranges[t]
        m = m16[s0:s1]
        for h in range(2):
            mh = (m >> (8 * h)) & 0xFF
            mh = mh[mh != 0]
            ent_half += len(mh)
            for b0 in range(0, len(mh), Bn):
                mb = mh[b0:b0 + Bn]
                batches_half += 1
                c = max(int(((mb >> sb) & 1).sum()) for sb in range(8))
                it_half += c
                half_iters[t, h] += c
    print(f"half-tile lists: {it_half / P:.3f} 8-wide iterations per pair ({batches_half} batches, {ent_half / P:.3f} entries per pair); "
          f"ideal {sum(bin(v).count('1') for v in m16) / 8.0 / P:.3f}")
    # (d) how much of the gap to the ideal is the batch boundary?  Same lists with other batch sizes, with ONE batch of
    # look-ahead (a lane group that has finished its part of batch b continues in batch b+1 while the slowest group
    # finishes b), and the bound no batching scheme can beat (the longest of the half's eight sub-block lists).
    def half_lists():
        for t in range(T):
            s0, s1 = ranges[t]
            m = m16[s0:s1]
            for h in range(2):
                mh = (m >> (8 * h)) & 0xFF
                yield mh[mh != 0]
    per_batch = {bn: 0 for bn in (32, 64, 128, 256)}
    look = 0
    bound = 0
    for mh in half_lists():
        if not len(mh):
            continue
        hits = np.stack([(mh >> sb) & 1 for sb in range(8)], 1)            # (entries, 8)
        bound += int(hits.sum(0).max())
        for bn in per_batch:
            for b0 in range(0, len(mh), bn):
                per_batch[bn] += int(hits[b0:b0 + bn].sum(0).max())
        # one batch of look-ahead, batches of 64: backlog[g] = entries of already staged batches group g still has to do
        backlog = np.zeros(8, np.int64)
        nb = (len(mh) + Bn - 1) // Bn
        for bi_ in range(nb + 1):
            new = hits[bi_ * Bn:(bi_ + 1) * Bn].sum(0) if bi_ < nb else np.zeros(8, np.int64)
            # the round that retires batch bi_-1 runs until every group has finished ITS share of it; groups that are
            # done early work on batch bi_ (already staged) meanwhile
            if bi_ == 0:
                prev = new.copy(); backlog = np.zeros(8, np.int64); ahead = np.zeros(8, np.int64)
                continue
            rounds = int((prev - ahead).max()) if (prev - ahead).max() > 0 else 0
            look += rounds
            spare = rounds - (prev - ahead)                                  # iterations each group had left over
            ahead = np.minimum(np.maximum(spare, 0), new)                    # spent on the next batch
            prev = new
    print("half-tile lists, lock-step iterations per pair by staging scheme: " +
          ", ".join(f"batch {bn}: {v / P:.3f}" for bn, v in per_batch.items()) +
          f"; batch {Bn} with one batch of look-ahead: {look / P:.3f}; longest sub-block list (bound): {bound / P:.3f}")
    ideal = sum(bin(v).count("1") for v in m16) / 4.0
    print(f"lock-step iterations per pair: current {it_cur / P:.3f} ({batches_cur} batches), "
          f"quadrant lists {it_quad / P:.3f} ({batches_quad} batches, {quad_len.sum() / P:.3f} entries per pair), ideal {ideal / P:.3f}")
    # load balance (LPT on S slots, cost = iterations + per-entry staging weight)
    def lpt(costs, slots):
        import heapq
        h = [0.0] * slots
        heapq.heapify(h)
        for c in sorted(costs, reverse=True):
            heapq.heappush(h, heapq.heappop(h) + c)
        return max(h), sum(costs) / slots
    for views in (4, 16):
        for name, costs, slots in (("tiles/4096", np.tile(tile_iters, views), 4096), ("quadrants/4096", np.tile(quad_iters.reshape(-1), views), 4096),
                                   ("quadrants/8192", np.tile(quad_iters.reshape(-1), views), 8192),
                                   ("halves/4096", np.tile(half_iters.reshape(-1), views), 4096), ("halves/6144", np.tile(half_iters.reshape(-1), views), 6144)):
            mx, mean = lpt(list(costs.astype(float)), slots)
            print(f"views {views:2d} {name:15s}: items {len(costs)}, makespan/mean = {mx / mean:.3f} (max item {costs.max()}, mean load {mean:.0f})")


if __name__ == "__main__":
    main()
