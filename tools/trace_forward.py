#!/usr/bin/env python
"""Per-item cycle trace of the forward compositing kernel (debug build, tools/build_trace_lib.sh):
    LSR_LIB=build_variants/liblsr_trace.so LSR_TRACE=gpurun_out/trace_fwd.bin python tools/trace_forward.py [views]
prints how the launch's time is spent: item durations, wave lifetimes per SIMD, cycles per lock-step iteration
as a function of how many waves share the SIMD."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from latentsplat_amd.rasterizer import rasterize_views  # noqa: E402


def main():
    if sys.argv[1:2] == ["--analyze"]:      # offline: analyse an existing dump
        path = sys.argv[2]
    else:
        V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
        path = os.environ["LSR_TRACE"]
        dev = torch.device("cuda", 0)
        inp = bench.build_inputs(300_000, V, 256, dev, 1234)
        with torch.no_grad():
            for _ in range(3):
                rasterize_views(inp["views"], 256, 256, 0, inp["means"], inp["cov"], inp["opac"], features=inp["features"])
        torch.cuda.synchronize()
    raw = np.fromfile(path, dtype=np.uint64).reshape(-1, 6)
    raw = raw[raw[:, 1] > 0]
    # (round 6) [4] = first batch staged, [5] = batch loop left: the item's prologue (metadata + the two dependent loads of
    # its first batch) and epilogue (per-pixel stores)
    pro = (raw[:, 4].astype(np.int64) - raw[:, 0].astype(np.int64))[raw[:, 4] > 0]
    epi = (raw[:, 1].astype(np.int64) - raw[:, 5].astype(np.int64))[raw[:, 5] > 0]
    if len(pro):
        print(f"prologue (item picked -> first batch staged): mean {pro.mean():.0f} p50 {np.median(pro):.0f} p90 {np.percentile(pro, 90):.0f} cycles; "
              f"epilogue (loop left -> stores issued): mean {epi.mean():.0f} p50 {np.median(epi):.0f} p90 {np.percentile(epi, 90):.0f}")
    t0, t1 = raw[:, 0].astype(np.int64), raw[:, 1].astype(np.int64)
    hw = raw[:, 2]
    iters, ents = (raw[:, 3] >> np.uint64(32)).astype(np.int64), (raw[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    # every CU has its own cycle counter: times are relative to the CU's first item
    cu = ((hw >> np.uint64(32)).astype(np.int64) << 8) | ((hw >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64)
    for x in np.unique(cu):
        sel = cu == x
        b = t0[sel].min()
        t0[sel] -= b; t1[sel] -= b
    span = t1.max()
    dur = t1 - t0
    simd = (hw >> np.uint64(4)) & np.uint64(0xFFFFFFFFFFF)      # everything above the wave slot: simd, pipe, cu, sh, se, xcc
    cu_end = np.array([t1[cu == x].max() for x in np.unique(cu)])
    print(f"CU end times: min {cu_end.min()} mean {cu_end.mean():.0f} max {cu_end.max()}")
    wave = hw                                                    # wave slot identity
    print(f"items {len(raw)}, launch span {span} cycles; item duration mean {dur.mean():.0f} p50 {np.median(dur):.0f} p90 {np.percentile(dur, 90):.0f} max {dur.max()}")
    print(f"entries per item mean {ents.mean():.0f} max {ents.max()}; iterations per item mean {iters.mean():.0f} max {iters.max()}; total iterations {iters.sum()}")
    # per wave slot: busy time, end time
    us, inv = np.unique(wave, return_inverse=True)
    busy = np.bincount(inv, weights=dur)
    endt = np.zeros(len(us)); np.maximum.at(endt, inv, t1)
    nitem = np.bincount(inv)
    print(f"wave slots used {len(us)}; items per slot mean {nitem.mean():.2f} max {nitem.max()}; slot end time mean {endt.mean():.0f} ({endt.mean() / span:.2f} of span) p10 {np.percentile(endt, 10):.0f} p90 {np.percentile(endt, 90):.0f}")
    print(f"slot busy/end mean {np.mean(busy / np.maximum(endt, 1)):.3f}")
    ss, sinv = np.unique(simd, return_inverse=True)
    s_end = np.zeros(len(ss)); np.maximum.at(s_end, sinv, t1)
    s_iters = np.bincount(sinv, weights=iters)
    print(f"SIMDs {len(ss)}: iterations per SIMD mean {s_iters.mean():.0f} min {s_iters.min():.0f} max {s_iters.max():.0f}; SIMD end mean {s_end.mean():.0f} ({s_end.mean() / span:.2f} of span) min {s_end.min():.0f} max {s_end.max():.0f}")
    print(f"cycles per iteration per SIMD over its own end time: mean {(s_end / np.maximum(s_iters, 1)).mean():.1f}; over the span: {span / s_iters.mean():.1f}")
    # correlation of item duration with its iterations
    c = np.corrcoef(dur, iters)[0, 1]
    print(f"corr(duration, iterations) {c:.2f}; cycles per iteration per item (duration / iterations): p10 {np.percentile(dur / np.maximum(iters, 1), 10):.0f} p50 {np.percentile(dur / np.maximum(iters, 1), 50):.0f} p90 {np.percentile(dur / np.maximum(iters, 1), 90):.0f}")
    # timeline: number of active items in 20 slices
    edges = np.linspace(0, span, 21)
    act = [(int(((t0 < e1) & (t1 > e0)).sum())) for e0, e1 in zip(edges[:-1], edges[1:])]
    print("active items per 5 % slice of the span:", act)
    order = np.argsort(t0)
    first = (t0 < 0.02 * span)
    print(f"items started in the first 2 % of the span: {int(first.sum())}; of those, duration mean {dur[first].mean():.0f}, iterations mean {iters[first].mean():.0f}")
    late = ~first
    if late.any():
        print(f"queue items: {int(late.sum())}, duration mean {dur[late].mean():.0f}, iterations mean {iters[late].mean():.0f}")


if __name__ == "__main__":
    main()
