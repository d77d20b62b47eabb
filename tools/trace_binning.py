#!/usr/bin/env python
"""Phase-stamp traces of k_scatter and the first-tier k_sort_tiles (debug build, tools/build_trace_lib.sh: thread 0 of every
workgroup writes wall_clock64() at the phase boundaries):
    LSR_LIB=build_variants/liblsr_trace.so LSR_TRACE_SCATTER=gpurun_out/tr_scatter.bin LSR_TRACE_SORT=gpurun_out/tr_sort.bin \\
        python tools/trace_binning.py [views]
    python tools/trace_binning.py --analyze gpurun_out/tr_scatter.bin gpurun_out/tr_sort.bin
Prints, per kernel, the launch span, how many workgroups were in flight, and the mean / p90 duration of every phase."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCATTER = ["load issue + clear", "count", "reserve", "place + store issue", "drain"]
SORT = ["load + range", "histogram", "bucket scan", "place", "rank + sorted words", "point_list + half lists"]


def analyze(path, phases, tick_ns):
    raw = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
    raw = raw[raw[:, 0] > 0]
    if not len(raw):
        print(path, ": empty")
        return
    t = raw[:, :len(phases) + 1].astype(np.int64)
    done = t[:, -1] > 0
    short = (~done).sum()
    t = t[done]
    t0 = t[:, 0].min()
    span = (t[:, -1].max() - t0) * tick_ns / 1e3
    dur = (t[:, -1] - t[:, 0]) * tick_ns / 1e3
    print(f"{os.path.basename(path)}: {len(raw)} workgroups ({short} left early: empty / long lists), launch span {span:.1f} us, "
          f"workgroup duration mean {dur.mean():.2f} p50 {np.median(dur):.2f} p90 {np.percentile(dur, 90):.2f} max {dur.max():.2f} us")
    for i, name in enumerate(phases):
        d = (t[:, i + 1] - t[:, i]) * tick_ns / 1e3
        print(f"   {name:26s} mean {d.mean():6.2f}  p50 {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
    # occupancy over time: workgroups in flight in 20 slices
    edges = np.linspace(t0, t[:, -1].max(), 21)
    act = [int(((t[:, 0] < e1) & (t[:, -1] > e0)).sum()) for e0, e1 in zip(edges[:-1], edges[1:])]
    print("   workgroups in flight per 5 % slice:", act)
    start = (t[:, 0] - t0) * tick_ns / 1e3
    print(f"   start times: p10 {np.percentile(start, 10):.1f} p50 {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} us")


def main():
    if sys.argv[1:2] == ["--analyze"]:
        tick = float(os.environ.get("LSR_TICK_NS", "10"))
        analyze(sys.argv[2], SCATTER, tick)
        analyze(sys.argv[3], SORT, tick)
        return
    import torch
    import bench
    from latentsplat_amd.rasterizer import rasterize_views
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda", 0)
    inp = bench.build_inputs(300_000, V, 256, dev, 1234)
    with torch.no_grad():
        for _ in range(3):
            rasterize_views(inp["views"], 256, 256, 0, inp["means"], inp["cov"], inp["opac"], features=inp["features"])
    torch.cuda.synchronize()
    tick = 10.0   # wall_clock64 runs at 100 MHz on MI300-class parts (hipDeviceAttributeWallClockRate = 100000 kHz)
    analyze(os.environ["LSR_TRACE_SCATTER"], SCATTER, tick)
    analyze(os.environ["LSR_TRACE_SORT"], SORT, tick)


if __name__ == "__main__":
    main()
