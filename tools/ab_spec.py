#!/usr/bin/env python
"""Forward step time of the bench workload (16 views x 300 k) under the three host protocols, alternating in one process:
exact (prepare + render: the host reads the pair count between the two halves), speculative (everything launched at once
with a workspace sized from earlier calls, counts read afterwards) and no-sync (counts never read).
    python tools/ab_spec.py [--rounds 5] [--steps 200] [--views 16]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from latentsplat_amd import _lib  # noqa: E402
from latentsplat_amd import rasterizer as rz  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--gaussians", type=int, default=300_000)
    ap.add_argument("--bwd", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    _lib.load()
    V, S = args.views, 256
    inp = bench.build_inputs(args.gaussians, V, S, dev, 1234)
    gf = torch.randn((V, 4, S, S), device=dev)

    def call(**kw):
        if args.bwd:
            m, c, o, f = (t.detach().requires_grad_(True) for t in (inp["means"], inp["cov"], inp["opac"], inp["features"]))
            rz.rasterize_views(inp["views"], S, S, 0, m, c, o, features=f, **kw)[1].backward(gf)
        else:
            with torch.no_grad():
                rz.rasterize_views(inp["views"], S, S, 0, inp["means"], inp["cov"], inp["opac"], features=inp["features"], **kw)

    call()
    st = rz.last_forward_status()
    nosync_kw = dict(pair_capacity=int(1.25 * st["num_pairs"]) + 4096, max_tile_hint=4096)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        call()
    torch.cuda.synchronize(dev)
    res = {"exact": [], "speculative": [], "nosync": []}
    for rnd in range(args.rounds):
        for mode in res:
            rz._SPECULATE = mode == "speculative"
            kw = nosync_kw if mode == "nosync" else {}
            for _ in range(20):
                call(**kw)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                call(**kw)
            torch.cuda.synchronize(dev)
            res[mode].append(1e3 * (time.perf_counter() - t0) / args.steps)
    for mode, v in res.items():
        v = sorted(v)
        print(json.dumps(dict(mode=mode, bwd=args.bwd, min=round(v[0], 4), median=round(v[len(v) // 2], 4), max=round(v[-1], 4), stats=dict(rz.SPECULATION_STATS))))


if __name__ == "__main__":
    main()
