#!/usr/bin/env python
"""Times lsr_forward_prepare ALONE (projection kernel + tile scan; nothing downstream runs) for development-knob sets in
one process — used for phase ablations of the single-pass binning whose results are deliberately wrong
(LSR_SEG_ABLATE) and must therefore never reach the sort / compositing kernels.

    python tools/ab_prepare.py [--views 16] [--gaussians 300000] '{"LSR_SEGMENTS":0}' '{"LSR_SEG_ABLATE":1}' ..."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from latentsplat_amd import _lib  # noqa: E402
from latentsplat_amd._lib import Dims, Inputs  # noqa: E402

DEFAULTS = {"LSR_SEGMENTS": 1, "LSR_SEG_ABLATE": 0, "LSR_PRE_VB": 0, "LSR_PRE_ITEMS": 8, "LSR_FOLD_SCAN": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--gaussians", type=int, default=300_000)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("sets", nargs="*")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    V, G, S = args.views, args.gaussians, 256
    inp = bench.build_inputs(G, V, S, dev, 1234)
    d = Dims(V, G, S, S, 4, 0, 0, 0, 0, 0, 0, 0, 0, 9, 0, 0, 0, 0, 0, 0)
    p = lambda x: C.c_void_p(x.data_ptr())
    ci = Inputs(p(inp["views"]), p(inp["means"]), p(inp["cov"]), p(inp["opac"]), None, p(inp["features"]))
    radii = torch.zeros((V, G), dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    sets = [{}] + [json.loads(s) for s in args.sets]
    names = sorted({k for s in sets for k in s})
    for rnd in range(args.rounds):
        for s in sets:
            for n in names:
                _lib.set_knob(n, DEFAULTS.get(n, 0))
            for k, v in s.items():
                _lib.set_knob(k, int(v))
            geom = torch.zeros(lib.lsr_geom_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
            npairs, maxtile = C.c_int64(0), C.c_int32(0)

            def call():
                _lib.check(lib.lsr_forward_prepare(C.byref(d), C.byref(ci), p(geom), p(radii), C.byref(npairs), C.byref(maxtile), stream), "prepare")
            for _ in range(20):
                call()
            torch.cuda.synchronize(dev)
            _lib.profile_read(); _lib.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                call()
            torch.cuda.synchronize(dev)
            wall = 1e3 * (time.perf_counter() - t0) / args.steps
            _lib.profile_enable(False)
            prof = {k: round(ms / n, 4) for k, (ms, n) in _lib.profile_read().items() if n}
            print(json.dumps(dict(round=rnd, knobs=s, wall_ms=round(wall, 4), kernels=prof, pairs=npairs.value, longest=maxtile.value)), flush=True)
            del geom


if __name__ == "__main__":
    main()
