#!/usr/bin/env python
"""A/B of the library's development knobs in ONE process (lsr_debug_set_knob): per-kernel times (hipEvents inside the
library) and wall-clock step times of the bench workloads for each knob set, alternating, several rounds.

    python tools/ab_knobs.py [--rounds 2] [--workloads raster16,cfg3,cfg4] '{"LSR_FOLD_SCAN":0}' '{"LSR_FUSE_SH":0}' ...

The first (implicit) set is the default configuration {}.  A knob set stays in force until the next one resets it: every
knob named anywhere on the command line is reset to its default (given as NAME=default in --defaults, else 0) first."""
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
from latentsplat_amd.rasterizer import rasterize_views  # noqa: E402

DEFAULTS = {"LSR_REORDER": 1, "LSR_CLEAR_BESIDE": -1, "LSR_FWD_BINQ": -1, "LSR_BWD_BINQ": 0, "LSR_BWD_REV": 2, "LSR_BWD_PARTS": -1, "LSR_BWD_PRIO_PCT": -1, "LSR_FWD_PRIO_PCT": -1, "LSR_FWD_RECORD": 1, "LSR_SEGMENTS": 1, "LSR_PRE_ITEMS": 8, "LSR_FOLD_SCAN": 1, "LSR_HOST_POLL": 1, "LSR_SORT_LPT": 1, "LSR_FUSE_SH": 1, "LSR_FWD_VARIANT": 0, "LSR_BWD_VARIANT": 0, "LSR_FWD_ROWS": -1, "LSR_FWD_ROWREC": 1, "LSR_FWD_QUAD": -1, "LSR_FWD_LIVE": 1}


def timed(fn, steps, dev):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return 1e3 * (time.perf_counter() - t0) / steps


def kernels(fn, steps, dev):
    _lib.profile_read(); _lib.profile_enable(True)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    _lib.profile_enable(False)
    return {k: round(ms / n, 4) for k, (ms, n) in _lib.profile_read().items() if n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--workloads", default="raster16,cfg3,cfg4")
    ap.add_argument("--gaussians", type=int, default=300_000, help="scene size of the raster16 workload")
    ap.add_argument("--views", type=int, default=16, help="views of the raster16 workload")
    ap.add_argument("--opacity-scale", type=float, default=1.0 / 3.0, help="raster16: opacities U(0, scale) (1: pixels run out of transmittance)")
    ap.add_argument("--sigma", type=float, nargs=2, default=(0.3, 3.0), help="raster16: projected sigma range in pixels")
    ap.add_argument("--encoder-shaped", action="store_true", help="cfg3 / cfg4: pixel-aligned scenes in ray order (synthetic.make_encoder_scene)")
    ap.add_argument("sets", nargs="*")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    _lib.load()
    sets = [{}] + [json.loads(s) for s in args.sets]
    names = sorted({k for s in sets for k in s})
    wl = {}
    want = args.workloads.split(",")
    if "raster16" in want:
        inp = bench.build_inputs(args.gaussians, args.views, 256, dev, 1234, opacity_scale=args.opacity_scale, sigma_px=tuple(args.sigma))
        gf = torch.randn((args.views, 4, 256, 256), device=dev)

        def r_fwd():
            with torch.no_grad():
                rasterize_views(inp["views"], 256, 256, 0, inp["means"], inp["cov"], inp["opac"], features=inp["features"])

        def r_fb():
            m, c, o, f = (t.detach().requires_grad_(True) for t in (inp["means"], inp["cov"], inp["opac"], inp["features"]))
            rasterize_views(inp["views"], 256, 256, 0, m, c, o, features=f)[1].backward(gf)
        wl["raster16"] = (r_fwd, r_fb)
    if "nosync16" in want:      # the no-sync forward (pair capacity 1.5 x the measured count), back to back on one stream
        from latentsplat_amd.rasterizer import last_forward_status
        inp2 = bench.build_inputs(args.gaussians, 16, 256, dev, 1234)
        with torch.no_grad():
            rasterize_views(inp2["views"], 256, 256, 0, inp2["means"], inp2["cov"], inp2["opac"], features=inp2["features"])
        st = last_forward_status()
        kw = dict(pair_capacity=int(1.5 * st["num_pairs"]), max_tile_hint=int(st["max_tile_pairs"]))

        def n_fwd():
            with torch.no_grad():
                rasterize_views(inp2["views"], 256, 256, 0, inp2["means"], inp2["cov"], inp2["opac"], features=inp2["features"], **kw)
        wl["nosync16"] = (n_fwd, n_fwd)
    for name, scenes in (("cfg3", 1), ("cfg4", 4)):
        if name not in want:
            continue
        from latentsplat_amd import decoder as dec
        from latentsplat_amd.synthetic import make_encoder_scene, make_scene
        if args.encoder_shaped:
            scs = [make_encoder_scene(seed=4321 + i).to(dev) for i in range(scenes)]
        else:
            scs = [make_scene(393_216, image_size=256, views=4, color_sh_degree=4, feature_channels=4, feature_sh_degree=2,
                              seed=4321 + i).to(dev) for i in range(scenes)]
        st = lambda n: torch.stack([getattr(sc, n) for sc in scs])
        leaf = lambda n: st(n).contiguous().requires_grad_(True)
        gauss = dec.Gaussians(leaf("means"), leaf("covariances"), leaf("opacities"), leaf("color_sh"), leaf("feature_sh"))
        d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), [0.0, 0.0, 0.0]).to(dev)
        a = (gauss, st("extrinsics"), st("intrinsics"), st("near"), st("far"), (256, 256))
        gc = torch.randn((scenes, 4, 3, 256, 256), device=dev)
        gl = torch.randn((scenes, 4, 4, 256, 256), device=dev)
        leaves = (gauss.means, gauss.covariances, gauss.opacities, gauss.color_harmonics, gauss.feature_harmonics)

        def d_fwd(d=d, a=a):
            with torch.no_grad():
                d.forward(*a)

        def d_fb(d=d, a=a, gc=gc, gl=gl, leaves=leaves):
            o = d.forward(*a)
            torch.autograd.backward([o.color, o.feature_posterior.mean], [gc, gl])
            for t in leaves:
                t.grad = None
        wl[name] = (d_fwd, d_fb)
    for rnd in range(args.rounds):
        for s in sets:
            for n in names:
                _lib.set_knob(n, DEFAULTS.get(n, 0))
            for k, v in s.items():
                _lib.set_knob(k, int(v))
            for name, (f, fb) in wl.items():
                res = {"round": rnd, "knobs": s, "workload": name, "fwd_ms": round(timed(f, args.steps, dev), 4),
                       "fwdbwd_ms": round(timed(fb, args.steps, dev), 4)}
                res["kernels_fwd"] = kernels(f, 10, dev)
                res["kernels_fwdbwd"] = kernels(fb, 10, dev)
                print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
