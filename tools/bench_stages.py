#!/usr/bin/env python
"""Quick per-stage kernel times (hipEvents inside the library) for kernel A/B experiments.
usage: python tools/bench_stages.py [tag]     prints one line per workload:
  raster16 : 16 views x 300k Gaussians, 4-ch features (configs[1]/[2]), forward + backward
  decoder4 : DecoderSplattingCUDA at configs[3] (1 x 4 views, 393 216 Gaussians, colour SH 4 + latent SH 2)
Environment knobs of the library (LSR_SPLIT, LSR_LIMIT, ...) are read once per process."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from latentsplat_amd import _lib  # noqa: E402
from latentsplat_amd.rasterizer import rasterize_views  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    V, G = 16, 300_000
    for a in sys.argv:
        if a.startswith("--views="):
            V = int(a.split("=")[1])
        if a.startswith("--gaussians="):
            G = int(a.split("=")[1])
    dev = torch.device("cuda", 0)
    _lib.load()
    out = {"tag": tag, "views": V, "gaussians": G}
    inp = bench.build_inputs(G, V, 256, dev, 1234)
    gf = torch.randn((V, 4, 256, 256), device=dev)

    def step():
        m, c, o, f = (t.detach().requires_grad_(True) for t in (inp["means"], inp["cov"], inp["opac"], inp["features"]))
        res = rasterize_views(inp["views"], 256, 256, 0, m, c, o, features=f)
        res[1].backward(gf)

    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < (0.5 if "--warm" in sys.argv else 0.0):
        step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.profile_read(); _lib.profile_enable(True)
    for _ in range(40 if "--warm" in sys.argv else 10):
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    out["raster16"] = {k: round(ms / n, 4) for k, (ms, n) in _lib.profile_read().items() if n}
    del inp
    if "--no-decoder" not in sys.argv:
        from latentsplat_amd import decoder as dec
        from latentsplat_amd.synthetic import make_scene
        sc = make_scene(393_216, image_size=256, views=4, color_sh_degree=4, feature_channels=4, feature_sh_degree=2, seed=4321).to(dev)
        leaf = lambda t: t[None].contiguous().requires_grad_(True)
        gauss = dec.Gaussians(leaf(sc.means), leaf(sc.covariances), leaf(sc.opacities), leaf(sc.color_sh), leaf(sc.feature_sh))
        d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), [0.0, 0.0, 0.0]).to(dev)
        args = (gauss, sc.extrinsics[None], sc.intrinsics[None], sc.near[None], sc.far[None], (256, 256))
        gc, gl = torch.randn((1, 4, 3, 256, 256), device=dev), torch.randn((1, 4, 4, 256, 256), device=dev)

        def dstep():
            o = d.forward(*args)
            torch.autograd.backward([o.color, o.feature_posterior.mean], [gc, gl])

        for _ in range(3):
            dstep()
        torch.cuda.synchronize()
        _lib.profile_read(); _lib.profile_enable(True)
        for _ in range(10):
            dstep()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        out["decoder4"] = {k: round(ms / n, 4) for k, (ms, n) in _lib.profile_read().items() if n}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
