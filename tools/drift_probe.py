#!/usr/bin/env python
"""Does a long loop of configs[4] forward+backward steps slow down over time (power / clock management)?  Prints the mean
step time of consecutive blocks of 20 steps, the shader clock (tools/microbench/clock_probe.hip) and the per-kernel times
at the start and at the end; then the per-kernel times of the NO-SYNC forward of the bench workload next to the
synchronous one (the no-sync loop on one stream was measured slower than the synchronous loop)."""
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
from latentsplat_amd import decoder as dec  # noqa: E402
from latentsplat_amd.rasterizer import last_forward_status, rasterize_views  # noqa: E402
from latentsplat_amd.synthetic import make_scene  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    _lib.load()
    probe = C.CDLL(os.path.join(ROOT, "tools", "microbench", "libclock_probe.so"))
    probe.clock_probe_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    khz = probe.clock_probe_wall_khz()
    side = torch.cuda.Stream(dev)
    buf = torch.zeros(2 * 64, dtype=torch.int64, device=dev)
    scenes = 4
    scs = [make_scene(393_216, image_size=256, views=4, color_sh_degree=4, feature_channels=4, feature_sh_degree=2, seed=4321 + i).to(dev) for i in range(scenes)]
    st = lambda n: torch.stack([getattr(sc, n) for sc in scs])
    leaf = lambda n: st(n).contiguous().requires_grad_(True)
    gauss = dec.Gaussians(leaf("means"), leaf("covariances"), leaf("opacities"), leaf("color_sh"), leaf("feature_sh"))
    d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), [0.0, 0.0, 0.0]).to(dev)
    a = (gauss, st("extrinsics"), st("intrinsics"), st("near"), st("far"), (256, 256))
    gc = torch.randn((scenes, 4, 3, 256, 256), device=dev)
    gl = torch.randn((scenes, 4, 4, 256, 256), device=dev)
    leaves = (gauss.means, gauss.covariances, gauss.opacities, gauss.color_harmonics, gauss.feature_harmonics)

    def fb():
        o = d.forward(*a)
        torch.autograd.backward([o.color, o.feature_posterior.mean], [gc, gl])
        for t in leaves:
            t.grad = None

    for _ in range(5):
        fb()
    torch.cuda.synchronize(dev)
    blocks = []
    for b in range(12):
        probe.clock_probe_launch(C.c_void_p(buf.data_ptr()), b, 200, C.c_void_p(side.cuda_stream))
        t0 = time.perf_counter()
        for _ in range(20):
            fb()
        torch.cuda.synchronize(dev)
        blocks.append(round(1e3 * (time.perf_counter() - t0) / 20, 4))
    bb = buf.cpu().numpy().reshape(-1, 2)[:12]
    print(json.dumps({"cfg4_fwdbwd_ms_per_step_blocks_of_20": blocks,
                      "sclk_mhz_at_block_start": [round(float(c / w * khz / 1e3)) if w else None for c, w in bb],
                      "memory_allocated_gb": round(torch.cuda.memory_allocated(dev) / 1e9, 2), "reserved_gb": round(torch.cuda.memory_reserved(dev) / 1e9, 2)}), flush=True)
    _lib.profile_read(); _lib.profile_enable(True)
    for _ in range(10):
        fb()
    torch.cuda.synchronize(dev)
    _lib.profile_enable(False)
    print(json.dumps({"cfg4_kernel_ms_after_250_steps": {k: round(ms / n, 4) for k, (ms, n) in _lib.profile_read().items() if n}}), flush=True)
    del gauss, scs, a, gc, gl, leaves
    torch.cuda.empty_cache()
    # ---- synchronous vs no-sync forward of the bench workload, per kernel ----
    inp = bench.build_inputs(300_000, 16, 256, dev, 1234)
    with torch.no_grad():
        call = lambda **kw: rasterize_views(inp["views"], 256, 256, 0, inp["means"], inp["cov"], inp["opac"], features=inp["features"], **kw)
        call(); stt = last_forward_status()
        kw = dict(pair_capacity=int(1.5 * stt["num_pairs"]), max_tile_hint=int(stt["max_tile_pairs"]))
        for name, k in (("sync", {}), ("nosync", kw), ("sync", {}), ("nosync", kw)):
            for _ in range(100):
                call(**k)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(100):
                call(**k)
            torch.cuda.synchronize(dev)
            wall = 1e3 * (time.perf_counter() - t0) / 100
            _lib.profile_read(); _lib.profile_enable(True)
            for _ in range(20):
                call(**k)
            torch.cuda.synchronize(dev)
            _lib.profile_enable(False)
            print(json.dumps({"mode": name, "ms_per_step": round(wall, 4), "kernel_ms": {kk: round(ms / n, 4) for kk, (ms, n) in _lib.profile_read().items() if n}}), flush=True)


if __name__ == "__main__":
    main()
