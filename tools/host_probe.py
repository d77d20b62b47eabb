#!/usr/bin/env python
"""How much HOST time one forward + backward step costs (Python + ctypes + HIP launch calls), measured with the no-sync forward
(nothing in it waits for the device) and a device synchronisation only every `--every` steps: if the host time per step
exceeds the device time per step, the step is host-bound on this box.

    python tools/host_probe.py [--workload raster16|cfg4] [--steps 200]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from latentsplat_amd.rasterizer import last_forward_status, rasterize_views  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    inp = bench.build_inputs(300_000, 16, 256, dev, 1234)
    gf = torch.randn((16, 4, 256, 256), device=dev)
    with torch.no_grad():
        rasterize_views(inp["views"], 256, 256, 0, inp["means"], inp["cov"], inp["opac"], features=inp["features"])
    st = last_forward_status()
    kw = dict(pair_capacity=int(1.3 * st["num_pairs"]), max_tile_hint=4096)
    res = {}
    for name, kwargs in (("synchronous / speculative forward", {}), ("no-sync forward", kw)):
        t_f = t_b = 0.0

        def step():
            nonlocal t_f, t_b
            m, c, o, f = (t.detach().requires_grad_(True) for t in (inp["means"], inp["cov"], inp["opac"], inp["features"]))
            t0 = time.perf_counter()
            out = rasterize_views(inp["views"], 256, 256, 0, m, c, o, features=f, **kwargs)
            t1 = time.perf_counter()
            out[1].backward(gf)
            t2 = time.perf_counter()
            t_f += t1 - t0; t_b += t2 - t1
        for _ in range(20):
            step()
        torch.cuda.synchronize(dev)
        t_f = t_b = 0.0
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        t_all = time.perf_counter() - t0
        res[name] = dict(host_forward_ms=1e3 * t_f / a.steps, host_backward_ms=1e3 * t_b / a.steps, host_issue_ms_per_step=1e3 * t_issue / a.steps,
                         wall_ms_per_step=1e3 * t_all / a.steps)
        print(name, {k: round(v, 4) for k, v in res[name].items()}, flush=True)


if __name__ == "__main__":
    main()
