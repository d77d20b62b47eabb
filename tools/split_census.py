#!/usr/bin/env python
"""Census for VERDICT r5 item 2 (splitting the costliest half-tile items of the compositing launches along their lists).

A list segment [m, n) composites linearly in the transmittance T_m in front of it, so a second wave could start it from
T' = 1 and the halves be combined afterwards — EXCEPT for pixels whose `T (1 - alpha) < 1e-4` stop falls inside [m, n):
the second wave cannot know where they stop (it does not have T_m), over-blends, and the pixel has to be redone from the
first half's state.  This tool counts, on the bench scene (16 views x 300 k Gaussians, 256 x 256) and through the product's
own forward (n_contrib = the per-pixel list prefix the backward walks; half_count = the list length), how often that is:

  * per half-tile item: list length hn, walked length (largest n_contrib of its pixels), pixels that stop at all, pixels
    whose stop falls in the second half of the WALKED list;
  * for the K costliest items (the ones a splitter would split): the fraction with any / with > 5 % / > 25 % such pixels.

    python tools/split_census.py [--views 16] [--gaussians 300000] [--out gpurun_out/split_census.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--gaussians", type=int, default=300_000)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from latentsplat_amd.synthetic import make_scene
    from tests import util
    dev = torch.device("cuda", 0)
    V, S = a.views, a.size
    sc = make_scene(a.gaussians, image_size=S, views=V, color_sh_degree=None, feature_channels=4, feature_sh_degree=0, seed=a.seed)
    bi = util.boundary_inputs(sc, S, S)
    run = util.HipRun(bi, dev, shared_means=True)
    nc, hc = run.n_contrib(), run.half_count()          # (V, H, W), (V*T, 2)
    T, gx = run.T, (S + 15) // 16
    items = []
    for v in range(V):
        for t in range(T):
            ty, tx = divmod(t, gx)
            for h in range(2):
                hn = int(hc[v * T + t, h])
                blk = nc[v, ty * 16 + 8 * h: ty * 16 + 8 * h + 8, tx * 16: tx * 16 + 16].reshape(-1)
                walked = int(blk.max()) if blk.size else 0
                stopped = blk < hn                       # the pixel ran out of transmittance at entry blk + 1
                # the stop lies in the second half of what the item walks (a splitter would cut at walked / 2)
                late = stopped & (blk >= walked // 2)
                items.append((hn, walked, int(stopped.sum()), int(late.sum()), int(blk.size)))
    it = np.array(items, np.int64)
    order = np.argsort(-it[:, 1], kind="stable")        # costliest first by WALKED length
    res = dict(views=V, gaussians=a.gaussians, size=S, items=int(len(it)),
               list_len=dict(mean=float(it[:, 0].mean()), max=int(it[:, 0].max())),
               walked_len=dict(mean=float(it[:, 1].mean()), max=int(it[:, 1].max()), p50=float(np.median(it[:, 1])), p90=float(np.percentile(it[:, 1], 90))),
               walked_over_list=float(it[:, 1].sum() / max(1, it[:, 0].sum())),
               pixels_that_stop=float(it[:, 2].sum() / it[:, 4].sum()),
               items_where_every_pixel_stops=float((it[:, 2] == it[:, 4]).mean()),
               top={})
    for K in (512, 1024, 2048, 4096):
        top = it[order[:K]]
        frac_late = top[:, 3] / np.maximum(1, top[:, 4])
        res["top"][str(K)] = dict(walked_mean=float(top[:, 1].mean()), walked_min=int(top[:, 1].min()),
                                  any_late_stop=float((top[:, 3] > 0).mean()), over_5pct_late=float((frac_late > 0.05).mean()),
                                  over_25pct_late=float((frac_late > 0.25).mean()), mean_late_pixel_frac=float(frac_late.mean()),
                                  mean_stopped_pixel_frac=float((top[:, 2] / np.maximum(1, top[:, 4])).mean()))
    print(json.dumps(res, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
        np.save(os.path.splitext(a.out)[0] + "_items.npy", it.astype(np.int32))


if __name__ == "__main__":
    main()
