#!/usr/bin/env python
"""how much of the "bit-exact tile / sort indices" contract depends on whether the projection stage
contracts a*b+c into fused multiply-adds (nvcc's default -fmad=true, which the real fork was built with) or keeps every
float operation separate (the convention of oracle/raster_oracle.c, preprocess.hip and every index test).

Runs the CPU oracle in BOTH conventions (oracle.set_fma_contraction) over the configs[1] scene (300 000 Gaussians, 4-ch
features) and the configs[3] scene (393 216 Gaussians, colour SH 4 + latent SH 2), a few views each, and reports how
many radii, tile rectangles, depth bits, (Gaussian, tile) pairs and sorted-list positions differ, and what that does to
the rendered images.  CPU only (test infrastructure); writes profiles/r04_contraction_census.json.

    python tools/contraction_census.py [--views 4] [--out profiles/r04_contraction_census.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latentsplat_amd.synthetic import make_scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests import util  # noqa: E402


def compare(bi, v):
    orc.set_fma_contraction(False)
    a = util.oracle_forward(bi, v)
    orc.set_fma_contraction(True)
    b = util.oracle_forward(bi, v)
    orc.set_fma_contraction(False)
    G = a["radii"].shape[0]
    vis = (a["radii"] > 0) | (b["radii"] > 0)
    dbits_a, dbits_b = a["gdepth"].view(np.uint32), b["gdepth"].view(np.uint32)
    res = dict(gaussians=int(G), visible=int(vis.sum()),
               culled_differently=int(((a["radii"] > 0) != (b["radii"] > 0)).sum()),
               radii_differ=int((a["radii"] != b["radii"]).sum()),
               rects_differ=int((a["rect"] != b["rect"]).any(axis=1).sum()),
               depth_bits_differ=int((dbits_a != dbits_b)[vis].sum()),
               depth_max_ulp=int(np.abs(dbits_a.astype(np.int64) - dbits_b.astype(np.int64))[vis].max()),
               pixel_mean_bits_differ=int((a["xy"].view(np.uint32) != b["xy"].view(np.uint32)).any(axis=1)[vis].sum()),
               conic_bits_differ=int((a["conic_opacity"][:, :3].view(np.uint32) != b["conic_opacity"][:, :3].view(np.uint32)).any(axis=1)[vis].sum()),
               pairs=(int(len(a["point_list"])), int(len(b["point_list"]))))
    # sorted lists, tile by tile: positions that hold a different Gaussian; tiles whose list differs at all;
    # pure order swaps (same multiset, different order) vs membership changes
    ra, rb = a["ranges"].astype(np.int64), b["ranges"].astype(np.int64)
    pos_diff = tiles_diff = swaps_only = member = 0
    for t in range(ra.shape[0]):
        la, lb = a["point_list"][ra[t, 0]:ra[t, 1]], b["point_list"][rb[t, 0]:rb[t, 1]]
        if len(la) == len(lb) and np.array_equal(la, lb):
            continue
        tiles_diff += 1
        if len(la) == len(lb):
            pos_diff += int((la != lb).sum())
            if np.array_equal(np.sort(la), np.sort(lb)):
                swaps_only += 1
            else:
                member += 1
        else:
            member += 1
            n = min(len(la), len(lb))
            pos_diff += int((la[:n] != lb[:n]).sum()) + abs(len(la) - len(lb))
    res.update(list_positions_differ=pos_diff, tiles_with_a_different_list=tiles_diff, tiles_order_swaps_only=swaps_only,
               tiles_membership_changes=member, tiles=int(ra.shape[0]))
    img = {}
    for k in ("color", "feature", "mask", "depth"):
        if a.get(k) is not None and b.get(k) is not None and np.size(a[k]):
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            img[k] = dict(max_abs=float(d.max()), pixels_over_1e_4=int((d > 1e-4).sum()), elements=int(d.size))
    res["images_uncontracted_vs_contracted"] = img
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_contraction_census.json"))
    args = ap.parse_args()
    out = {"what": "oracle, uncontracted vs FMA-contracted projection (oracle_set_fma_contraction), per view",
           "rule": "LLVM default fadd/fsub-of-fmul combine on the published expression trees (glm mat3 products with their zero terms)"}
    scenes = {
        "configs[1]: 300000 Gaussians, 4-ch features, 256x256": dict(G=300_000, color_sh_degree=None, feature_sh_degree=0, seed=1234),
        "configs[3]: 393216 Gaussians, colour SH 4 + latent SH 2, 256x256": dict(G=393_216, color_sh_degree=4, feature_sh_degree=2, seed=4321),
    }
    for name, cfg in scenes.items():
        sc = make_scene(cfg["G"], image_size=256, views=args.views, color_sh_degree=cfg["color_sh_degree"], feature_channels=4,
                        feature_sh_degree=cfg["feature_sh_degree"], seed=cfg["seed"])
        bi = util.boundary_inputs(sc, 256, 256)
        per_view = [compare(bi, v) for v in range(args.views)]
        tot = {k: int(sum(r[k] for r in per_view)) for k in per_view[0] if isinstance(per_view[0][k], int)}
        tot["pairs"] = [int(sum(r["pairs"][0] for r in per_view)), int(sum(r["pairs"][1] for r in per_view))]
        tot["depth_max_ulp"] = int(max(r["depth_max_ulp"] for r in per_view))
        tot["image_max_abs"] = {k: max(r["images_uncontracted_vs_contracted"][k]["max_abs"] for r in per_view)
                                for k in per_view[0]["images_uncontracted_vs_contracted"]}
        tot["image_pixels_over_1e_4"] = {k: int(sum(r["images_uncontracted_vs_contracted"][k]["pixels_over_1e_4"] for r in per_view))
                                         for k in per_view[0]["images_uncontracted_vs_contracted"]}
        out[name] = dict(views=args.views, total=tot, per_view=per_view)
        print(name, json.dumps(tot))
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
