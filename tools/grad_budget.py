#!/usr/bin/env python
"""Where does the gradient error budget of BASELINE configs[2] go?  

Runs the 300 k-Gaussian scene forward + backward on the MI355X and compares the gradients with
  o32: the oracle's float32 restatement of the PUBLISHED backward recurrence (back to front, T rebuilt as
       T / (1 - alpha) from the stored T_final, running float blend of the colour behind an entry), and
  o64: the same derivative evaluated in double under the same float32 keep / skip / stop decisions
       (oracle_render_backward_f64) — the exact gradient of the rendered function.
Prints, per gradient tensor, the worst and the 99.9th-percentile row error of each pair relative to the
tensor's scale max(1, max |o64|), and per-element relative errors, plus where the worst HIP-vs-o32 rows sit.
Writes gpurun_out/grad_budget.json.   usage: python tools/grad_budget.py [gaussians] [size]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    from latentsplat_amd.rasterizer import rasterize_views
    dev = torch.device("cuda", 0)
    sc = util.make_scene(G, image_size=S, views=2, color_sh_degree=None, feature_channels=4)
    bi = util.boundary_inputs(sc, S, S)
    views = util.view_table(bi, dev)
    req = lambda k: bi[k].to(dev).clone().requires_grad_(True)
    m, c, o, f = req("means"), req("cov6"), req("opac"), req("features")
    out = rasterize_views(views[:1], S, S, 0, m[:1], c[:1], o, features=f[:1])
    g = torch.randn(out[1].shape, generator=torch.Generator().manual_seed(11))
    grads = torch.autograd.grad((out[1] * g.to(dev)).sum(), (m, c, o, f))
    ofw = util.oracle_forward(bi, 0)
    n = lambda t: None if t is None else t.detach().numpy()
    args = (util.oracle_view(bi, 0), n(bi["means"][0]), n(bi["cov6"][0]), n(bi["opac"]), None, None, n(bi["features"][0]), ofw, None, g[0].numpy())
    b32 = util.orc.backward(*args)
    b64 = util.orc.backward(*args, f64=True)
    direct, behind = util.fragile_gaussians(ofw, S)
    frag = np.zeros(G, bool); frag[direct] = True; frag[behind] = True
    res = {"gaussians": G, "size": S, "fragile_rows": int(frag.sum())}
    for name, got, k in (("means3D", grads[0][0], "means3D"), ("cov3D", grads[1][0], "cov3D"), ("opacities", grads[2], "opacities"), ("features", grads[3][0], "features")):
        hip = got.cpu().numpy().astype(np.float64).reshape(G, -1)
        w32, w64 = b32[k].astype(np.float64).reshape(G, -1), b64[k].astype(np.float64).reshape(G, -1)
        scale = max(1.0, np.abs(w64).max())
        row = lambda a, b: np.abs(a - b).max(1)
        e = {"hip_vs_o64": row(hip, w64), "o32_vs_o64": row(w32, w64), "hip_vs_o32": row(hip, w32)}
        r = {"scale": scale}
        for key, err in e.items():
            clean = err[~frag]
            r[key] = dict(worst_over_scale=float(clean.max() / scale), p999_over_scale=float(np.percentile(clean, 99.9) / scale),
                          median_over_scale=float(np.median(clean) / scale))
        # per-element mixed tolerance |err| <= tol * max(1, |want_elem|): the smallest tol that holds for all non-fragile rows
        for key, (a, b) in (("hip_vs_o64", (hip, w64)), ("hip_vs_o32", (hip, w32)), ("o32_vs_o64", (w32, w64))):
            rel = (np.abs(a - b) / np.maximum(1.0, np.abs(b)))[~frag]
            r[key]["worst_per_element_mixed"] = float(rel.max())
            relrow = (np.abs(a - b).max(1) / np.maximum(1.0, np.abs(b).max(1)))[~frag]      # per-ROW scale max(1, |row|_inf)
            r[key]["worst_per_row_mixed"] = float(relrow.max())
            r[key]["p999_per_row_mixed"] = float(np.percentile(relrow, 99.9))
        worst = np.argsort(np.where(frag, 0, e["hip_vs_o32"]))[-5:][::-1]
        r["worst_rows_hip_vs_o32"] = [dict(row=int(i), err_over_scale=float(e["hip_vs_o32"][i] / scale), o32_vs_o64=float(e["o32_vs_o64"][i] / scale),
                                           hip_vs_o64=float(e["hip_vs_o64"][i] / scale), opacity=float(bi["opac"][i]), radius=int(ofw["radii"][i])) for i in worst]
        res[name] = r
        print(f"dL/d{name:10s} scale {scale:9.3e} | worst err/scale  HIP-o64 {r['hip_vs_o64']['worst_over_scale']:.2e}  o32-o64 {r['o32_vs_o64']['worst_over_scale']:.2e}  "
              f"HIP-o32 {r['hip_vs_o32']['worst_over_scale']:.2e} | p99.9  {r['hip_vs_o64']['p999_over_scale']:.1e} {r['o32_vs_o64']['p999_over_scale']:.1e} {r['hip_vs_o32']['p999_over_scale']:.1e}"
              f" | per-element mixed  {r['hip_vs_o64']['worst_per_element_mixed']:.2e} {r['o32_vs_o64']['worst_per_element_mixed']:.2e} {r['hip_vs_o32']['worst_per_element_mixed']:.2e}"
              f" | per-row mixed  {r['hip_vs_o64']['worst_per_row_mixed']:.2e} {r['o32_vs_o64']['worst_per_row_mixed']:.2e} {r['hip_vs_o32']['worst_per_row_mixed']:.2e}"
              f" (p99.9 {r['hip_vs_o64']['p999_per_row_mixed']:.1e} {r['o32_vs_o64']['p999_per_row_mixed']:.1e} {r['hip_vs_o32']['p999_per_row_mixed']:.1e})")
        for wr in r["worst_rows_hip_vs_o32"][:3]:
            print("     worst HIP-o32 row", wr)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "grad_budget.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
