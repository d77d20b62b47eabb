"""Replay of the vectors written by tools/dump_fork_vectors.py (format_version 1) through any module
with the `diff_gaussian_rasterization` API, and the comparison rules of the fork-parity tests:
images / mask / depth <= 1e-4 abs, the fifth return value equal, gradients <= 1e-4 of their scale."""
from __future__ import annotations

import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ABS_TOL = 1e-4


def fork_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "fork_*.npz")))


SKIP_REASON = ("NO FORK VECTORS: tests/golden/fork_*.npz are absent, so the kernels are still pinned only to the "
               "in-repo oracle ('parity unpinned', oracle/raster_oracle.c).  Run `python tools/dump_fork_vectors.py` "
               "on a CUDA machine with github.com/Chrixtar/latent-gaussian-rasterization installed and commit the files.")


def replay(path, mod, device):
    """Feeds the recorded tensors of every view to `mod` exactly like the dump did; returns
    {key: (got, want)} over outputs and gradients."""
    z = np.load(path, allow_pickle=False)
    assert int(z["format_version"]) == 1
    dev = torch.device(device)
    H, W, V, deg = int(z["H"]), int(z["W"]), int(z["V"]), int(z["sh_degree"])
    res = {}
    for v in range(V):
        p = f"v{v}_"
        t = lambda k: torch.from_numpy(z[p + "in_" + k]).to(dev).requires_grad_(True) if (p + "in_" + k) in z.files else None
        means3D, cov3D, opac, shs, feats = t("means3D"), t("cov3D"), t("opacities"), t("shs"), t("features")
        means2D = torch.zeros_like(means3D, requires_grad=True)
        c = lambda k: torch.from_numpy(z[p + k]).to(dev)
        settings = mod.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=float(z[p + "tanfov"][0]), tanfovy=float(z[p + "tanfov"][1]),
            bg=torch.from_numpy(z["bg"]).to(dev), scale_modifier=1.0, viewmatrix=c("viewmatrix"), projmatrix=c("projmatrix"),
            sh_degree=deg, campos=c("campos"), prefiltered=False, debug=False)
        out = mod.GaussianRasterizer(settings)(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None,
                                               features=feats, opacities=opac, cov3D_precomp=cov3D)
        image, feature_map, mask, depth, fifth = out
        loss = 0.0
        for key, o in (("image", image), ("feature_map", feature_map)):
            if (p + "upstream_" + key) in z.files:
                assert o is not None, f"{key} missing from the replayed module's output"
                loss = loss + (o * torch.from_numpy(z[p + "upstream_" + key]).to(dev)).sum()
        loss.backward()
        n = lambda x: None if x is None else x.detach().float().cpu().numpy()
        for key, o in (("image", image), ("feature_map", feature_map), ("mask", mask), ("depth", depth)):
            if (p + "out_" + key) in z.files:
                res[p + "out_" + key] = (n(o).reshape(z[p + "out_" + key].shape), z[p + "out_" + key])
        if (p + "out4") in z.files and torch.is_tensor(fifth):
            res[p + "out4"] = (fifth.detach().cpu().numpy(), z[p + "out4"])
        for key, x in (("means3D", means3D), ("cov3D", cov3D), ("opacities", opac), ("shs", shs), ("features", feats)):
            if x is not None:
                res[p + "grad_" + key] = (n(x.grad), z[p + "grad_" + key])
        if z[p + "grad_means2D"].size and means2D.grad is not None:
            res[p + "grad_means2D"] = (n(means2D.grad), z[p + "grad_means2D"])
    return res


def compare(res, image_tol=ABS_TOL, grad_tol=ABS_TOL, image_outliers=0.0):
    """Returns a list of human-readable mismatches (empty = parity).  `image_outliers`: fraction of
    pixels allowed to differ by more (near-discontinuity decisions; 0 for exact replays)."""
    bad = []
    for key, (got, want) in sorted(res.items()):
        got, want = np.asarray(got), np.asarray(want)
        if got.shape != want.shape:
            bad.append(f"{key}: shape {got.shape} vs {want.shape}")
            continue
        if key.endswith("out4"):
            if not np.array_equal(got, want):
                bad.append(f"{key}: fifth return value differs in {(got != want).sum()} of {got.size} entries")
            continue
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        if "_out_" in key:
            n_bad = int((err > image_tol).sum())
            if n_bad > image_outliers * err.size:
                bad.append(f"{key}: {n_bad} of {err.size} values off by more than {image_tol:.0e} (max {err.max():.3e})")
        else:
            scale = max(1.0, float(np.abs(want).max()))
            if err.max() > grad_tol * scale:
                bad.append(f"{key}: max err {err.max():.3e} > {grad_tol:.0e} x scale {scale:.3e}")
    return bad
