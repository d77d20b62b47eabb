#!/usr/bin/env python
"""Generate tests/golden/adapter_*.npz by IMPORTING the reference's GaussianAdapter
(/root/reference/src/model/encoder/common/gaussian_adapter.py) in this container and running it —
forward and autograd backward — on seeded inputs shaped the way the encoder calls it
(src/model/encoder/encoder_epipolar.py:184-193: cameras (b v 1 1 1 ..), rows (b v r srf 1 ..),
depths (b v r srf spp)).  Only the vectors travel; nothing of the reference is copied.

Stubs: jaxtyping / e3nn placeholders as in tests/golden/make_golden.py; ``rotate_sh`` (e3nn Wigner-D,
not installable here) is replaced by the identity — the geometry outputs pinned here (means,
covariances, scales, rotations, opacities and their gradients) do not depend on it.
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # name: (b, v, h, w, srf, spp, color_deg, feat_deg, feat_ch, scale_min, scale_max)
    "epipolar_like": (2, 2, 6, 8, 1, 3, 1, 1, 4, 0.5, 15.0),
    "two_surfaces": (1, 3, 5, 5, 2, 1, 0, 0, 8, 0.1, 3.0),
}


def camera_batch(b, v, gen):
    """Random rigid camera-to-world matrices and normalised intrinsics with skew-free but
    non-square focal lengths and off-centre principal points."""
    q = torch.randn(b, v, 4, generator=gen)
    q = q / q.norm(dim=-1, keepdim=True)
    x, y, z, w = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(b, v, 3, 3)
    E = torch.eye(4).repeat(b, v, 1, 1)
    E[..., :3, :3] = R
    E[..., :3, 3] = torch.randn(b, v, 3, generator=gen)
    K = torch.eye(3).repeat(b, v, 1, 1)
    K[..., 0, 0] = 0.6 + torch.rand(b, v, generator=gen)
    K[..., 1, 1] = 0.6 + torch.rand(b, v, generator=gen)
    K[..., 0, 2] = 0.5 + 0.1 * (torch.rand(b, v, generator=gen) - 0.5)
    K[..., 1, 2] = 0.5 + 0.1 * (torch.rand(b, v, generator=gen) - 0.5)
    return E, K


def main():
    from make_golden import _install_stubs, _recording_module
    _install_stubs(_recording_module())
    sys.modules["src.model.encoder.common"] = type(sys)("src.model.encoder.common")
    sys.modules["src.model.encoder.common"].__path__ = [os.path.join(REF, "src/model/encoder/common")]
    ga = importlib.import_module("src.model.encoder.common.gaussian_adapter")
    ga.rotate_sh = lambda sh, rot: sh + 0 * rot.sum()          # identity with the broadcast shape
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (b, v, h, w, srf, spp, cdeg, fdeg, fch, smin, smax) in CASES.items():
        gen = torch.Generator().manual_seed(77)
        adapter = ga.GaussianAdapter(ga.GaussianAdapterCfg(smin, smax, cdeg, fdeg), fch)
        E, K = camera_batch(b, v, gen)
        r = h * w
        xs = (torch.arange(w) + 0.5) / w
        ys = (torch.arange(h) + 0.5) / h
        grid = torch.stack(torch.meshgrid(xs, ys, indexing="xy"), -1).reshape(r, 1, 2)
        coords = (grid + (torch.rand(b, v, r, srf, 2, generator=gen) - 0.5) / torch.tensor([w, h])).requires_grad_()
        depths = (0.5 + 4 * torch.rand(b, v, r, srf, spp, generator=gen)).requires_grad_()
        opac = torch.rand(b, v, r, srf, spp, generator=gen)
        raw = torch.randn(b, v, r, srf, 2 + adapter.d_in, generator=gen).requires_grad_()   # Linear output incl. the xy offset
        g = adapter.forward(E[:, :, None, None, None], K[:, :, None, None, None], coords[..., None, :],
                            depths, opac, raw[..., None, 2:], (h, w))
        gm = torch.randn(g.means.shape, generator=gen)
        gc = torch.randn(g.covariances.shape, generator=gen)
        gs = torch.randn(g.scales.shape, generator=gen)
        (g.means * gm).sum().backward(retain_graph=True)
        grads_geo = [torch.zeros_like(t) if t.grad is None else t.grad.clone() for t in (coords, depths, raw)]
        for t in (coords, depths, raw):
            t.grad = None
        ((g.means * gm).sum() + (g.covariances * gc).sum()).backward(retain_graph=True)
        grads_mc = [torch.zeros_like(t) if t.grad is None else t.grad.clone() for t in (coords, depths, raw)]
        for t in (coords, depths, raw):
            t.grad = None
        ((g.means * gm).sum() + (g.covariances * gc).sum() + (g.scales * gs).sum()).backward()
        grads_all = [torch.zeros_like(t) if t.grad is None else t.grad.clone() for t in (coords, depths, raw)]
        n = lambda t: t.detach().numpy()
        np.savez_compressed(
            os.path.join(out_dir, f"adapter_{name}.npz"),
            extrinsics=n(E), intrinsics=n(K), coordinates=n(coords), depths=n(depths), opacities=n(opac),
            raw=n(raw), image_shape=np.array([h, w], np.int32), scale_range=np.array([smin, smax], np.float32),
            sh_degrees=np.array([cdeg, fdeg, fch], np.int32),
            means=n(g.means), covariances=n(g.covariances), scales=n(g.scales), rotations=n(g.rotations),
            out_opacities=n(g.opacities), color_harmonics=n(g.color_harmonics),
            feature_harmonics=n(g.feature_harmonics),
            g_means=n(gm), g_covariances=n(gc), g_scales=n(gs),
            d_coordinates_m=n(grads_geo[0]), d_depths_m=n(grads_geo[1]), d_raw_m=n(grads_geo[2]),
            d_coordinates_mc=n(grads_mc[0]), d_depths_mc=n(grads_mc[1]), d_raw_mc=n(grads_mc[2]),
            d_coordinates_mcs=n(grads_all[0]), d_depths_mcs=n(grads_all[1]), d_raw_mcs=n(grads_all[2]))
        print(name, {k: tuple(getattr(g, k).shape) for k in ("means", "covariances", "scales", "rotations")})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("the reference is only available in the build container")
    sys.path.insert(0, REF)
    main()
