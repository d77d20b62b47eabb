#!/usr/bin/env python
"""Dump input / output / gradient vectors of the REAL rasterizer fork so this repository's kernels
can be pinned against it.

Why this exists: the reference's kernel arithmetic lives in an external, un-vendored, un-pinned CUDA
package (`git+https://github.com/Chrixtar/latent-gaussian-rasterization`, requirements.txt:33 of the
reference; imported at src/model/decoder/cuda_splatting.py:6-9).  It cannot be built or imported in
the MI355X build container, so `oracle/raster_oracle.c` restates the published algorithm and says
"parity unpinned".  Run THIS script once on any CUDA machine where that package is installed:

    pip install git+https://github.com/Chrixtar/latent-gaussian-rasterization
    python tools/dump_fork_vectors.py --out tests/golden            # writes fork_*.npz (< 3 MB in total)

then commit the files.  `tests/test_fork_vectors_gpu.py` picks up `tests/golden/fork_*.npz`, replays
exactly the recorded tensors through this repository's `GaussianRasterizer` on the MI355X and
compares images (<= 1e-4 abs), the 5th return value and every input gradient (<= 1e-4 of its
scale); `tests/test_fork_vectors_cpu.py` does the same for the CPU oracle.  Until the files exist
those tests skip with a loud reason.

The script needs only torch + numpy + the rasterizer module; scene construction uses this
repository's `latentsplat_amd/synthetic.py` and the host-side wrapper math of
`latentsplat_amd/decoder/cuda_splatting.py` (both pure PyTorch, importable without the HIP library).
Every call is made exactly like the reference makes it (cuda_splatting.py:132-158): one
`GaussianRasterizationSettings` with the 12 keyword fields and one `GaussianRasterizer(settings)(...)`
call per view, `means2D` a zero tensor that requires grad, 5-tuple unpacked.

What the vectors decide (SURVEY.md Appendix A.4 "fork deltas", all [INF]/[UNK] today):
  * mask = 1 - T_final?  depth = sum alpha T z, un-normalised?      -> case fork_probe_layers
  * colour SH axis convention at degree >= 1 and the degree-4 band   -> case fork_probe_sh_axes
  * what the fifth return value is                                   -> stored as `out4` when it is a tensor
  * thresholds / clamps inherited from the intermediate pixelSplat fork -> cases fork_cfg0 / cfg1 / cfg3

`--module` selects the rasterizer module (default `diff_gaussian_rasterization`); the test-suite runs
the script against this repository's own drop-in and against the CPU oracle to keep the file format
and the consumer honest.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FORMAT_VERSION = 1


def _boundary(scene, H, W, bg):
    """Scene -> exactly the tensors the reference hands to the rasterizer, per view (CPU)."""
    from latentsplat_amd.decoder import cuda_splatting as cs
    from latentsplat_amd.decoder.geometry import get_fov
    V = scene.extrinsics.shape[0]
    means = scene.means[None].expand(V, -1, -1)
    covs = scene.covariances[None].expand(V, -1, -1, -1)
    ext, nr, fr, means, covs = cs._scale_scene(scene.extrinsics, scene.near, scene.far, means, covs)
    fov_x, fov_y = get_fov(scene.intrinsics).unbind(-1)
    cams = cs._cameras(ext, nr, fr, fov_x, fov_y)
    csh = None if scene.color_sh is None else scene.color_sh[None]
    fsh = None if scene.feature_sh is None else scene.feature_sh[None]
    degree, shs, colors_precomp, features = cs._payload(means, cams.campos, csh, fsh, True)
    return dict(V=V, H=H, W=W, sh_degree=int(degree), bg=torch.tensor(bg, dtype=torch.float32),
                viewmatrix=cams.view_matrix.contiguous(), projmatrix=cams.full_projection.contiguous(),
                campos=cams.campos.contiguous(), tanfovx=cams.tan_fov_x, tanfovy=cams.tan_fov_y,
                means3D=means.contiguous(), cov3D=cs._pack_covariances(covs).contiguous(),
                opacities=scene.opacities[:, None].contiguous(),
                shs=None if shs is None else shs[0].contiguous(),
                features=None if features is None else features.contiguous())


def _probe_layers():
    """Three fronto-parallel Gaussians stacked on the optical axis + one off to the side: pins
    mask = 1 - T, the depth output (sum alpha T z, normalised or not) and the background term."""
    from latentsplat_amd.synthetic import Scene
    means = torch.tensor([[0.0, 0.0, 2.0], [0.02, 0.01, 3.0], [-0.03, 0.02, 5.0], [0.6, -0.4, 4.0]])
    s = torch.tensor([0.08, 0.15, 0.4, 0.2])
    cov = torch.diag_embed(torch.stack([s * s, s * s, 0.25 * s * s], -1))
    opac = torch.tensor([0.5, 0.7, 0.9, 0.3])
    color_sh = torch.tensor([[1.0, 0.2, -0.5], [0.3, 0.9, 0.1], [-0.2, 0.4, 1.2], [0.8, 0.8, 0.8]])[..., None]
    feature_sh = torch.tensor([[0.2, -0.1], [0.4, 0.3], [-0.3, 0.5], [0.1, 0.1]])[..., None]
    K = torch.tensor([[0.8, 0, 0.5], [0, 0.8, 0.5], [0, 0, 1.0]])
    return Scene(means, cov, opac, color_sh, feature_sh, torch.eye(4)[None], K[None], torch.tensor([1.0]), torch.tensor([50.0]))


def _probe_sh_axes():
    """Isotropic Gaussians seen along distinct directions, each with ONE non-zero SH coefficient per
    colour channel in bands 1..4: the rendered colours read off the basis function the kernel uses
    for that coefficient, i.e. the axis convention / ordering / sign of every band."""
    from latentsplat_amd.synthetic import Scene
    dirs = torch.tensor([[0.0, 0.0, 1.0], [0.35, 0.0, 1.0], [-0.35, 0.0, 1.0], [0.0, 0.35, 1.0], [0.0, -0.35, 1.0],
                         [0.3, 0.25, 1.0], [-0.3, 0.25, 1.0], [0.3, -0.25, 1.0], [-0.3, -0.25, 1.0],
                         [0.15, 0.3, 1.0], [-0.15, -0.3, 1.0], [0.33, 0.1, 1.0]])
    n = dirs.shape[0]
    means = dirs * torch.linspace(3.0, 5.0, n)[:, None]
    s = torch.full((n,), 0.12)
    cov = torch.diag_embed(torch.stack([s * s, s * s, s * s], -1))
    opac = torch.full((n,), 0.8)
    color_sh = torch.zeros(n, 3, 25)
    color_sh[:, :, 0] = 0.3                      # keeps 0.5 + SH positive next to the probed coefficient
    gen = torch.Generator().manual_seed(7)
    for i in range(n):
        for c in range(3):
            k = 1 + int(torch.randint(0, 24, (1,), generator=gen))
            color_sh[i, c, k] = 0.6 * (1 if (i + c) % 2 == 0 else -1)
    K = torch.tensor([[0.8, 0, 0.5], [0, 0.8, 0.5], [0, 0, 1.0]])
    return Scene(means, cov, opac, color_sh, None, torch.eye(4)[None], K[None], torch.tensor([1.0]), torch.tensor([50.0]))


def cases():
    """name -> (scene, H, W, background).  Fixture sizes: every .npz stays well under 1 MB."""
    from latentsplat_amd.synthetic import make_scene
    return {
        # BASELINE configs[0] flavour: RGB, SH degree 0
        "fork_cfg0_rgb_deg0": (make_scene(1500, image_size=64, views=1, color_sh_degree=0, feature_channels=None, seed=101), 64, 64, (0.1, 0.2, 0.3)),
        # BASELINE configs[1]/[2] flavour: 4 latent channels + opacity, no colour
        "fork_cfg1_feat4": (make_scene(1500, image_size=64, views=2, color_sh_degree=None, feature_channels=4, seed=102), 64, 64, (0.0, 0.0, 0.0)),
        # BASELINE configs[3]/[4] flavour: colour SH degree 4 (25 coefficients) + 4-ch latent SH degree 2
        "fork_cfg3_sh4_feat4": (make_scene(1200, image_size=64, views=2, color_sh_degree=4, feature_channels=4, feature_sh_degree=2, seed=103), 64, 64, (0.0, 0.0, 0.0)),
        # ragged image size, opaque stacks (early termination), larger splats
        "fork_ragged_opaque": (make_scene(800, image_size=64, views=1, color_sh_degree=1, feature_channels=8, seed=104, sigma_px=(1.0, 6.0), opacity_scale=1.0), 56, 72, (0.5, 0.5, 0.5)),
        "fork_probe_layers": (_probe_layers(), 64, 64, (0.25, 0.5, 0.75)),
        "fork_probe_sh_axes": (_probe_sh_axes(), 96, 96, (0.0, 0.0, 0.0)),
    }


def run_case(mod, name, scene, H, W, bg, device, seed=0):
    bi = _boundary(scene, H, W, bg)
    dev = torch.device(device)
    gen = torch.Generator().manual_seed(seed)
    rec = dict(format_version=np.int32(FORMAT_VERSION), case=np.array(name), H=np.int32(H), W=np.int32(W),
               V=np.int32(bi["V"]), sh_degree=np.int32(bi["sh_degree"]), bg=bi["bg"].numpy())
    for v in range(bi["V"]):
        req = lambda t: None if t is None else t.to(dev).clone().requires_grad_(True)
        means3D, cov3D, opac = req(bi["means3D"][v]), req(bi["cov3D"][v]), req(bi["opacities"])
        shs = req(bi["shs"])
        feats = None if bi["features"] is None else req(bi["features"][v])
        means2D = torch.zeros_like(means3D, requires_grad=True)   # reference: cuda_splatting.py:126-130
        settings = mod.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=float(bi["tanfovx"][v]), tanfovy=float(bi["tanfovy"][v]),
            bg=bi["bg"].to(dev), scale_modifier=1.0, viewmatrix=bi["viewmatrix"][v].to(dev),
            projmatrix=bi["projmatrix"][v].to(dev), sh_degree=bi["sh_degree"], campos=bi["campos"][v].to(dev),
            prefiltered=False, debug=False)
        rasterizer = mod.GaussianRasterizer(settings)
        out = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, features=feats,
                         opacities=opac, cov3D_precomp=cov3D)
        assert len(out) == 5, f"expected the fork's 5-tuple, got {len(out)} values"
        image, feature_map, mask, depth, fifth = out
        loss = 0.0
        ups = {}
        for key, o in (("image", image), ("feature_map", feature_map)):
            if o is not None:
                g = torch.randn(o.shape, generator=gen)
                ups[key] = g
                loss = loss + (o * g.to(dev)).sum()
        loss.backward()
        n = lambda t: None if t is None else t.detach().float().cpu().numpy()
        p = f"v{v}_"
        for key, t in (("means3D", means3D), ("cov3D", cov3D), ("opacities", opac), ("shs", shs), ("features", feats)):
            if t is not None:
                rec[p + "in_" + key] = n(t)
                rec[p + "grad_" + key] = n(t.grad)
        rec[p + "grad_means2D"] = n(means2D.grad) if means2D.grad is not None else np.zeros((0,), np.float32)
        for key in ("viewmatrix", "projmatrix", "campos"):
            rec[p + key] = bi[key][v].numpy()
        rec[p + "tanfov"] = np.array([float(bi["tanfovx"][v]), float(bi["tanfovy"][v])], np.float32)
        for key, t in (("image", image), ("feature_map", feature_map), ("mask", mask), ("depth", depth)):
            if t is not None:
                rec[p + "out_" + key] = n(t)
        if torch.is_tensor(fifth):
            rec[p + "out4"] = fifth.detach().cpu().numpy()
        for key, g in ups.items():
            rec[p + "upstream_" + key] = g.numpy()
    return rec


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--module", default="diff_gaussian_rasterization")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--prefix", default="", help="file name prefix (tests use it to keep self-test dumps apart)")
    ap.add_argument("--only", default="", help="comma-separated case names")
    args = ap.parse_args(argv)
    mod = importlib.import_module(args.module)
    os.makedirs(args.out, exist_ok=True)
    meta = dict(module=args.module, module_file=getattr(mod, "__file__", "?"), torch=torch.__version__,
                device=(torch.cuda.get_device_name(0) if args.device.startswith("cuda") and torch.cuda.is_available() else args.device))
    for name, (scene, H, W, bg) in cases().items():
        if args.only and name not in args.only.split(","):
            continue
        rec = run_case(mod, name, scene, H, W, bg, args.device)
        rec["meta_json"] = np.array(json.dumps(meta))
        path = os.path.join(args.out, args.prefix + name + ".npz")
        np.savez_compressed(path, **rec)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
