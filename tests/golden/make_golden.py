#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING the reference's own Python wrapper (this container
only: /root/reference does not exist on the GPU box, so only the resulting vectors travel).

What is pinned:
  * boundary_*.npz  — for seeded inputs, the exact ``GaussianRasterizationSettings`` fields and
    the kwargs tensors that the reference's ``DecoderSplattingCUDA.forward`` / ``render_cuda``
    (/root/reference/src/model/decoder/{decoder_splatting_cuda,cuda_splatting}.py) hand to
    ``diff_gaussian_rasterization.GaussianRasterizer`` — recorded with a fake rasterizer;
  * decoder_*.npz   — the ``DecoderOutput`` tensors of the reference wrapper when the module
    ``diff_gaussian_rasterization`` is served by this repo's CPU oracle (oracle/oracle.py).
    They pin the wrapper's pre/post-processing around the kernel (scale invariance, SH feature
    evaluation, mask -> logvar, stacking); the kernel arithmetic itself stays "parity unpinned"
    (the reference ships no kernel source or vectors).

The reference is imported with the stubs described in SURVEY.md Appendix B: ``jaxtyping`` and
``e3nn`` placeholders, empty namespace packages for the heavy ``src.*`` package __init__s, and a
replacement ``diff_gaussian_rasterization`` module.  Nothing from the reference is copied.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from latentsplat_amd.synthetic import make_scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402

RECORD = []


def _install_stubs(rasterizer_module):
    class _AnnMeta(type):  # Float[Tensor, "b 3"] -> a plain type usable in Optional[...] and `|`
        def __getitem__(cls, item):
            return cls

        def __or__(cls, other):
            return cls

        def __ror__(cls, other):
            return cls

    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Int64", "Bool", "Shaped", "UInt8", "Int"):
        setattr(jt, n, _AnnMeta(n, (), {}))
    sys.modules["jaxtyping"] = jt
    e3 = types.ModuleType("e3nn")
    o3 = types.ModuleType("e3nn.o3")
    o3.matrix_to_angles = None
    o3.wigner_D = None
    e3.o3 = o3
    sys.modules["e3nn"], sys.modules["e3nn.o3"] = e3, o3
    sys.modules["diff_gaussian_rasterization"] = rasterizer_module
    for name in ("src", "src.model", "src.model.encoder", "src.model.encoder.epipolar",
                 "src.model.decoder", "src.geometry", "src.misc"):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = m


def _recording_module():
    m = types.ModuleType("diff_gaussian_rasterization")
    m.GaussianRasterizationSettings = orc.GaussianRasterizationSettings

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.s = raster_settings

        def forward(self, **kw):
            RECORD.append((self.s, {k: (None if v is None else v.detach().clone()) for k, v in kw.items()}))
            H, W = self.s.image_height, self.s.image_width
            G = kw["means3D"].shape[0]
            f = kw.get("features")
            has_color = kw.get("shs") is not None or kw.get("colors_precomp") is not None
            return (torch.zeros(3, H, W) if has_color else None,
                    torch.zeros(f.shape[1], H, W) if f is not None else None,
                    torch.zeros(1, H, W), torch.zeros(1, H, W), torch.zeros(G, dtype=torch.int32))

    m.GaussianRasterizer = GaussianRasterizer
    return m


def _oracle_module():
    m = types.ModuleType("diff_gaussian_rasterization")
    m.GaussianRasterizationSettings = orc.GaussianRasterizationSettings
    m.GaussianRasterizer = orc.GaussianRasterizer
    return m


def _import_reference():
    for k in [k for k in sys.modules if k.startswith("src.model.decoder.") or k in ("src.model.types",)]:
        del sys.modules[k]
    dec = importlib.import_module("src.model.decoder.decoder_splatting_cuda")
    cs = importlib.import_module("src.model.decoder.cuda_splatting")
    types_mod = importlib.import_module("src.model.types")
    return dec, cs, types_mod


SCENARIOS = {
    # name: (scene kwargs, decoder.forward kwargs, variational)
    "train_rgb4_feat4": (dict(G=600, image_size=64, views=3, color_sh_degree=4, feature_channels=4, feature_sh_degree=2), {}, False),
    "features_only": (dict(G=500, image_size=64, views=2, color_sh_degree=4, feature_channels=4, feature_sh_degree=2), dict(return_colors=False), False),
    "variational_8ch": (dict(G=400, image_size=48, views=2, color_sh_degree=2, feature_channels=8, feature_sh_degree=1), {}, True),
    "disparity_depth": (dict(G=400, image_size=48, views=2, color_sh_degree=1, feature_channels=4, feature_sh_degree=0), dict(depth_mode="disparity"), False),
}


def _scene_batch(kw):
    """b=2 scenes (different seeds) x v views, with per-view near/far that differ."""
    kw = dict(kw)
    G = kw.pop("G")
    size = kw.pop("image_size")
    scenes = [make_scene(G, image_size=size, seed=1234 + s, **kw) for s in range(2)]
    stack = lambda name: None if getattr(scenes[0], name) is None else torch.stack([getattr(s, name) for s in scenes])
    near = torch.stack([s.near for s in scenes]).clone()
    near[:, 1:] = near[:, 1:] * 1.25          # exercise the per-view scale invariance
    return dict(means=stack("means"), covariances=stack("covariances"), opacities=stack("opacities"),
                color_harmonics=stack("color_sh"), feature_harmonics=stack("feature_sh"),
                extrinsics=stack("extrinsics"), intrinsics=stack("intrinsics"), near=near,
                far=stack("far"), image_shape=(size, size))


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def orthographic(out_dir):
    """render_cuda_orthographic (/root/reference/src/model/decoder/cuda_splatting.py:170-292): the
    fake-orthographic camera.  The reference only works for batch 1 (it assigns the (batch,) distance
    to ONE element of a single 4x4 `move_back`, :226-227) and hands TENSOR-valued tanfovx / tanfovy
    to the settings (:260-261).  Recorded: the boundary (settings + kwargs) with the recording
    rasterizer, the RenderOutput with the oracle rasterizer, and the `dump` escape hatch."""
    sc = make_scene(500, image_size=48, views=1, color_sh_degree=2, feature_channels=4, feature_sh_degree=1, seed=77)
    g = torch.Generator().manual_seed(3)
    ext = sc.extrinsics.clone()
    ext[0, :3, 3] = torch.tensor([0.3, -0.2, -1.0])
    args = dict(extrinsics=ext, width=torch.tensor([3.0]), height=torch.tensor([2.4]), near=sc.near.clone(), far=sc.far.clone(),
                image_shape=(40, 48), background_features=torch.rand(1, 3, generator=g),
                gaussian_means=sc.means[None].contiguous(), gaussian_covariances=sc.covariances[None].contiguous(),
                gaussian_opacities=sc.opacities[None].contiguous(),
                gaussian_color_sh_coefficients=sc.color_sh[None].contiguous(),
                gaussian_feature_sh_coefficients=sc.feature_sh[None].contiguous())
    _install_stubs(_recording_module())
    dec, cs, tm = _import_reference()
    RECORD.clear()
    dump = {}
    cs.render_cuda_orthographic(**args, dump=dump)
    assert len(RECORD) == 1
    s, kw = RECORD[0]
    rec = {f"in_{k}": _np(v) for k, v in args.items() if torch.is_tensor(v)}
    rec["in_image_shape"] = np.array(args["image_shape"], np.int32)
    rec["call0_tanfovx"] = np.float32(float(s.tanfovx)); rec["call0_tanfovy"] = np.float32(float(s.tanfovy))
    rec["call0_tanfov_is_tensor"] = np.array([torch.is_tensor(s.tanfovx), torch.is_tensor(s.tanfovy)])
    rec["call0_sh_degree"] = np.int32(s.sh_degree)
    for fld in ("bg", "viewmatrix", "projmatrix", "campos"):
        rec[f"call0_{fld}"] = _np(getattr(s, fld))
    for k, v in kw.items():
        if v is not None and k != "means2D":
            rec[f"call0_{k}"] = _np(v)
    for k, v in dump.items():
        rec[f"dump_{k}"] = _np(v)
    _install_stubs(_oracle_module())
    dec, cs, tm = _import_reference()
    out = cs.render_cuda_orthographic(**args)
    np.savez_compressed(os.path.join(out_dir, "orthographic.npz"), **rec, out_color=_np(out.color),
                        out_feature=_np(out.feature), out_mask=_np(out.mask), out_depth=_np(out.depth))
    print("orthographic", {k: tuple(getattr(out, k).shape) for k in ("color", "feature", "mask", "depth")})


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, (skw, fkw, variational) in SCENARIOS.items():
        batch = _scene_batch(skw)
        bg = [0.1, 0.3, 0.7]
        # ---- pass 1: record the boundary with a fake rasterizer ----
        _install_stubs(_recording_module())
        dec, cs, tm = _import_reference()
        RECORD.clear()
        decoder = dec.DecoderSplattingCUDA(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), bg, variational)
        gauss = tm.Gaussians(batch["means"], batch["covariances"], batch["opacities"],
                             batch["color_harmonics"], batch["feature_harmonics"])
        decoder.forward(gauss, batch["extrinsics"], batch["intrinsics"], batch["near"], batch["far"],
                        batch["image_shape"], **fkw)
        rec = {}
        for i, (s, kw) in enumerate(RECORD):
            rec[f"call{i}_tanfovx"] = np.float32(s.tanfovx)
            rec[f"call{i}_tanfovy"] = np.float32(s.tanfovy)
            rec[f"call{i}_sh_degree"] = np.int32(s.sh_degree)
            for fld in ("bg", "viewmatrix", "projmatrix", "campos"):
                rec[f"call{i}_{fld}"] = _np(getattr(s, fld))
            for k, v in kw.items():
                if v is not None and k != "means2D":
                    rec[f"call{i}_{k}"] = _np(v)
        rec["num_calls"] = np.int32(len(RECORD))
        inputs = {f"in_{k}": _np(v) for k, v in batch.items() if torch.is_tensor(v)}
        inputs["in_image_shape"] = np.array(batch["image_shape"], np.int32)
        inputs["in_bg"] = np.array(bg, np.float32)
        np.savez_compressed(os.path.join(out_dir, f"boundary_{name}.npz"), **inputs, **rec)

        # ---- pass 2: reference wrapper + oracle rasterizer -> DecoderOutput ----
        _install_stubs(_oracle_module())
        dec, cs, tm = _import_reference()
        decoder = dec.DecoderSplattingCUDA(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), bg, variational)
        gauss = tm.Gaussians(batch["means"], batch["covariances"], batch["opacities"],
                             batch["color_harmonics"], batch["feature_harmonics"])
        out = decoder.forward(gauss, batch["extrinsics"], batch["intrinsics"], batch["near"], batch["far"],
                              batch["image_shape"], **fkw)
        res = dict(mask=_np(out.mask), depth=_np(out.depth))
        if out.color is not None:
            res["color"] = _np(out.color)
        if out.feature_posterior is not None:
            res["posterior_mean"] = _np(out.feature_posterior.mean)
            res["posterior_logvar"] = _np(out.feature_posterior.logvar)
        np.savez_compressed(os.path.join(out_dir, f"decoder_{name}.npz"), **res)
        print(name, "calls:", len(RECORD), {k: v.shape for k, v in res.items()})

    orthographic(out_dir)

    # ---- helper-level vectors: get_fov / get_projection_matrix / eval_sh ----
    _install_stubs(_recording_module())
    dec, cs, tm = _import_reference()
    proj = importlib.import_module("src.geometry.projection")
    shu = importlib.import_module("src.misc.sh_utils")
    g = torch.Generator().manual_seed(5)
    K = torch.eye(3).repeat(6, 1, 1)
    K[:, 0, 0] = torch.rand(6, generator=g) + 0.5
    K[:, 1, 1] = torch.rand(6, generator=g) + 0.5
    K[:, 0, 2] = 0.5 + (torch.rand(6, generator=g) - 0.5) * 0.2
    K[:, 1, 2] = 0.5 + (torch.rand(6, generator=g) - 0.5) * 0.2
    fov = proj.get_fov(K)
    near = torch.rand(6, generator=g) + 0.2
    far = near + torch.rand(6, generator=g) * 50 + 1
    P = cs.get_projection_matrix(near, far, fov[:, 0], fov[:, 1])
    sh = torch.randn(50, 5, 25, generator=g)
    d = torch.randn(50, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    evals = {f"eval_sh_deg{k}": _np(shu.eval_sh(k, sh, d)) for k in range(5)}
    np.savez_compressed(os.path.join(out_dir, "helpers.npz"), K=_np(K), fov=_np(fov), near=_np(near),
                        far=_np(far), proj=_np(P), sh=_np(sh), dirs=_np(d), **evals)
    print("helpers done")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("the reference is only available in the build container")
    sys.path.insert(0, REF)
    main()
