#!/usr/bin/env python
"""Generate tests/golden/latent_*.npz with the reference's own classes (this container only):
``DecoderSplattingCUDA.render_to_decoder_output`` (src/model/decoder/decoder_splatting_cuda.py:37-56)
builds the posterior from seeded feature / mask maps, ``DiagonalGaussianDistribution.sample``
(src/model/diagonal_gaussian_distribution.py:75-80) draws the sample under a fixed seed (the same
draw is recorded as `noise`), and the result is rescaled with the ATen operator torchvision's
``resize(antialias=True)`` dispatches to (torchvision itself is not installed here).  Autograd
of that chain gives the gradient vectors.  Only the vectors travel.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # name: (b, v, C, H, W, factor, variational)
    "train_like": (1, 2, 4, 64, 64, 8, False),
    "ragged_factor4": (2, 1, 3, 40, 72, 4, False),
    "variational": (1, 2, 4, 32, 48, 8, True),
}


def main():
    from make_golden import _import_reference, _install_stubs, _recording_module
    _install_stubs(_recording_module())
    dec, cs, tm = _import_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (b, v, C, H, W, f, variational) in CASES.items():
        gen = torch.Generator().manual_seed(2024)
        decoder = dec.DecoderSplattingCUDA(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), [0.0, 0.0, 0.0], variational)
        fch = 2 * C if variational else C
        feature = torch.randn(b * v, fch, H, W, generator=gen)
        if variational:
            feature[:, C:] *= 3.0
            feature[0, C, 0, :4] = torch.tensor([-50.0, 40.0, -30.0, 20.0])     # exercise the clamp and its edges
        feature.requires_grad_()
        mask = torch.rand(b * v, H, W, generator=gen) ** 0.3
        mask[0, 0, :3] = torch.tensor([0.0, 1.0, 1.0 - 1e-7])                   # empty, opaque, almost opaque
        color = torch.rand(b * v, 3, H, W, generator=gen)
        out = decoder.render_to_decoder_output(cs.RenderOutput(color, feature, mask, torch.zeros_like(mask)), b, v)
        post = out.feature_posterior
        torch.manual_seed(99)
        sample = post.sample()
        torch.manual_seed(99)
        noise = torch.randn_like(post.mean)
        assert torch.equal(sample, post.mean + post.std * noise)
        z = F.interpolate(sample.view(1, -1, H, W), size=(H // f, W // f), mode="bilinear", align_corners=False,
                          antialias=True).view(b, v, C, H // f, W // f)
        skip = torch.cat((out.color.detach(), sample), dim=-3)
        gz = torch.randn(z.shape, generator=gen)
        gskip = torch.randn(skip.shape, generator=gen)
        (z * gz).sum().backward(retain_graph=True)
        d_z_only = feature.grad.clone()
        feature.grad = None
        ((z * gz).sum() + (skip * gskip).sum()).backward()
        n = lambda t: t.detach().numpy()
        np.savez_compressed(os.path.join(out_dir, f"latent_{name}.npz"),
                            feature=n(feature), mask=n(mask), color=n(color), noise=n(noise),
                            factor=np.int32(f), variational=np.int32(variational), bv=np.array([b, v], np.int32),
                            logvar=n(post.logvar), sample=n(sample), z=n(z), skip=n(skip),
                            g_z=n(gz), g_skip=n(gskip), d_feature_z=n(d_z_only), d_feature_all=n(feature.grad))
        print(name, tuple(z.shape), tuple(skip.shape))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("the reference is only available in the build container")
    sys.path.insert(0, REF)
    main()
