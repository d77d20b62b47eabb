"""CPU oracle of the latent epilogue — TEST INFRASTRUCTURE ONLY (imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path).

Restates, with torch CPU ops and autograd for the backward, what the reference does between the
rasterizer and the VAE decoder (paths relative to /root/reference):
  * src/model/decoder/decoder_splatting_cuda.py:46-47   mean / logvar from the rendered maps
  * src/model/diagonal_gaussian_distribution.py:55-63   clamp(logvar, -30, 20), std = exp(logvar / 2)
  * diagonal_gaussian_distribution.py:75-80             sample = mean + std * noise
  * src/model/model_wrapper.py:266-274,376              rescale(sample, 1 / supersampling)
  * model_wrapper.py:382                                skip_z = cat(color.detach(), sample)

Parity status: PINNED for sample / logvar / skip (tests/golden/latent_*.npz are produced by the
reference's own DecoderSplattingCUDA.render_to_decoder_output + DiagonalGaussianDistribution,
tests/golden/make_golden_latent.py).  `rescale` is torchvision's tensor `resize(antialias=True)`;
torchvision is not installed in the build image, so the anchor is the ATen operator it dispatches
to — `torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=True)` —
called here and when generating the fixtures.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def rescale(x: torch.Tensor, factor: int) -> torch.Tensor:
    batch, (h, w) = x.shape[:-2], x.shape[-2:]
    size = (h // factor, w // factor)
    return F.interpolate(x.reshape(1, -1, h, w), size=size, mode="bilinear", align_corners=False,
                         antialias=True).reshape(*batch, *size)


def latent_epilogue(features, mask, noise, color, factor: int, variational: bool = False,
                    interval=(-30.0, 20.0)):
    """features (..., C|2C, H, W), mask (..., H, W), noise (..., C, H, W)|None, color (..., 3, H, W)|None
    -> dict(sample, z, skip, logvar)."""
    if variational:
        mean, logvar = features.chunk(2, dim=-3)
    else:
        mean = features
        logvar = (1 - mask.detach().unsqueeze(-3)).log().expand_as(features)
    logvar = torch.clamp(logvar, *interval)
    sample = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise
    skip = None if color is None else torch.cat((color.detach(), sample), dim=-3)
    return dict(sample=sample, z=rescale(sample, factor), skip=skip, logvar=logvar)


def aa_weight_matrix(in_size: int, out_size: int) -> torch.Tensor:
    """(out, in) matrix of the separable anti-aliased bilinear filter as restated in
    csrc/latent_epilogue.hip (float32 arithmetic): rescale(x) == Wy @ x @ Wx^T."""
    import numpy as np
    scale = np.float32(in_size) / np.float32(out_size)
    M = np.zeros((out_size, in_size), np.float32)
    for o in range(out_size):
        center = np.float32(scale * np.float32(o + 0.5))
        lo = max(int(np.float32(center - scale + np.float32(0.5))), 0)
        hi = min(int(np.float32(center + scale + np.float32(0.5))), in_size)
        xs = np.arange(lo, hi, dtype=np.float32)
        w = np.maximum(np.float32(0), np.float32(1) - np.abs((xs - center + np.float32(0.5)) * (np.float32(1) / scale)))
        M[o, lo:hi] = w / w.sum(dtype=np.float32)
    return torch.from_numpy(M)
