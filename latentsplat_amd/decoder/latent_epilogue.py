"""Latent epilogue of the decoder: posterior sample, anti-aliased
downscale and skip concatenation of the rendered feature maps as ONE HIP launch each way
(csrc/latent_epilogue.hip, C ABI include/lsr_latent.h).

Reference counterparts (paths relative to /root/reference):
  * ``DecoderSplattingCUDA.render_to_decoder_output`` — src/model/decoder/decoder_splatting_cuda.py:37-56
    (``logvar = log(1 - mask.detach())``, or the upper half of the channels when ``variational``);
  * ``DiagonalGaussianDistribution.sample`` — src/model/diagonal_gaussian_distribution.py:55-63,75-80;
  * ``ModelWrapper.rescale`` — src/model/model_wrapper.py:266-274 (torchvision ``resize`` with
    ``antialias=True``), called at :376 with ``1 / supersampling_factor``;
  * the skip concatenation ``cat((output.color.detach(), latent_sample), dim=-3)`` — :382.

``sample_rescale_skip`` is what ``training_step`` (:374-383) / ``test_step`` do between
``decoder.forward`` and ``autoencoder.decode``.  ROCm float32 tensors only; no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from fractions import Fraction
from typing import NamedTuple, Optional

import torch
from torch import Tensor

from .. import _lib
from .._lib import LatentDims, LatentInputs, LatentOutGrads, LatentOutputs
from .decoder import DecoderOutput

LOGVAR_FROM_MASK, LOGVAR_FROM_FEATURES = 0, 1


def _ptr(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t: Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _require(*tensors: Optional[Tensor]) -> None:
    for t in tensors:
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise _lib.LsrError("the latent epilogue needs float32 ROCm tensors (no CPU fallback)")


class _LatentEpilogue(torch.autograd.Function):
    """features (V, C|2C, H, W), mask (V, H, W)|None, noise (V, C, H, W)|None, color (V,3,H,W)|None
    -> skip (V, cc+C, H, W)|None, z (V, C, h, w)|None, logvar (V, 1|C, H, W)|None."""

    @staticmethod
    def forward(ctx, features, mask, noise, color, out_size, logvar_mode, want_skip, want_logvar, interval):
        _require(features, mask, noise, color)
        lib = _lib.load()
        features = features.contiguous()
        mask = None if mask is None else mask.contiguous()
        noise = None if noise is None else noise.contiguous()
        color = None if color is None else color.contiguous()
        V, F, H, W = features.shape
        Cc = F // 2 if logvar_mode == LOGVAR_FROM_FEATURES else F
        cc = 3 if (color is not None and want_skip) else 0
        oh, ow = out_size if out_size is not None else (0, 0)
        dims = LatentDims(V, Cc, H, W, oh, ow, logvar_mode, cc, interval[0], interval[1], 0, 0)
        dev = features.device
        skip = torch.empty((V, cc + Cc, H, W), device=dev) if want_skip else None
        z = torch.empty((V, Cc, oh, ow), device=dev) if out_size is not None else None
        lv_ch = Cc if logvar_mode == LOGVAR_FROM_FEATURES else 1
        logvar = torch.empty((V, lv_ch, H, W), device=dev) if want_logvar else None
        inp = LatentInputs(_ptr(features), _ptr(mask), _ptr(noise), _ptr(color))
        out = LatentOutputs(_ptr(skip), _ptr(z), _ptr(logvar))
        _lib.check(lib.lsr_latent_forward(C.byref(dims), C.byref(inp), C.byref(out), _stream(features)),
                   "lsr_latent_forward")
        ctx.save_for_backward(features, mask, noise)
        ctx.dims = dims
        ctx.set_materialize_grads(False)
        outs = (skip, z, logvar)
        ctx.mark_non_differentiable(*[t for t in (logvar,) if t is not None])
        return outs

    @staticmethod
    def backward(ctx, g_skip, g_z, _g_logvar):
        features, mask, noise = ctx.saved_tensors
        lib = _lib.load()
        g_skip = None if g_skip is None else g_skip.contiguous()
        g_z = None if g_z is None else g_z.contiguous()
        d_features = torch.empty_like(features)
        if g_skip is None and g_z is None:
            return d_features.zero_(), None, None, None, None, None, None, None, None
        inp = LatentInputs(_ptr(features), _ptr(mask), _ptr(noise), None)
        dout = LatentOutGrads(_ptr(g_skip), _ptr(g_z))
        _lib.check(lib.lsr_latent_backward(C.byref(ctx.dims), C.byref(inp), C.byref(dout), _ptr(d_features),
                                           _stream(features)), "lsr_latent_backward")
        return d_features, None, None, None, None, None, None, None, None


def get_scaled_size(scale: Fraction, size) -> tuple[int, ...]:
    """``ModelWrapper.get_scaled_size`` (model_wrapper.py:243-245): exact integer sizes only."""
    out = []
    for s in size:
        r = Fraction(scale) * s
        if r.denominator != 1:
            raise ValueError(f"{scale} * {s} is not an integer")
        out.append(int(r))
    return tuple(out)


def rescale(x: Tensor, scale_factor: Fraction) -> Tensor:
    """``ModelWrapper.rescale`` (model_wrapper.py:266-274): anti-aliased bilinear resize of the last
    two dimensions.  Downscaling (the training path, 1/supersampling, :374-379) runs on the fused HIP
    kernel; upscaling (``get_inv(scale_factor)``, video rendering only, :900-901) is the same ATen
    operator torchvision's ``resize`` dispatches to in the reference — stock PyTorch-ROCm, on the
    device, nothing to fuse it with."""
    batch_dims, spatial = x.shape[:-2], x.shape[-2:]
    size = get_scaled_size(scale_factor, spatial)
    if size[0] > spatial[0] or size[1] > spatial[1]:
        up = torch.nn.functional.interpolate(x.reshape(1, -1, *spatial), size=tuple(size), mode="bilinear",
                                             align_corners=False, antialias=True)
        return up.reshape(*batch_dims, *size)
    planes = x.reshape(1, -1, *spatial)
    _, z, _ = _LatentEpilogue.apply(planes, None, None, None, size, LOGVAR_FROM_MASK, False, False, (-30.0, 20.0))
    return z.reshape(*batch_dims, *size)


class LatentEpilogue(NamedTuple):
    latent_sample: Tensor            # (b, v, C, H, W)   feature_posterior.sample()
    z: Optional[Tensor]              # (b, v, C, h, w)   rescale(latent_sample, 1/supersampling)
    skip_z: Optional[Tensor]         # (b, v, 3+C, H, W) cat(color.detach(), latent_sample); None without colour
    logvar: Tensor                   # clamped posterior logvar, (b, v, 1|C, H, W)


def sample_rescale_skip(features: Tensor, mask: Optional[Tensor], color: Optional[Tensor] = None,
                        supersampling_factor: int | Fraction = 8, variational: bool = False,
                        noise: Optional[Tensor] = None, deterministic: bool = False,
                        logvar_interval: tuple[float, float] = (-30.0, 20.0)) -> LatentEpilogue:
    """features (b, v, C|2C, H, W) rendered feature map, mask (b, v, H, W), color (b, v, 3, H, W).

    ``noise`` defaults to ``torch.randn`` of the sample's shape drawn from the current generator —
    the same single draw ``DiagonalGaussianDistribution.sample`` makes; ``deterministic`` takes the
    posterior mean instead.  Gradients flow to ``features`` only (mask and colour are detached in
    the reference as well)."""
    b, v, F, H, W = features.shape
    Cc = F // 2 if variational else F
    if noise is None and not deterministic:
        noise = torch.randn((b, v, Cc, H, W), device=features.device, dtype=features.dtype)
    size = get_scaled_size(Fraction(1) / Fraction(supersampling_factor), (H, W))
    flat = lambda t, tail: None if t is None else t.reshape((b * v,) + tail)
    skip, z, logvar = _LatentEpilogue.apply(
        flat(features, (F, H, W)), None if variational else flat(mask.detach(), (H, W)), flat(noise, (Cc, H, W)),
        None if color is None else flat(color.detach(), (3, H, W)), size,
        LOGVAR_FROM_FEATURES if variational else LOGVAR_FROM_MASK, True, True, tuple(logvar_interval))
    cc = 0 if color is None else 3
    skip = skip.unflatten(0, (b, v))
    return LatentEpilogue(latent_sample=skip[:, :, cc:], z=z.unflatten(0, (b, v)),
                          skip_z=skip if color is not None else None, logvar=logvar.unflatten(0, (b, v)))


def decoder_output_epilogue(output: DecoderOutput, supersampling_factor: int | Fraction = 8,
                            noise: Optional[Tensor] = None) -> LatentEpilogue:
    """Convenience over a ``DecoderOutput`` of the non-variational decoder (the only configuration
    the reference's experiments use): its posterior mean is the rendered feature map."""
    return sample_rescale_skip(output.feature_posterior.mean, output.mask, output.color, supersampling_factor,
                               noise=noise)
