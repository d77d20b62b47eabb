"""Mirror of the reference's ``GaussianAdapter``
(/root/reference/src/model/encoder/common/gaussian_adapter.py:13-139) whose geometry —
scale map, quaternion → rotation, covariance, camera-to-world rotation, ray unprojection, means
(:78-102; gaussians.py:8-44; src/geometry/projection.py:74-114) — runs as ONE hand-written HIP
kernel forward and one backward (csrc/adapter.hip, C ABI include/lsr_adapter.h) instead of ~45
PyTorch ops each way.

Same names, argument meaning and return type as the reference.  ROCm tensors only — there is no
CPU fallback.  ``rotate_sh`` (e3nn Wigner-D, gaussian_adapter.py:107-108) is not part of the
kernel: it is taken from e3nn when that is installed, or injected through ``rotate_sh=``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from math import isqrt, prod
from typing import Callable, Optional

import torch
from torch import Tensor, nn

from . import _lib
from ._lib import AdapterDims, AdapterInGrads, AdapterInputs, AdapterOutGrads, AdapterOutputs


@dataclass
class Gaussians:                      # gaussian_adapter.py:13-21
    means: Tensor
    covariances: Tensor
    scales: Tensor
    rotations: Tensor
    color_harmonics: Tensor
    feature_harmonics: Tensor
    opacities: Tensor


@dataclass
class GaussianAdapterCfg:             # gaussian_adapter.py:24-29
    gaussian_scale_min: float
    gaussian_scale_max: float
    color_sh_degree: int
    feature_sh_degree: int


def _ptr(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t: Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _row_stride(t: Tensor) -> Optional[int]:
    """Element stride between consecutive rows when ``t`` (..., n) can be walked as a flat list of
    rows with one constant stride and unit inner stride; None if it cannot."""
    if t.stride(-1) != 1 and t.shape[-1] > 1:
        return None
    dims = [(s, st) for s, st in zip(t.shape[:-1], t.stride()[:-1]) if s != 1]
    if not dims:
        return t.shape[-1]
    for (_, st_outer), (s_inner, st_inner) in zip(dims[:-1], dims[1:]):
        if st_outer != st_inner * s_inner:
            return None
    return dims[-1][1]


class _AdapterGeometry(torch.autograd.Function):
    """(extrinsics (cam,4,4), intrinsics (cam,3,3), coordinates (cam,rays,2), depths
    (cam,rays,S), raw (cam,rays,>=7 view)) -> means, covariances, scales, rotations."""

    @staticmethod
    def forward(ctx, extrinsics, intrinsics, coordinates, depths, raw, height, width, scale_min, scale_max,
                eps, cov_elems):
        for t in (extrinsics, intrinsics, coordinates, depths, raw):
            if not t.is_cuda or t.dtype != torch.float32:
                raise _lib.LsrError("GaussianAdapter geometry needs float32 ROCm tensors (no CPU fallback)")
        lib = _lib.load()
        cams, rays, samples = depths.shape
        extrinsics, intrinsics = extrinsics.contiguous(), intrinsics.contiguous()
        coordinates, depths = coordinates.contiguous(), depths.contiguous()
        stride = _row_stride(raw)
        if stride is None or stride < 7:
            raw = raw[..., :7].contiguous()
            stride = 7
        dims = AdapterDims(cams, rays, samples, height, width, cov_elems, scale_min, scale_max, eps, stride, 0, 0)
        dev = depths.device
        means = torch.empty((cams, rays, samples, 3), device=dev)
        cov = torch.empty((cams, rays, samples) + ((3, 3) if cov_elems == 9 else (6,)), device=dev)
        scales = torch.empty((cams, rays, samples, 3), device=dev)
        rotations = torch.empty((cams, rays, 4), device=dev)
        inp = AdapterInputs(_ptr(extrinsics), _ptr(intrinsics), _ptr(coordinates), _ptr(depths), _ptr(raw))
        out = AdapterOutputs(_ptr(means), _ptr(cov), _ptr(scales), _ptr(rotations))
        _lib.check(lib.lsr_adapter_forward(C.byref(dims), C.byref(inp), C.byref(out), _stream(depths)),
                   "lsr_adapter_forward")
        ctx.save_for_backward(extrinsics, intrinsics, coordinates, depths, raw)
        ctx.dims = dims
        ctx.set_materialize_grads(False)
        return means, cov, scales, rotations

    @staticmethod
    def backward(ctx, g_means, g_cov, g_scales, g_rot):
        extrinsics, intrinsics, coordinates, depths, raw = ctx.saved_tensors
        dims = ctx.dims
        lib = _lib.load()
        dev = depths.device
        cams, rays, samples = depths.shape
        g_means = torch.zeros((cams, rays, samples, 3), device=dev) if g_means is None else g_means.contiguous()
        g_cov = (torch.zeros((cams, rays, samples, dims.cov_elems), device=dev) if g_cov is None
                 else g_cov.contiguous())
        g_scales = None if g_scales is None else g_scales.contiguous()
        g_rot = None if g_rot is None else g_rot.contiguous()
        d_coord = torch.empty((cams, rays, 2), device=dev)
        d_depth = torch.empty((cams, rays, samples), device=dev)
        d_raw = torch.empty((cams, rays, 7), device=dev)
        inp = AdapterInputs(_ptr(extrinsics), _ptr(intrinsics), _ptr(coordinates), _ptr(depths), _ptr(raw))
        dout = AdapterOutGrads(_ptr(g_means), _ptr(g_cov), _ptr(g_scales), _ptr(g_rot))
        din = AdapterInGrads(_ptr(d_coord), _ptr(d_depth), _ptr(d_raw))
        _lib.check(lib.lsr_adapter_backward(C.byref(dims), C.byref(inp), C.byref(dout), C.byref(din),
                                            _stream(depths)), "lsr_adapter_backward")
        return None, None, d_coord, d_depth, d_raw, None, None, None, None, None, None


def adapter_geometry(extrinsics: Tensor, intrinsics: Tensor, coordinates: Tensor, depths: Tensor, raw: Tensor,
                     image_shape: tuple[int, int], scale_min: float, scale_max: float, eps: float = 1e-8,
                     packed_covariance: bool = False):
    """Functional entry.  extrinsics (cam,4,4) camera-to-world, intrinsics (cam,3,3) normalised,
    coordinates (cam,rays,2), depths (cam,rays,samples), raw (cam,rays,>=7): columns 0..2 raw
    scales, 3..6 raw quaternion xyzw (may be a strided view of a wider matrix).
    Returns means (cam,rays,S,3), covariances (cam,rays,S,3,3) — or (cam,rays,S,6) packed upper
    triangle, the layout the rasterizer consumes — scales (cam,rays,S,3), rotations (cam,rays,4)."""
    h, w = image_shape
    return _AdapterGeometry.apply(extrinsics, intrinsics, coordinates, depths, raw[..., :7], int(h), int(w),
                                  float(scale_min), float(scale_max), float(eps), 6 if packed_covariance else 9)


def _e3nn_rotate_sh(sh_coefficients: Tensor, rotations: Tensor) -> Tensor:
    """SH coefficient rotation by per-degree Wigner-D matrices (what gaussian_adapter.py:107-108
    obtains from src/misc/sh_utils.py:100-120); requires e3nn."""
    try:
        from e3nn.o3 import matrix_to_angles, wigner_D
    except ImportError as e:  # pragma: no cover - e3nn is absent from the build image
        raise _lib.LsrError("rotate_sh needs e3nn; install it or pass rotate_sh= to GaussianAdapter") from e
    n = sh_coefficients.shape[-1]
    angles = matrix_to_angles(rotations)
    bands = []
    for degree in range(isqrt(n)):
        with torch.device(sh_coefficients.device):
            D = wigner_D(degree, *angles).to(sh_coefficients.dtype)
        band = sh_coefficients[..., degree * degree:(degree + 1) * (degree + 1)]
        bands.append((D @ band[..., None])[..., 0])
    return torch.cat(bands, dim=-1)


class GaussianAdapter(nn.Module):     # gaussian_adapter.py:32-139
    cfg: GaussianAdapterCfg

    def __init__(self, cfg: GaussianAdapterCfg, n_feature_channels: int,
                 rotate_sh: Optional[Callable[[Tensor, Tensor], Tensor]] = None):
        super().__init__()
        self.cfg = cfg
        self.n_feature_channels = n_feature_channels
        self._rotate_sh = rotate_sh or _e3nn_rotate_sh
        for name, degree, size in (("color_sh_mask", cfg.color_sh_degree, self.d_color_sh),
                                   ("feature_sh_mask", cfg.feature_sh_degree, self.d_feature_sh)):
            mask = torch.ones((size,), dtype=torch.float32)
            for deg in range(1, degree + 1):
                mask[deg ** 2:(deg + 1) ** 2] = 0.1 * 0.25 ** deg
            self.register_buffer(name, mask, persistent=False)

    @property
    def d_color_sh(self) -> int:
        return (self.cfg.color_sh_degree + 1) ** 2

    @property
    def d_feature_sh(self) -> int:
        return (self.cfg.feature_sh_degree + 1) ** 2

    @property
    def d_in(self) -> int:
        return 7 + 3 * self.d_color_sh + self.n_feature_channels * self.d_feature_sh

    def get_scale_multiplier(self, intrinsics: Tensor, pixel_size: Tensor, multiplier: float = 0.1) -> Tensor:
        xy = multiplier * torch.einsum("...ij,j->...i", intrinsics[..., :2, :2].inverse(), pixel_size)
        return xy.sum(dim=-1)

    def forward(self, extrinsics: Tensor, intrinsics: Tensor, coordinates: Tensor, depths: Tensor,
                opacities: Tensor, raw_gaussians: Tensor, image_shape: tuple[int, int], eps: float = 1e-8
                ) -> Gaussians:
        batch = tuple(opacities.shape)
        nb = len(batch)

        def padded(t: Tensor, trailing: int):
            shape = tuple(t.shape[:t.dim() - trailing])
            return (1,) * (nb - len(shape)) + shape

        eb, ib = padded(extrinsics, 2), padded(intrinsics, 2)
        cb, rb = padded(coordinates, 1), padded(raw_gaussians, 1)
        # cameras vary over a leading prefix of the batch dims only (the encoder's (b, v))
        p = nb
        while p > 0 and eb[p - 1] == 1 and ib[p - 1] == 1:
            p -= 1
        shared_over_samples = nb > p and cb[-1] == 1 and rb[-1] == 1
        samples = batch[-1] if shared_over_samples else 1
        ray_dims = batch[p:nb - 1] if shared_over_samples else batch[p:]
        cams, rays = prod(batch[:p]), prod(ray_dims)
        row_batch = batch[:p] + tuple(ray_dims)

        def rows(t: Tensor, pad: tuple, width: int):
            t = t.reshape(pad + (width,))
            if shared_over_samples:
                t = t.squeeze(-2)
            return t.expand(row_batch + (width,)).reshape(cams, rays, width)

        ext = extrinsics.reshape(eb + (4, 4)).expand(batch[:p] + (1,) * (nb - p) + (4, 4)).reshape(cams, 4, 4)
        itr = intrinsics.reshape(ib + (3, 3)).expand(batch[:p] + (1,) * (nb - p) + (3, 3)).reshape(cams, 3, 3)
        coords = rows(coordinates, cb, 2)
        raw7 = rows(raw_gaussians[..., :7], rb, 7)
        dep = depths.expand(batch).reshape(cams, rays, samples)
        means, cov, scales, rot = adapter_geometry(ext, itr, coords, dep, raw7, image_shape,
                                                   self.cfg.gaussian_scale_min, self.cfg.gaussian_scale_max, eps)

        color_sh, feature_sh = raw_gaussians[..., 7:].split(
            (3 * self.d_color_sh, self.n_feature_channels * self.d_feature_sh), dim=-1)
        color_sh = color_sh.reshape(*color_sh.shape[:-1], 3, self.d_color_sh)
        feature_sh = feature_sh.reshape(*feature_sh.shape[:-1], self.n_feature_channels, self.d_feature_sh)
        color_sh = color_sh.broadcast_to((*batch, 3, self.d_color_sh)) * self.color_sh_mask
        feature_sh = feature_sh.broadcast_to((*batch, self.n_feature_channels, self.d_feature_sh)) * self.feature_sh_mask
        c2w = extrinsics[..., :3, :3]
        rot = rot.reshape(row_batch + ((1,) if shared_over_samples else ()) + (4,))
        return Gaussians(
            means=means.reshape(batch + (3,)),
            covariances=cov.reshape(batch + (3, 3)),
            color_harmonics=self._rotate_sh(color_sh, c2w[..., None, :, :]),
            feature_harmonics=self._rotate_sh(feature_sh, c2w[..., None, :, :]),
            opacities=opacities,
            scales=scales.reshape(batch + (3,)),
            rotations=rot.broadcast_to(batch + (4,)),
        )
