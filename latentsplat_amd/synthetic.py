"""Seeded synthetic scenes for tests and bench.

No datasets or checkpoints exist offline, so every workload is a random-init scene whose
statistics follow the reference encoder's output ranges:
  * camera: extrinsics = I, normalised K = [[.8,0,.5],[0,.8,.5],[0,0,1]], near .5 / far 40
    (reference config/dataset/co3d_hydrant.yaml:9), black background (co3d_hydrant.yaml:16);
  * means: pixel coordinate uniform in [-0.1,1.1]^2, depth log-uniform in [1.5 near, far/4];
  * covariance R diag(s^2) R^T, projected sigma log-uniform in [0.3,3] px, axis ratio .3-1
    (reference encoder/common/gaussian_adapter.py:78-85,116-127);
  * opacity U(0,1)/3 (reference encoder_epipolar.py:190).
All tensors are generated on CPU with a seeded generator (bit-identical on every machine).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

SEED = 1234
NEAR, FAR = 0.5, 40.0


@dataclass
class Scene:
    means: torch.Tensor          # (G,3) world
    covariances: torch.Tensor    # (G,3,3)
    opacities: torch.Tensor      # (G,)
    color_sh: torch.Tensor | None      # (G,3,K)
    feature_sh: torch.Tensor | None    # (G,C,Kf)
    extrinsics: torch.Tensor     # (V,4,4) camera-to-world
    intrinsics: torch.Tensor     # (V,3,3) normalised
    near: torch.Tensor           # (V,)
    far: torch.Tensor            # (V,)

    def to(self, device):
        f = lambda t: None if t is None else t.to(device)
        return Scene(*(f(getattr(self, k)) for k in self.__dataclass_fields__))


def _rand_rot(n, gen):
    q = torch.randn(n, 4, generator=gen)
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(n, 3, 3)


def _loguniform(n, lo, hi, gen):
    return torch.exp(torch.rand(n, generator=gen) * (math.log(hi) - math.log(lo)) + math.log(lo))


def make_scene(G: int, image_size: int = 256, views: int = 1, color_sh_degree: int | None = None,
               feature_channels: int | None = 4, feature_sh_degree: int = 0, seed: int = SEED,
               sigma_px=(0.3, 3.0), opacity_scale: float = 1.0 / 3.0) -> Scene:
    gen = torch.Generator().manual_seed(seed)
    fxn = 0.8
    u = torch.rand(G, generator=gen) * 1.2 - 0.1
    v = torch.rand(G, generator=gen) * 1.2 - 0.1
    z = _loguniform(G, 1.5 * NEAR, FAR / 4, gen)
    means = torch.stack([(u - 0.5) / fxn * z, (v - 0.5) / fxn * z, z], -1)
    s_major = _loguniform(G, sigma_px[0], sigma_px[1], gen) * z / (fxn * image_size)
    ratios = torch.rand(G, 2, generator=gen) * 0.7 + 0.3
    s = torch.stack([s_major, s_major * ratios[:, 0], s_major * ratios[:, 1]], -1)
    R = _rand_rot(G, gen)
    cov = R @ torch.diag_embed(s * s) @ R.transpose(1, 2)
    cov = 0.5 * (cov + cov.transpose(1, 2))
    opac = torch.rand(G, generator=gen) * opacity_scale
    color_sh = None
    if color_sh_degree is not None:
        K = (color_sh_degree + 1) ** 2
        # per-degree attenuation 0.1 * 0.25^deg (reference gaussian_adapter.py:44-61)
        att = torch.ones(K)
        for d in range(1, color_sh_degree + 1):
            att[d * d:(d + 1) ** 2] = 0.1 * 0.25 ** d
        color_sh = torch.randn(G, 3, K, generator=gen) * att
    feature_sh = None
    if feature_channels:
        Kf = (feature_sh_degree + 1) ** 2
        feature_sh = torch.randn(G, feature_channels, Kf, generator=gen) * 0.3
        if Kf > 1:
            att = torch.ones(Kf)
            for d in range(1, feature_sh_degree + 1):
                att[d * d:(d + 1) ** 2] = 0.1 * 0.25 ** d
            feature_sh = feature_sh * att
    extr = torch.eye(4).repeat(views, 1, 1)
    if views > 1:  # small random pose perturbations around the scene for views 1..V-1
        ang = (torch.rand(views, 3, generator=gen) - 0.5) * 0.3
        ang[0] = 0
        cx, sx = torch.cos(ang[:, 0]), torch.sin(ang[:, 0])
        cy, sy = torch.cos(ang[:, 1]), torch.sin(ang[:, 1])
        cz, sz = torch.cos(ang[:, 2]), torch.sin(ang[:, 2])
        one, zero = torch.ones(views), torch.zeros(views)
        Rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], -1).reshape(views, 3, 3)
        Ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], -1).reshape(views, 3, 3)
        Rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], -1).reshape(views, 3, 3)
        extr[:, :3, :3] = Rz @ Ry @ Rx
        tr = (torch.rand(views, 3, generator=gen) - 0.5) * 0.6
        tr[0] = 0
        extr[:, :3, 3] = tr
    intr = torch.tensor([[fxn, 0, 0.5], [0, fxn, 0.5], [0, 0, 1.0]]).repeat(views, 1, 1)
    return Scene(means.float(), cov.float(), opac.float(), color_sh, feature_sh, extr.float(),
                 intr.float(), torch.full((views,), NEAR), torch.full((views,), FAR))


def _quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    """(…,4) quaternions xyzw, as the reference's build_covariance reads them (src/model/encoder/common/gaussians.py:8-44)."""
    x, y, z, w = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(*q.shape[:-1], 3, 3)


def make_encoder_scene(context_views: int = 2, size: int = 256, samples: int = 3, views: int = 4,
                       color_sh_degree: int | None = 4, feature_channels: int | None = 4, feature_sh_degree: int = 2,
                       seed: int = SEED, scale_min: float = 0.5, scale_max: float = 15.0) -> Scene:
    """The Gaussian distribution the reference's encoder hands to the decoder (BASELINE configs[3] / [4]): PIXEL-ALIGNED, one
    ray per pixel of every context view and `samples` Gaussians along it — context_views x size^2 x samples of them
    (2 x 256^2 x 3 = 393 216), in (view, ray, sample) order, i.e. spatially coherent in memory, unlike make_scene's cloud.

    Follows /root/reference/src/model/encoder/encoder_epipolar.py:184-236 and encoder/common/gaussian_adapter.py:75-114 with
    random-init network outputs in their place: ray through the pixel centre plus a sigmoid offset of up to half a pixel
    (encoder_epipolar.py:180-183); depths between near and far (log-uniform, a draw per sample); scales
    `(min + (max - min) sigmoid(n)) * depth * 0.1 * sum(K^-1 pixel_size)` (gaussian_adapter.py:78-85,116-127: 0.1 ... 3 px
    projected); unit quaternions; covariance R_c2w R S S^T R^T R_c2w^T (:97-99); mean = origin + direction * depth (:101-102);
    opacity U(0,1) / samples (encoder_epipolar.py:190: map_pdf_to_opacity(...) / gpp); harmonics masked per degree
    (gaussian_adapter.py:44-61).  `views` target cameras are small perturbations of the first context camera, like
    make_scene's.  CPU tensors from a seeded generator (bit-identical everywhere)."""
    gen = torch.Generator().manual_seed(seed)
    fxn = 0.8
    K = torch.tensor([[fxn, 0, 0.5], [0, fxn, 0.5], [0, 0, 1.0]])
    rays = size * size
    # context cameras: the first at the origin, the others shifted / turned a little (a stereo-like context pair)
    ctx = torch.eye(4).repeat(context_views, 1, 1)
    for c in range(1, context_views):
        ang = 0.12 * c
        ctx[c, :3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
        ctx[c, :3, 3] = torch.tensor([0.35 * c, 0.02 * c, 0.0])
    ys, xs = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    centre = (torch.stack([xs, ys], -1).reshape(rays, 2).float() + 0.5) / size                       # (r, 2) in [0, 1]
    offset = torch.sigmoid(torch.randn(context_views, rays, 2, generator=gen)) - 0.5
    coords = centre[None] + offset / size                                                             # (v, r, 2)
    depth = _loguniform(context_views * rays * samples, 1.5 * NEAR, FAR / 4, gen).reshape(context_views, rays, samples)
    # rays: direction = K^-1 (x, y, 1), normalised; world = R_c2w direction (src/geometry/projection.py:74-114)
    d_cam = torch.cat([(coords - 0.5) / fxn, torch.ones(context_views, rays, 1)], -1)
    d_cam = d_cam / d_cam.norm(dim=-1, keepdim=True)
    R = ctx[:, None, :3, :3]                                                                          # (v, 1, 3, 3)
    d_world = (R @ d_cam[..., None])[..., 0]
    means = ctx[:, None, None, :3, 3] + d_world[:, :, None, :] * depth[..., None]                     # (v, r, s, 3)
    multiplier = 0.1 * (1.0 / fxn / size + 1.0 / fxn / size)
    s_raw = scale_min + (scale_max - scale_min) * torch.sigmoid(torch.randn(context_views, rays, 3, generator=gen))
    scales = s_raw[:, :, None, :] * depth[..., None] * multiplier                                     # (v, r, s, 3)
    q = torch.randn(context_views, rays, 4, generator=gen)
    q = q / (q.norm(dim=-1, keepdim=True) + 1e-8)
    Rq = _quat_to_rot(q)[:, :, None]                                                                   # (v, r, 1, 3, 3)
    M = R[:, :, None] @ Rq @ torch.diag_embed(scales)
    cov = M @ M.transpose(-1, -2)
    cov = 0.5 * (cov + cov.transpose(-1, -2))
    G = context_views * rays * samples
    opac = torch.rand(G, generator=gen) / samples
    color_sh = None
    if color_sh_degree is not None:
        Kc = (color_sh_degree + 1) ** 2
        att = torch.ones(Kc)
        for dgr in range(1, color_sh_degree + 1):
            att[dgr * dgr:(dgr + 1) ** 2] = 0.1 * 0.25 ** dgr
        color_sh = torch.randn(G, 3, Kc, generator=gen) * att
    feature_sh = None
    if feature_channels:
        Kf = (feature_sh_degree + 1) ** 2
        att = torch.ones(Kf)
        for dgr in range(1, feature_sh_degree + 1):
            att[dgr * dgr:(dgr + 1) ** 2] = 0.1 * 0.25 ** dgr
        feature_sh = torch.randn(G, feature_channels, Kf, generator=gen) * 0.3 * att
    tgt = make_scene(1, image_size=size, views=views, color_sh_degree=None, feature_channels=None, seed=seed + 1)
    return Scene(means.reshape(G, 3).float(), cov.reshape(G, 3, 3).float(), opac.float(), color_sh, feature_sh,
                 tgt.extrinsics, tgt.intrinsics, tgt.near, tgt.far)
