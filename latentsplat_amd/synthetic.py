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
