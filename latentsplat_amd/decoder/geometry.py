"""Camera / SH helpers on the rasterizer path (host side, stock PyTorch).

Behavioural counterparts of the reference helpers (paths relative to /root/reference):
  * ``get_fov``            — src/geometry/projection.py:233-247
  * ``homogenize_points``  — src/geometry/projection.py:9-13
  * ``eval_sh``            — src/misc/sh_utils.py:42-97 (e3nn axis convention, degree <= 4)
  * ``depth_to_relative_disparity`` — src/model/encoder/epipolar/conversions.py:17-27
"""
from __future__ import annotations

import torch
from torch import Tensor


def homogenize_points(points: Tensor) -> Tensor:
    """(..., d) -> (..., d+1) with a trailing 1."""
    return torch.nn.functional.pad(points, (0, 1), value=1.0)


def get_fov(intrinsics: Tensor) -> Tensor:
    """Field of view (radians) of normalised intrinsics (B,3,3) -> (B,2) as (fov_x, fov_y):
    the angle between the rays through the midpoints of opposite image edges."""
    inv = torch.linalg.inv(intrinsics)
    edge_mid = intrinsics.new_tensor([[0.0, 0.5, 1.0], [1.0, 0.5, 1.0], [0.5, 0.0, 1.0], [0.5, 1.0, 1.0]])
    rays = torch.einsum("bij,ej->bei", inv, edge_mid)          # (B,4,3): left, right, top, bottom
    rays = rays / rays.norm(dim=-1, keepdim=True)
    fov_x = (rays[:, 0] * rays[:, 1]).sum(-1).acos()
    fov_y = (rays[:, 2] * rays[:, 3]).sum(-1).acos()
    return torch.stack((fov_x, fov_y), dim=-1)


# Real SH constants (identical to the rasterizer's, lsr_sh.h) and, per band, the polynomial in
# the *reference's* axis naming.  Kept as a table of lambdas so the evaluation below is a single
# accumulate loop.
_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
       -0.4570457994644658, 1.445305721320277, -0.5900435899266435)
_C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892,
       0.10578554691520431, -0.6690465435572892, 0.47308734787878004, -1.7701307697799304,
       0.6258357354491761)


def sh_basis_e3nn(deg: int, dirs: Tensor) -> Tensor:
    """Basis values (..., (deg+1)^2) for unit ``dirs`` (..., 3) in the reference's convention
    (l=1 band is  -C1 x, +C1 y, -C1 z  — src/misc/sh_utils.py:62-65)."""
    assert 0 <= deg <= 4
    x, y, z = dirs.unbind(-1)
    one = torch.ones_like(x)
    terms = [_C0 * one]
    if deg >= 1:
        terms += [-_C1 * x, _C1 * y, -_C1 * z]
    if deg >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        terms += [_C2[0] * xz, _C2[1] * xy, _C2[2] * (2.0 * yy - zz - xx), _C2[3] * yz, _C2[4] * (zz - xx)]
    if deg >= 3:
        terms += [_C3[0] * x * (3 * zz - xx), _C3[1] * xz * y, _C3[2] * x * (4 * yy - zz - xx),
                  _C3[3] * y * (2 * yy - 3 * zz - 3 * xx), _C3[4] * z * (4 * yy - zz - xx),
                  _C3[5] * z * (zz - xx), _C3[6] * z * (zz - 3 * xx)]
    if deg >= 4:
        terms += [_C4[0] * xz * (zz - xx), _C4[1] * xy * (3 * zz - xx), _C4[2] * xz * (7 * yy - 1),
                  _C4[3] * xy * (7 * yy - 3), _C4[4] * (yy * (35 * yy - 30) + 3),
                  _C4[5] * yz * (7 * yy - 3), _C4[6] * (zz - xx) * (7 * yy - 1),
                  _C4[7] * yz * (zz - 3 * xx), _C4[8] * (zz * (zz - 3 * xx) - xx * (3 * zz - xx))]
    return torch.stack(terms, dim=-1)


def eval_sh(deg: int, sh: Tensor, dirs: Tensor) -> Tensor:
    """sh (..., C, >= (deg+1)^2), dirs (..., 3) unit vectors -> (..., C)."""
    n = (deg + 1) ** 2
    assert sh.shape[-1] >= n
    basis = sh_basis_e3nn(deg, dirs)                       # (..., n)
    return (sh[..., :n] * basis[..., None, :]).sum(-1)


def depth_to_relative_disparity(depth: Tensor, near: Tensor, far: Tensor, eps: float = 1e-10) -> Tensor:
    """0 at the near plane, 1 at the far plane, linear in disparity."""
    d_near, d_far, d = 1 / (near + eps), 1 / (far + eps), 1 / (depth + eps)
    return 1 - (d - d_far) / (d_near - d_far + eps)
