"""CPU oracle of the Gaussian adapter tail — TEST INFRASTRUCTURE ONLY (imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path).

A restatement, operation by operation, of what the reference computes for the geometry of its
Gaussians, written against torch tensors on the CPU so that autograd provides the backward:

  * /root/reference/src/model/encoder/common/gaussian_adapter.py:78-85   scale map
  * gaussian_adapter.py:116-127                                          get_scale_multiplier
  * gaussian_adapter.py:88                                               quaternion normalisation
  * /root/reference/src/model/encoder/common/gaussians.py:8-31           quaternion_to_matrix (xyzw)
  * gaussians.py:34-44                                                   build_covariance
  * gaussian_adapter.py:96-98                                            c2w rotation of the covariance
  * /root/reference/src/geometry/projection.py:74-114                    unproject / get_world_rays
  * gaussian_adapter.py:101-102                                          means

Parity status: PINNED — tests/test_adapter_cpu.py checks this file against
tests/golden/adapter_*.npz, which tests/golden/make_golden_adapter.py produced by importing and running
the reference's own GaussianAdapter (forward and autograd backward) in the build container.
"""
from __future__ import annotations

import torch


def scale_multiplier(intrinsics: torch.Tensor, height: int, width: int, multiplier: float = 0.1) -> torch.Tensor:
    pixel_size = 1 / torch.tensor((width, height), dtype=intrinsics.dtype)
    xy = multiplier * torch.einsum("...ij,j->...i", torch.linalg.inv(intrinsics[..., :2, :2]), pixel_size)
    return xy.sum(dim=-1)


def quaternion_to_matrix(q: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    i, j, k, r = torch.unbind(q, dim=-1)
    two_s = 2 / ((q * q).sum(dim=-1) + eps)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(*q.shape[:-1], 3, 3)


def adapter_geometry(extrinsics, intrinsics, coordinates, depths, raw_scales, raw_rotations, image_shape,
                     scale_min, scale_max, eps: float = 1e-8):
    """Shapes: extrinsics (cam,4,4), intrinsics (cam,3,3), coordinates (cam,rays,2),
    depths (cam,rays,samples), raw_scales (cam,rays,3), raw_rotations (cam,rays,4).
    Returns means (cam,rays,samples,3), covariances (cam,rays,samples,3,3),
    scales (cam,rays,samples,3), rotations (cam,rays,4)."""
    h, w = image_shape
    scales = scale_min + (scale_max - scale_min) * raw_scales.sigmoid()
    mult = scale_multiplier(intrinsics, h, w)                                     # (cam,)
    scales = scales[:, :, None, :] * depths[..., None] * mult[:, None, None, None]
    rotations = raw_rotations / (raw_rotations.norm(dim=-1, keepdim=True) + eps)
    R = quaternion_to_matrix(rotations)[:, :, None]                               # (cam,rays,1,3,3)
    S = torch.diag_embed(scales)
    cov = R @ S @ S.transpose(-1, -2) @ R.transpose(-1, -2)
    c2w = extrinsics[:, None, None, :3, :3]
    cov = c2w @ cov @ c2w.transpose(-1, -2)
    ones = torch.ones_like(coordinates[..., :1])
    d = torch.einsum("cij,crj->cri", torch.linalg.inv(intrinsics), torch.cat((coordinates, ones), -1))
    d = d / d.norm(dim=-1, keepdim=True)
    d = torch.einsum("cij,crj->cri", extrinsics[:, :3, :3], d)
    origins = extrinsics[:, None, None, :3, 3]
    means = origins + d[:, :, None, :] * depths[..., None]
    return means, cov, scales, rotations


def adapter_forward_backward(inputs: dict, grads: dict, dtype=torch.float64):
    """inputs: numpy arrays named as adapter_geometry's tensor arguments + image_shape, scale_min,
    scale_max; grads: upstream numpy gradients for any of means / covariances / scales / rotations.
    Returns (outputs dict, input-gradient dict) as numpy."""
    t = {k: torch.tensor(inputs[k], dtype=dtype) for k in
         ("extrinsics", "intrinsics", "coordinates", "depths", "raw_scales", "raw_rotations")}
    for k in ("coordinates", "depths", "raw_scales", "raw_rotations"):
        t[k].requires_grad_()
    out = adapter_geometry(t["extrinsics"], t["intrinsics"], t["coordinates"], t["depths"], t["raw_scales"],
                           t["raw_rotations"], tuple(int(x) for x in inputs["image_shape"]),
                           float(inputs["scale_min"]), float(inputs["scale_max"]))
    names = ("means", "covariances", "scales", "rotations")
    loss = sum((o * torch.tensor(grads[n], dtype=dtype)).sum() for n, o in zip(names, out) if grads.get(n) is not None)
    din = {}
    if torch.is_tensor(loss):
        loss.backward()
        din = {k: (torch.zeros_like(t[k]) if t[k].grad is None else t[k].grad).numpy()
               for k in ("coordinates", "depths", "raw_scales", "raw_rotations")}
    return {n: o.detach().numpy() for n, o in zip(names, out)}, din
