"""Python half of the splatting path: camera setup, scale invariance, SH handling, and the call
into the MI355X rasterizer.  Same public names, arguments and return types as the reference's
/root/reference/src/model/decoder/cuda_splatting.py (``get_projection_matrix`` :19-46,
``RenderOutput`` :49-54, ``render_cuda`` :56-167, ``render_cuda_orthographic`` :170-292,
``render_depth_cuda`` :298-340), re-designed so that

  * all views of a call go to the device in ONE batched rasterizer launch sequence
    (the reference loops ``for i in range(b)`` with two ``.item()`` syncs per view, :124-162);
  * nothing is synchronised with the host here (tan(fov) stays on the device);
  * ``render_scenes`` (new) takes scene-major inputs so Gaussians are NOT replicated per view
    (the reference ``repeat``s every tensor v times, decoder_splatting_cuda.py:71-87);
  * the per-view pre-pass the reference runs in PyTorch — ``1/near`` scaling of means and
    covariances (:75-82), packing the covariance triangle (:148,157), the ``(g,3,n)->(g,n,3)``
    copy of the colour SH (:91) and the latent-feature SH evaluation (:94-101) — happens inside
    the kernels (scene scale in the view table, 3x3 covariances, channel-major SH, ``feature_sh``).
    ``_scale_scene`` / ``_payload`` / ``_pack_covariances`` remain as the host-side statement of
    that math (fallback for SH shapes the kernel does not fuse, and what the tests pin against
    the reference).
"""
from __future__ import annotations

from dataclasses import dataclass
from math import isqrt
from typing import Literal, Optional

import torch
from torch import Tensor

from ..rasterizer import build_view_table, fused_feature_sh_supported, make_view_table, rasterize_views
from .geometry import depth_to_relative_disparity, eval_sh, get_fov, homogenize_points


def get_projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> Tensor:
    """(B,) x4 -> (B,4,4) perspective matrix: x,y to (-1,1), z to (0,1), +z forward (no flip)."""
    tan_x, tan_y = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
    right, top = tan_x * near, tan_y * near
    left, bottom = -right, -top
    P = near.new_zeros((near.shape[0], 4, 4), dtype=torch.float32)
    P[:, 0, 0] = 2 * near / (right - left)
    P[:, 1, 1] = 2 * near / (top - bottom)
    P[:, 0, 2] = (right + left) / (right - left)
    P[:, 1, 2] = (top + bottom) / (top - bottom)
    P[:, 2, 2] = far / (far - near)
    P[:, 2, 3] = -(far * near) / (far - near)
    P[:, 3, 2] = 1
    return P


@dataclass
class RenderOutput:
    color: Optional[Tensor]    # (batch, 3, h, w)
    feature: Optional[Tensor]  # (batch, channels, h, w)
    mask: Tensor               # (batch, h, w)
    depth: Tensor              # (batch, h, w)


@dataclass
class _Cameras:
    view_matrix: Tensor      # (B,4,4) row-major memory = transposed world->view
    full_projection: Tensor  # (B,4,4) transposed world->clip
    campos: Tensor           # (B,3)
    tan_fov_x: Tensor        # (B,)
    tan_fov_y: Tensor        # (B,)


def _cameras(extrinsics: Tensor, near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> _Cameras:
    proj_t = get_projection_matrix(near, far, fov_x, fov_y).transpose(1, 2)
    view_t = torch.linalg.inv(extrinsics).transpose(1, 2)
    return _Cameras(view_t, view_t @ proj_t, extrinsics[:, :3, 3], (0.5 * fov_x).tan(), (0.5 * fov_y).tan())


def _pack_covariances(cov: Tensor) -> Tensor:
    """(..., 3, 3) -> (..., 6) upper triangle xx,xy,xz,yy,yz,zz (reference :148,157)."""
    return torch.stack([cov[..., 0, 0], cov[..., 0, 1], cov[..., 0, 2],
                        cov[..., 1, 1], cov[..., 1, 2], cov[..., 2, 2]], dim=-1)


def _payload(means: Tensor, campos: Tensor, color_sh: Optional[Tensor], feature_sh: Optional[Tensor],
             use_sh: bool):
    """Colour / feature inputs of the rasterizer.
    means (B,G,3) per view; campos (B,3); color_sh (B|1,G,3,d); feature_sh (B|1,G,C,d).
    Returns (sh_degree, shs (B|1,G,d,3)|None, colors_precomp|None, features (B,G,C)|None)."""
    degree, shs, colors_precomp, features = 0, None, None, None
    if use_sh:
        if color_sh is not None:
            degree = isqrt(color_sh.shape[-1]) - 1
            shs = color_sh.transpose(-1, -2).contiguous()
        if feature_sh is not None:
            # latent features are view dependent: evaluated here, composited as plain channels
            direction = means - campos[:, None]
            direction = direction / direction.norm(dim=-1, keepdim=True)
            features = 0.5 + eval_sh(isqrt(feature_sh.shape[-1]) - 1, feature_sh, direction)
    else:
        if color_sh is not None:
            colors_precomp = color_sh[..., 0]
        if feature_sh is not None:
            features = feature_sh[..., 0]
    return degree, shs, colors_precomp, features


def _squeeze_shared(t: Optional[Tensor]) -> Optional[Tensor]:
    """(1,G,...) -> (G,...) so the rasterizer treats it as shared by all views.  ``squeeze`` rather than
    ``t[0]``: the backward of a select allocates a zero tensor of the full shape and copies the gradient
    into it (a 174 MB fill + copy for the colour harmonics at configs[3]); squeeze's is a view."""
    return t.squeeze(0) if (t is not None and t.shape[0] == 1) else t


def _view_table(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, background: Tensor,
                scale_invariant: bool) -> Tensor:
    """(B,44) camera table + scene scale.  On the MI355X: one kernel; for host tensors (the
    not-gpu tests, which substitute the rasterizer) the same math in PyTorch."""
    if extrinsics.is_cuda:
        return build_view_table(extrinsics, intrinsics, near, far, background, scale_invariant)
    cams, scale = _scaled_cameras(extrinsics, intrinsics, near, far, scale_invariant)
    return make_view_table(cams.view_matrix, cams.full_projection, cams.campos, cams.tan_fov_x,
                           cams.tan_fov_y, background, scale)


def _render_views(views: Tensor, image_shape, means: Tensor, covariances: Tensor, opacities: Tensor,
                  color_sh, feature_sh, use_sh: bool) -> RenderOutput:
    """``views`` (B,44); UNSCALED means (S,G,3) / covariances (S,G,3,3); opacities (S,G);
    *_sh (S,G,.,.) with S = 1 (shared), B (one slice per view) or a divisor of B (scenes of B/S
    consecutive views each)."""
    h, w = image_shape
    degree, kw = 0, {}
    if use_sh:
        if color_sh is not None:
            degree = isqrt(color_sh.shape[-1]) - 1
            kw.update(shs=_squeeze_shared(color_sh), shs_channel_major=True)   # stored (.., 3, n) layout
        if feature_sh is not None:
            if fused_feature_sh_supported(feature_sh):
                kw.update(feature_sh=_squeeze_shared(feature_sh))
            else:   # evaluate on the host like the reference does
                scaled = means * views[:, 40, None, None]
                kw.update(features=_payload(scaled, views[:, 32:35], None, feature_sh, True)[3])
    else:
        if color_sh is not None:
            kw.update(colors_precomp=_squeeze_shared(color_sh[..., 0]))
        if feature_sh is not None:
            kw.update(features=_squeeze_shared(feature_sh[..., 0]))
    color, feature, mask, depth, _ = rasterize_views(
        views, h, w, degree, _squeeze_shared(means), _squeeze_shared(covariances),
        _squeeze_shared(opacities[..., None]), **kw)
    return RenderOutput(color, feature, mask, depth)


def _scaled_cameras(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, scale_invariant: bool):
    """Cameras of the scene rescaled so that near == 1 (reference :75-82), plus that scale."""
    scale = None
    if scale_invariant:
        scale = 1 / near
        extrinsics = extrinsics.clone()
        extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * scale[:, None]
        near, far = near * scale, far * scale
    fov_x, fov_y = get_fov(intrinsics).unbind(dim=-1)
    return _cameras(extrinsics, near, far, fov_x, fov_y), scale


def _scale_scene(extrinsics: Tensor, near: Tensor, far: Tensor, means: Tensor, covariances: Tensor):
    """Scale-invariant rendering: rescale the scene so that near == 1 (reference :75-82)."""
    scale = 1 / near
    extrinsics = extrinsics.clone()
    extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * scale[:, None]
    return (extrinsics, near * scale, far * scale, means * scale[:, None, None],
            covariances * (scale[:, None, None, None] ** 2))


def render_cuda(
    extrinsics: Tensor,             # (batch,4,4) camera-to-world
    intrinsics: Tensor,             # (batch,3,3) normalised
    near: Tensor,                   # (batch,)
    far: Tensor,                    # (batch,)
    image_shape: tuple[int, int],
    background_color: Tensor,       # (batch,3)
    gaussian_means: Tensor,         # (batch,gaussian,3)
    gaussian_covariances: Tensor,   # (batch,gaussian,3,3)
    gaussian_opacities: Tensor,     # (batch,gaussian)
    gaussian_color_sh_coefficients: Optional[Tensor] = None,    # (batch,gaussian,3,d_color_sh)
    gaussian_feature_sh_coefficients: Optional[Tensor] = None,  # (batch,gaussian,channels,d_feature_sh)
    scale_invariant: bool = True,
    use_sh: bool = True,
) -> RenderOutput:
    assert gaussian_color_sh_coefficients is not None or gaussian_feature_sh_coefficients is not None
    assert use_sh or gaussian_color_sh_coefficients.shape[-1] == 1
    views = _view_table(extrinsics, intrinsics, near, far, background_color, scale_invariant)
    return _render_views(views, image_shape, gaussian_means, gaussian_covariances, gaussian_opacities,
                         gaussian_color_sh_coefficients, gaussian_feature_sh_coefficients, use_sh)


def render_scenes(
    extrinsics: Tensor,             # (b,v,4,4)
    intrinsics: Tensor,             # (b,v,3,3)
    near: Tensor,                   # (b,v)
    far: Tensor,                    # (b,v)
    image_shape: tuple[int, int],
    background_color: Tensor,       # (3,)
    gaussian_means: Tensor,         # (b,g,3)      -- NOT replicated per view
    gaussian_covariances: Tensor,   # (b,g,3,3)
    gaussian_opacities: Tensor,     # (b,g)
    gaussian_color_sh_coefficients: Optional[Tensor] = None,    # (b,g,3,d)
    gaussian_feature_sh_coefficients: Optional[Tensor] = None,  # (b,g,c,d)
    scale_invariant: bool = True,
    use_sh: bool = True,
) -> RenderOutput:
    """Scene-major entry point: identical results to ``render_cuda`` on the v-fold replicated
    inputs, but colour SH / opacities of a scene are read once for all of its views.
    Returns tensors flattened over (b v) like ``render_cuda``."""
    assert gaussian_color_sh_coefficients is not None or gaussian_feature_sh_coefficients is not None
    b, v = extrinsics.shape[:2]
    outs = []
    # one camera-table launch for all b*v views
    views = _view_table(extrinsics.flatten(0, 1), intrinsics.flatten(0, 1), near.flatten(0, 1), far.flatten(0, 1),
                        background_color, scale_invariant)
    fsh_all = gaussian_feature_sh_coefficients
    if b == 1 or not use_sh or fsh_all is None or fused_feature_sh_supported(fsh_all):
        # ONE call for all b*v views: the b scenes are view groups of v views each (inputs keep their
        # leading scene dimension, nothing is replicated or concatenated)
        return _render_views(views, image_shape, gaussian_means, gaussian_covariances, gaussian_opacities,
                             gaussian_color_sh_coefficients, fsh_all, use_sh)
    for s in range(b):   # latent SH evaluated on the host (degree > 2 / too many coefficients): per scene
        csh = None if gaussian_color_sh_coefficients is None else gaussian_color_sh_coefficients[s][None]
        fsh = fsh_all[s][None]
        outs.append(_render_views(views[s * v:(s + 1) * v], image_shape, gaussian_means[s][None],
                                  gaussian_covariances[s][None], gaussian_opacities[s][None], csh, fsh, use_sh))
    cat = lambda xs: None if xs[0] is None else torch.cat(xs, dim=0)
    return RenderOutput(cat([o.color for o in outs]), cat([o.feature for o in outs]),
                        cat([o.mask for o in outs]), cat([o.depth for o in outs]))


def render_cuda_orthographic(
    extrinsics: Tensor,             # (batch,4,4)
    width: Tensor,                  # (batch,)
    height: Tensor,                 # (batch,)
    near: Tensor,
    far: Tensor,
    image_shape: tuple[int, int],
    background_features: Tensor,    # (batch,3)
    gaussian_means: Tensor,
    gaussian_covariances: Tensor,
    gaussian_opacities: Tensor,
    gaussian_color_sh_coefficients: Optional[Tensor] = None,
    gaussian_feature_sh_coefficients: Optional[Tensor] = None,
    fov_degrees: float = 0.1,
    use_sh: bool = True,
    dump: Optional[dict] = None,
) -> RenderOutput:
    """Fake orthographic camera: tiny field of view, camera moved back so the near plane spans
    ``width`` x ``height`` world units (reference :219-228). SH directions use the ORIGINAL camera
    position, as the reference does (:199-206 run before the move-back)."""
    b = extrinsics.shape[0]
    campos_for_sh = extrinsics[:, :3, 3]
    fov_x = torch.tensor(fov_degrees, device=extrinsics.device).deg2rad()
    tan_x = (0.5 * fov_x).tan()
    distance_to_near = (0.5 * width) / tan_x
    tan_y = 0.5 * height / distance_to_near
    fov_y = (2 * tan_y).atan()
    near, far = near + distance_to_near, far + distance_to_near
    move_back = torch.eye(4, dtype=torch.float32, device=extrinsics.device).repeat(b, 1, 1)
    move_back[:, 2, 3] = -distance_to_near
    extrinsics = extrinsics @ move_back
    if dump is not None:
        dump.update(extrinsics=extrinsics, fov_x=fov_x, fov_y=fov_y, near=near, far=far)
    cams = _cameras(extrinsics, near, far, fov_x.expand(b), fov_y)
    # the reference hands tan(fov_x) / tan(fov_y) (not tan of the half angle of fov_y's atan) to
    # the rasterizer exactly as computed above (:260-261)
    cams.tan_fov_x, cams.tan_fov_y = tan_x.expand(b), tan_y
    h, w = image_shape
    degree, shs, colors_precomp, features = _payload(gaussian_means, campos_for_sh,
                                                     gaussian_color_sh_coefficients,
                                                     gaussian_feature_sh_coefficients, use_sh)
    views = make_view_table(cams.view_matrix, cams.full_projection, cams.campos, cams.tan_fov_x,
                            cams.tan_fov_y, background_features)
    color, feature, mask, depth, _ = rasterize_views(
        views, h, w, degree, gaussian_means, _pack_covariances(gaussian_covariances),
        gaussian_opacities[..., None], shs=shs, colors_precomp=colors_precomp, features=features)
    return RenderOutput(color, feature, mask, depth)


DepthRenderingMode = Literal["depth", "disparity", "relative_disparity", "log"]


def _depth_as_color(extrinsics: Tensor, near: Tensor, far: Tensor, means: Tensor, mode: str) -> Tensor:
    """Per-Gaussian camera-space z (or a monotone function of it) (reference :310-323)."""
    cam = torch.einsum("bij,bgj->bgi", torch.linalg.inv(extrinsics), homogenize_points(means))
    value = cam[..., 2]
    if mode == "disparity":
        value = 1 / value
    elif mode == "relative_disparity":
        value = depth_to_relative_disparity(value, near[:, None], far[:, None])
    elif mode == "log":
        value = value.minimum(near[:, None]).maximum(far[:, None]).log()
    return value


def render_depth_cuda(
    extrinsics: Tensor,
    intrinsics: Tensor,
    near: Tensor,
    far: Tensor,
    image_shape: tuple[int, int],
    gaussian_means: Tensor,
    gaussian_covariances: Tensor,
    gaussian_opacities: Tensor,
    scale_invariant: bool = True,
    mode: DepthRenderingMode = "depth",
) -> Tensor:
    """Alpha-blended depth: the per-Gaussian value is rendered as a grey colour through the
    degree-0 SH path on a black background; returns the channel mean (batch,h,w)."""
    fake = _depth_as_color(extrinsics, near, far, gaussian_means, mode)
    b = fake.shape[0]
    color_sh = fake[:, :, None, None].expand(-1, -1, 3, 1)
    out = render_cuda(extrinsics, intrinsics, near, far, image_shape,
                      torch.zeros((b, 3), dtype=fake.dtype, device=fake.device),
                      gaussian_means, gaussian_covariances, gaussian_opacities, color_sh,
                      scale_invariant=scale_invariant)
    return out.color.mean(dim=1)


def render_depth_scenes(
    extrinsics: Tensor,             # (b,v,4,4)
    intrinsics: Tensor,             # (b,v,3,3)
    near: Tensor,                   # (b,v)
    far: Tensor,                    # (b,v)
    image_shape: tuple[int, int],
    gaussian_means: Tensor,         # (b,g,3)      -- NOT replicated per view
    gaussian_covariances: Tensor,   # (b,g,3,3)
    gaussian_opacities: Tensor,     # (b,g)
    scale_invariant: bool = True,
    mode: DepthRenderingMode = "depth",
) -> Tensor:
    """Scene-major form of ``render_depth_cuda`` (what ``DecoderSplattingCUDA.render_depth`` calls): the same
    images as the reference's v-fold ``repeat`` of the Gaussians (decoder_splatting_cuda.py:93-114), but only the
    grey "colour" — the one per-(view, Gaussian) quantity — is per view; means, covariances and opacities of a scene
    are handed to the rasterizer once, shared by its v views (one call per scene).  Returns (b,v,h,w)."""
    b, v = extrinsics.shape[:2]
    views = _view_table(extrinsics.flatten(0, 1), intrinsics.flatten(0, 1), near.flatten(0, 1), far.flatten(0, 1),
                        torch.zeros(3, dtype=torch.float32, device=extrinsics.device), scale_invariant)
    outs = []
    for s in range(b):
        fake = _depth_as_color(extrinsics[s], near[s], far[s], gaussian_means[s][None].expand(v, -1, -1), mode)   # (v,g)
        out = _render_views(views[s * v:(s + 1) * v], image_shape, gaussian_means[s][None], gaussian_covariances[s][None],
                            gaussian_opacities[s][None], fake[:, :, None, None].expand(-1, -1, 3, 1), None, True)
        outs.append(out.color.mean(dim=1))
    return torch.stack(outs)
