"""Host side of the rasterizer: the drop-in ``GaussianRasterizationSettings`` /
``GaussianRasterizer`` API of the reference's external dependency plus a multi-view batched
autograd op, both running on the hand-written HIP library (``liblsr_hip.so``).

Reference boundary (paths relative to /root/reference):
  * constructed with 12 keyword fields at src/model/decoder/cuda_splatting.py:132-145;
  * called as ``rasterizer(means3D=, means2D=, shs=, colors_precomp=, features=, opacities=,
    cov3D_precomp=)`` and unpacked as ``image, feature_map, mask, depth_map, _`` at :150-158;
  * matrices are row-major tensors holding the transposed matrices (:116-118).

No CPU fallback exists here on purpose: inputs must live on a ROCm device and the extension must
be loadable, otherwise a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
from torch import Tensor, nn

from . import _lib
from ._lib import Dims, InGrads, Inputs, LsrError, OutGrads, Outputs


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool
    debug: bool


def make_view_table(viewmatrix: Tensor, projmatrix: Tensor, campos: Tensor, tanfovx, tanfovy,
                    bg: Tensor, scene_scale=None) -> Tensor:
    """(V,44) device table: viewmatrix(16) projmatrix(16) campos(3) tanfovx tanfovy bg(3)
    scene_scale(1) reserved(3).  All arguments batched over V; tanfov / scene_scale may be python
    floats or tensors (no host sync).  ``scene_scale`` (default 1) is applied to the Gaussians'
    means (and squared to their covariances) inside the kernel."""
    V = viewmatrix.shape[0]
    dev, dt = viewmatrix.device, torch.float32

    def col(t):
        if not torch.is_tensor(t):
            t = torch.full((V,), float(t), dtype=dt, device=dev)
        return t.to(device=dev, dtype=dt).reshape(-1, 1).expand(V, 1)

    return torch.cat([viewmatrix.reshape(V, 16).to(dt), projmatrix.reshape(V, 16).to(dt),
                      campos.reshape(V, 3).to(dt), col(tanfovx), col(tanfovy),
                      bg.reshape(-1, 3).to(dt).expand(V, 3), col(1.0 if scene_scale is None else scene_scale),
                      torch.zeros((V, 3), dtype=dt, device=dev)], dim=1).contiguous()


def build_view_table(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, background: Tensor,
                     scale_invariant: bool = True) -> Tensor:
    """(V,44) camera table computed on the device by one kernel (``lsr_build_views``): what the
    reference derives per call with get_fov / get_projection_matrix / inverse / matmul
    (cuda_splatting.py:75-82,111-118).  ``background`` is ``(3,)`` or ``(V,3)``.  No gradient flows
    to the cameras (the reference's rasterizer has none either)."""
    lib = _lib.load()
    dev = extrinsics.device
    if dev.type != "cuda":
        raise LsrError("build_view_table runs on the MI355X; use make_view_table for host tensors")
    V = extrinsics.shape[0]
    f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    ext, intr, nr, fr, bg = f(extrinsics), f(intrinsics), f(near), f(far), f(background)
    out = torch.empty((V, _lib.VIEW_FLOATS), dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        _lib.check(lib.lsr_build_views(V, _ptr(ext), _ptr(intr), _ptr(nr), _ptr(fr), _ptr(bg),
                                       0 if bg.dim() == 1 else 3, 1 if scale_invariant else 0, _ptr(out), stream),
                   "lsr_build_views")
    return out


def _ptr(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _prep(t: Optional[Tensor], name: str, dev) -> Optional[Tensor]:
    if t is None:
        return None
    if t.device != dev:
        raise LsrError(f"{name} is on {t.device}, expected {dev}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _stride(t: Optional[Tensor], base_dims: int, V: int, name: str, groups: Optional[list] = None) -> int:
    """0 for a tensor shared by all views (G,...); elements per slice for a (V,G,...) tensor (one
    slice per view) or a (B,G,...) tensor with V % B == 0 (one slice per group of V/B consecutive
    views: B scenes rendered from V/B cameras each).  ``groups`` collects the slice counts seen."""
    if t is None or t.dim() == base_dims:
        return 0
    if t.dim() != base_dims + 1 or t.shape[0] < 1 or V % t.shape[0] != 0:
        raise LsrError(f"{name}: expected {base_dims} dims (shared) or a leading dim dividing the {V} views, "
                       f"got {tuple(t.shape)}")
    if groups is not None:
        groups.append(t.shape[0])
    return t[0].numel()


# statistics of the most recent forward (host-side; the pair count is known after prepare())
LAST_STATS: dict = {}

# Axis convention of the COLOUR SH basis (lsr_dims.color_sh_convention).  The 12-field settings tuple
# of the reference API has no room for it, so it is a process-wide switch:
#   "3dgs"      (default) the published 3DGS rasterizer's basis;
#   "reference" the reference's own eval_sh naming (src/misc/sh_utils.py:62-65), which its rotate_sh
#               (e3nn) is consistent with and which the fused latent-feature SH path always uses.
# Which one the external CUDA fork uses is unknown offline (INTEGRATION.md §"Colour SH convention").
_COLOR_SH_CONVENTION = {"3dgs": _lib.SH_AXES_3DGS, "reference": _lib.SH_AXES_REFERENCE}[
    __import__("os").environ.get("LSR_COLOR_SH_CONVENTION", "3dgs")]


def set_color_sh_convention(name: str) -> None:
    """``"3dgs"`` or ``"reference"`` — see the comment above; applies to subsequent forward calls."""
    global _COLOR_SH_CONVENTION
    _COLOR_SH_CONVENTION = {"3dgs": _lib.SH_AXES_3DGS, "reference": _lib.SH_AXES_REFERENCE}[name]


def get_color_sh_convention() -> str:
    return "reference" if _COLOR_SH_CONVENTION == _lib.SH_AXES_REFERENCE else "3dgs"


# ---- speculative workspace sizing (lsr_forward_speculative) ----
# The synchronous forward sizes its binning workspace after the device has counted the pairs, and the device idles while
# the host turns that number into the next launches.  Calls of a shape that has been rendered before launch the whole
# forward at once with room for 1.25 x the largest recent pair count (and a sort-tier hint above the longest recent tile
# list) and read the true counts afterwards, while the device is already sorting: identical results, no idle gap.  A scene
# that needs more than was provided is simply run again through the exact path (and raises the estimate).
# LSR_SPECULATIVE=0 turns it off.
_SPECULATE = __import__("os").environ.get("LSR_SPECULATIVE", "1") != "0"
# LSR_REACHED_ONLY=0: the binning keeps the published algorithm's pairs (every tile of the 3-sigma square), so that
# num_pairs / tile offsets / the canonical sorted list are the published ones; by default (ABI v9 LSR_FWD_REACHED_ONLY) only pairs
# whose alpha >= 1/255 footprint reaches the tile are counted, keyed and sorted — identical images, render lists and gradients
_REACHED_ONLY = __import__("os").environ.get("LSR_REACHED_ONLY", "1") != "0"


_EARLY_FRONT = __import__("os").environ.get("LSR_EARLY_FRONT", "1") != "0"     # ABI v10: the front half launched ahead of the allocations


def set_reached_only(on: bool) -> None:
    """Whether subsequent forwards bin only the pairs that can reach a pixel (default) or the published algorithm's pairs."""
    global _REACHED_ONLY
    _REACHED_ONLY = bool(on)


# LSR_CLEAR_IN_FORWARD=0: lsr_backward clears its gradient workspace itself (the pre-v9 behaviour)
_FWD_CLEARS_GRAD = __import__("os").environ.get("LSR_CLEAR_IN_FORWARD", "1") != "0"
# shape key -> (pairs, longest tile list): decaying maxima of the recent calls of the shape.  Bounded (the 64 most recently
# used shapes: G changes from step to step under densification) and guarded (forwards may run on several host threads).
_ESTIMATES: "__import__('collections').OrderedDict" = __import__("collections").OrderedDict()
_ESTIMATES_LOCK = __import__("threading").Lock()
_ESTIMATES_MAX = 64
SPECULATION_STATS = dict(speculative=0, reruns=0, exact=0)


def _estimate_get(key):
    with _ESTIMATES_LOCK:
        est = _ESTIMATES.get(key)
        if est is not None:
            _ESTIMATES.move_to_end(key)
        return est


def _estimate_update(key, pairs: int, longest: int) -> None:
    with _ESTIMATES_LOCK:
        old = _ESTIMATES.get(key, (0, 0))
        _ESTIMATES[key] = (max(pairs, int(0.95 * old[0])), max(longest, int(0.95 * old[1])))
        _ESTIMATES.move_to_end(key)
        while len(_ESTIMATES) > _ESTIMATES_MAX:
            _ESTIMATES.popitem(last=False)


def _tier_hint(longest: float) -> int:
    """The sort-tier boundary (1024, 2048, 4096, 8192, ...) at or above 1.1 x the expected longest list.  (A wider margin
    costs real time: a hint beyond 4096 adds the launch of the second sort tier — 8 us per call at the bench scene, whose
    longest list of 3289 keys a 1.3 x margin pushed over the boundary.)"""
    t = 1024
    while t < 1.1 * longest:
        t *= 2
    return t


class _Plan:
    """Everything one forward call hands to the matching backward."""
    __slots__ = ("dims", "geom", "bin", "img", "num_pairs", "radii", "V", "G", "H", "W", "C",
                 "color_mode", "K", "gradws")


class _RasterizeViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, views, means3D, means2D, cov3D, opacities, shs, colors_precomp, features,
                H: int, W: int, sh_degree: int, debug: bool, feat_sh_degree: int = -1,
                shs_channel_major: bool = False, pair_capacity: int = 0, max_tile_hint: int = 0, grad_mode: bool = True):
        lib = _lib.load()
        ctx.set_materialize_grads(False)  # unused outputs (mask / depth ...) arrive as None, not zeros
        dev = means3D.device
        if dev.type != "cuda":
            raise LsrError("latentsplat_amd rasterizer runs on MI355X only: tensors must be on a "
                           "ROCm ('cuda') device; there is no CPU fallback")
        if shs is not None and colors_precomp is not None:
            raise LsrError("provide at most one of shs / colors_precomp")
        if shs is None and colors_precomp is None and features is None:
            raise LsrError("nothing to render: shs, colors_precomp and features are all None")
        views = _prep(views, "views", dev)
        V = views.shape[0]
        means3D = _prep(means3D, "means3D", dev)
        cov3D = _prep(cov3D, "cov3D_precomp", dev)
        opacities = _prep(opacities, "opacities", dev)
        shs = _prep(shs, "shs", dev)
        colors_precomp = _prep(colors_precomp, "colors_precomp", dev)
        features = _prep(features, "features", dev)
        G = means3D.shape[-2]
        cov_elems = 9 if cov3D.shape[-2:] == (3, 3) else 6
        cov_base = 3 if cov_elems == 9 else 2
        feat_sh = features is not None and feat_sh_degree >= 0
        color = shs if shs is not None else colors_precomp
        color_mode = _lib.COLOR_SH if shs is not None else (
            _lib.COLOR_PRECOMP if colors_precomp is not None else _lib.COLOR_NONE)
        Cf = 0 if features is None else (features.shape[-2] if feat_sh else features.shape[-1])
        Kf = features.shape[-1] if feat_sh else 0
        # every per-Gaussian tensor must agree with means3D on G and carry the documented trailing
        # dims: the kernels index them by G without bounds checks
        def _expect(t, name, tail):
            if t is not None and tuple(t.shape[-len(tail):]) != tuple(tail):
                raise LsrError(f"{name}: trailing shape {tuple(t.shape[-len(tail):])}, expected {tuple(tail)} "
                               f"(G = {G} from means3D)")
        if means3D.shape[-1] != 3:
            raise LsrError(f"means3D: last dimension {means3D.shape[-1]}, expected 3")
        _expect(cov3D, "cov3D_precomp", (G, 3, 3) if cov_elems == 9 else (G, 6))
        _expect(opacities, "opacities", (G, 1))
        _expect(colors_precomp, "colors_precomp", (G, 3))
        if shs is not None:
            _expect(shs, "shs", (G, 3, shs.shape[-1]) if shs_channel_major else (G, shs.shape[-2], 3))
        if features is not None:
            _expect(features, "features", (G,) + tuple(features.shape[-2:] if feat_sh else features.shape[-1:]))
        if Cf > _lib.MAX_FEAT_CHANNELS:
            raise LsrError(f"features has {Cf} channels; at most {_lib.MAX_FEAT_CHANNELS} are supported")
        K = 0 if shs is None else (shs.shape[-1] if shs_channel_major else shs.shape[-2])
        slices: list = []
        strides = (_stride(means3D, 2, V, "means3D", slices), _stride(cov3D, cov_base, V, "cov3D_precomp", slices),
                   _stride(opacities, 2, V, "opacities", slices),
                   _stride(color, 3 if shs is not None else 2, V, "shs/colors_precomp", slices),
                   _stride(features, 3 if feat_sh else 2, V, "features", slices))
        if len(set(slices)) > 1:
            raise LsrError(f"per-view / per-scene inputs disagree on their leading dimension: {sorted(set(slices))}")
        vpg = V // slices[0] if slices else 0           # views per group; 1 = one slice per view
        if vpg > 1 and len(slices) != sum(t is not None for t in (means3D, cov3D, opacities, color, features)):
            raise LsrError("with per-scene inputs (leading dim < number of views) every per-Gaussian input must "
                           "carry the scene dimension")
        # what earlier calls of this shape needed (speculative workspace sizes; the capacity of the key segments: real
        # device memory, lsr_dims.seg_cap_hint — 1.3 x the longest recent tile list instead of the default 8192 keys)
        shape_key = (dev.index, V, G, H, W, Cf, color_mode, K, Kf, vpg, cov_elems, strides, _REACHED_ONLY)
        est = _estimate_get(shape_key) if (_SPECULATE and G > 0) else None
        d = Dims(V, G, H, W, Cf, color_mode, int(sh_degree), K, *strides,
                 cov_elems, _lib.FEAT_SH if feat_sh else _lib.FEAT_DIRECT, max(int(feat_sh_degree), 0), Kf,
                 1 if (shs is not None and shs_channel_major) else 0, vpg if vpg > 1 else 0,
                 _COLOR_SH_CONVENTION,
                 # a backward will follow: the forward narrows the render lists to where every entry contributed and
                 # zeroes the backward's gradient workspace on the side (ABI v9: lsr_backward then skips its clear)
                 # (grad_mode: torch.is_grad_enabled() of the CALLER — inside Function.forward it is always off)
                 ((_lib.FWD_FOR_BACKWARD | (_lib.FWD_CLEARS_GRAD if _FWD_CLEARS_GRAD else 0)) if (grad_mode and any(ctx.needs_input_grad) and G > 0) else 0)
                 | (_lib.FWD_REACHED_ONLY if _REACHED_ONLY else 0),
                 # (never below the sort-tier hint of the speculative forward: a launch structure chosen for lists of up to `hint`
                 # keys would otherwise include the fallback scatter for segments shorter than that)
                 min(max(int(1.3 * est[1]) + 64, _tier_hint(est[1])), 2 ** 31 - 1) if (est is not None and est[1] > 0) else 0)
        inp = Inputs(_ptr(views), _ptr(means3D), _ptr(cov3D), _ptr(opacities), _ptr(color), _ptr(features))
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        u8 = dict(dtype=torch.uint8, device=dev)
        geom_bytes = lib.lsr_geom_workspace_bytes(C.byref(d))
        if geom_bytes == 0:
            raise LsrError("invalid rasterizer dimensions / argument shapes")
        npairs, maxtile = C.c_int64(0), C.c_int32(0)
        # a speculative forward knows its pair capacity up front: the front half (projection, key emission, tile scan) is
        # launched as soon as its two buffers exist and the device works while the host allocates the other seven (ABI v10
        # lsr_forward_front; a synchronised call into an idle device: V = 1 0.128 -> 0.120 ms, V = 4 0.201 -> 0.193, the
        # reference's per-view loop 0.186 -> 0.178 per view).  Not for no-sync calls, which are issued back to back: there the
        # second C call is 5 us of host time per call that nothing hides (V = 1 streamed 0.083 -> 0.088).
        cap, hint = (int(1.25 * est[0]) + 4096, _tier_hint(est[1])) if est is not None else (0, 0)
        # (a capacity the 32-bit offsets of the speculative / no-sync forward cannot address: the exact path)
        speculate = pair_capacity <= 0 and est is not None and est[0] > 0 and not debug and cap < 2 ** 32
        front_cap = cap if speculate else 0
        with torch.cuda.device(dev):
            geom = torch.empty(geom_bytes, **u8)
            radii = torch.empty((V, G), dtype=torch.int32, device=dev)
            if _EARLY_FRONT and G > 0 and 0 < front_cap < 2 ** 32:
                d.forward_flags |= _lib.FWD_FRONT_DONE
                _lib.check(lib.lsr_forward_front(C.byref(d), C.byref(inp), _ptr(geom), _ptr(radii), front_cap, stream), "lsr_forward_front")
            img = torch.empty(lib.lsr_image_workspace_bytes(C.byref(d)), **u8)
            # everything that does not depend on the pair count is allocated BEFORE prepare(): the GPU
            # idles between prepare's synchronisation and the launches of render(), so that window
            # should hold as little host work as possible
            f32 = dict(dtype=torch.float32, device=dev)
            out_color = torch.empty((V, 3, H, W), **f32) if color is not None else None
            out_feat = torch.empty((V, Cf, H, W), **f32) if Cf else None
            out_mask = torch.empty((V, H, W), **f32)
            out_depth = torch.empty((V, H, W), **f32)
            gradws = (torch.empty(lib.lsr_grad_workspace_bytes(C.byref(d)), **u8)
                      if d.forward_flags & _lib.FWD_CLEARS_GRAD else None)
            outs = Outputs(_ptr(out_color), _ptr(out_feat), _ptr(out_mask), _ptr(out_depth), _ptr(radii), _ptr(gradws))
            if pair_capacity > 0:
                # latency mode: no host synchronisation; the pair count stays on the device
                # (last_forward_status() reads it back when the caller wants to check for overflow)
                npairs.value, maxtile.value = int(pair_capacity), int(max_tile_hint)
                layout_pairs = npairs.value
                binws = torch.empty(lib.lsr_binning_workspace_bytes(C.byref(d), npairs.value, 2 ** 31 - 1), **u8)
                _lib.check(lib.lsr_forward_nosync(C.byref(d), C.byref(inp), _ptr(geom), _ptr(binws), _ptr(img),
                                                  npairs.value, maxtile.value, C.byref(outs), stream),
                           "lsr_forward_nosync")
            else:
                done = False
                if speculate:
                    binws = torch.empty(lib.lsr_binning_workspace_bytes(C.byref(d), cap, hint), **u8)
                    ov = C.c_int32(0)
                    _lib.check(lib.lsr_forward_speculative(C.byref(d), C.byref(inp), _ptr(geom), _ptr(binws), _ptr(img), cap, hint,
                                                           C.byref(outs), C.byref(npairs), C.byref(maxtile), C.byref(ov), stream),
                               "lsr_forward_speculative")
                    done = not ov.value
                    SPECULATION_STATS["speculative" if done else "reruns"] += 1
                    if done:
                        layout_pairs = cap          # the workspace layout the backward has to use
                if not done:
                    SPECULATION_STATS["exact"] += 1
                    d.forward_flags &= ~_lib.FWD_FRONT_DONE      # (a speculative forward that fell short is run again from the start)
                    try:
                        _lib.check(lib.lsr_forward_prepare(C.byref(d), C.byref(inp), _ptr(geom), _ptr(radii),
                                                           C.byref(npairs), C.byref(maxtile), stream),
                                   "lsr_forward_prepare")
                        # the largest pair-count dependent allocation (the likeliest out-of-memory site of a forward)
                        binws = torch.empty(lib.lsr_binning_workspace_bytes(C.byref(d), npairs.value, maxtile.value), **u8)
                        _lib.check(lib.lsr_forward_render(C.byref(d), C.byref(inp), _ptr(geom), _ptr(binws), _ptr(img),
                                                          npairs.value, maxtile.value, C.byref(outs), stream),
                                   "lsr_forward_render")
                        layout_pairs = npairs.value
                    except BaseException:
                        # (a no-op since ABI v7 — every launch of a forward is on the caller's stream — kept for v6 libraries)
                        lib.lsr_forward_abandon(stream)
                        raise
                if _SPECULATE:
                    _estimate_update(shape_key, npairs.value, maxtile.value)
            if debug:
                torch.cuda.synchronize(dev)
                if pair_capacity <= 0:
                    # debug mode of the reference API (settings.debug): the synchronous forward sizes its binning
                    # workspace with the host's pair count; the device flags a list that did not fit (it cannot
                    # happen unless the two counts disagree) — surface it instead of rendering a tile short
                    n, mt, ov = C.c_int64(0), C.c_int32(0), C.c_int32(0)
                    _lib.check(lib.lsr_forward_status(C.byref(d), _ptr(geom), C.byref(n), C.byref(mt), C.byref(ov), stream),
                               "lsr_forward_status")
                    if ov.value or n.value != npairs.value:
                        raise LsrError(f"synchronous forward: device pair count {n.value} / overflow {ov.value} disagrees "
                                       f"with the host's {npairs.value}")
        plan = _Plan()
        plan.dims, plan.geom, plan.bin, plan.img = d, geom, binws, img
        plan.num_pairs, plan.radii, plan.gradws = layout_pairs, radii, gradws
        plan.V, plan.G, plan.H, plan.W, plan.C, plan.color_mode, plan.K = V, G, H, W, Cf, color_mode, K
        ctx.plan = plan
        LAST_STATS.update(num_pairs=None if pair_capacity > 0 else npairs.value, max_tile_pairs=maxtile.value, views=V,
                          gaussians=G, pair_capacity=int(pair_capacity))
        # what last_forward_status() reports: the host already knows the counts of a synchronous forward;
        # a no-sync forward keeps its (caller-requested) plan alive so the header can be read back later
        global _LAST_PLAN, _LAST_STATUS
        if pair_capacity > 0:
            # only what lsr_forward_status reads: the dims and the geometry workspace (its header); the capacity
            # sized binning workspace and the image workspace stay owned by the autograd graph alone
            keep = _Plan()
            keep.dims, keep.geom = d, geom
            _LAST_PLAN, _LAST_STATUS = keep, None
        else:
            _LAST_PLAN, _LAST_STATUS = None, dict(num_pairs=npairs.value, max_tile_pairs=maxtile.value, overflow=False)
        ctx.debug = debug
        ctx.m2d_shape = None if means2D is None else tuple(means2D.shape)
        ctx.has = (shs is not None, colors_precomp is not None, features is not None)
        ctx.mark_non_differentiable(radii)
        empty = views.new_empty(0)
        outs = (out_color if out_color is not None else empty,
                out_feat if out_feat is not None else empty, out_mask, out_depth, radii)
        # the rendered images are inputs of the backward (lsr_backward's `fwd`)
        ctx.save_for_backward(views, means3D, cov3D, opacities,
                              color if color is not None else views.new_empty(0),
                              features if features is not None else views.new_empty(0),
                              outs[0], outs[1], out_depth)
        return outs

    @staticmethod
    def backward(ctx, g_color, g_feat, g_mask, g_depth, _g_radii):
        lib = _lib.load()
        plan: _Plan = ctx.plan
        views, means3D, cov3D, opacities, color, features, f_color, f_feat, f_depth = ctx.saved_tensors
        has_shs, has_cp, has_f = ctx.has
        dev = means3D.device
        d = plan.dims
        V, G = plan.V, plan.G
        f32 = dict(dtype=torch.float32, device=dev)

        def g_in(t):  # contiguous fp32 grad or None
            return None if t is None else t.contiguous().float()

        g_color = g_in(g_color) if (has_shs or has_cp) else None
        g_feat = g_in(g_feat) if has_f else None
        g_mask, g_depth = g_in(g_mask), g_in(g_depth)
        inp = Inputs(_ptr(views), _ptr(means3D), _ptr(cov3D), _ptr(opacities),
                     _ptr(color) if (has_shs or has_cp) else None, _ptr(features) if has_f else None)
        d_means = torch.empty_like(means3D)
        d_cov = torch.empty_like(cov3D)
        d_opac = torch.empty_like(opacities)
        d_color = torch.empty_like(color) if (has_shs or has_cp) else None
        d_feat = torch.empty_like(features) if has_f else None
        want_m2d = ctx.m2d_shape is not None and ctx.needs_input_grad[2]
        d_m2d = torch.empty((V, G, 3), **f32) if want_m2d else None   # 12 B per (view, Gaussian) nobody reads otherwise
        # the gradient workspace the forward zeroed; a second backward over the same graph (retain_graph) gets a fresh one
        # and has lsr_backward clear it itself (dims without the flag)
        gradws, plan.gradws = plan.gradws, None
        if gradws is None:
            if d.forward_flags & _lib.FWD_CLEARS_GRAD:
                d = Dims.from_buffer_copy(d)
                d.forward_flags &= ~_lib.FWD_CLEARS_GRAD
            gradws = torch.empty(lib.lsr_grad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
        gout = OutGrads(_ptr(g_color), _ptr(g_feat), _ptr(g_mask), _ptr(g_depth))
        fwd = Outputs(_ptr(f_color) if f_color.numel() else None, _ptr(f_feat) if f_feat.numel() else None,
                      None, _ptr(f_depth), None)
        gin = InGrads(_ptr(d_means), _ptr(d_cov), _ptr(d_opac), _ptr(d_color), _ptr(d_feat), _ptr(d_m2d))
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(lib.lsr_backward(C.byref(d), C.byref(inp), _ptr(plan.geom), _ptr(plan.bin),
                                        _ptr(plan.img), plan.num_pairs, _ptr(plan.radii), C.byref(fwd),
                                        C.byref(gout), _ptr(gradws), C.byref(gin), stream), "lsr_backward")
            if ctx.debug:
                torch.cuda.synchronize(dev)
        if not want_m2d:
            d_m2d = None
        elif len(ctx.m2d_shape) == 2:  # one (G,3) tensor shared by the views
            d_m2d = d_m2d.sum(0) if V > 1 else d_m2d[0]
        # order: views, means3D, means2D, cov3D, opacities, shs, colors_precomp, features, H, W, deg,
        #        debug, feat_sh_degree, shs_channel_major
        return (None, d_means, d_m2d, d_cov, d_opac, d_color if has_shs else None,
                d_color if has_cp else None, d_feat, None, None, None, None, None, None, None, None, None)


def rasterize_views(views: Tensor, image_height: int, image_width: int, sh_degree: int, means3D: Tensor,
                    cov3D_precomp: Tensor, opacities: Tensor, shs: Optional[Tensor] = None,
                    colors_precomp: Optional[Tensor] = None, features: Optional[Tensor] = None,
                    means2D: Optional[Tensor] = None, debug: bool = False,
                    feature_sh: Optional[Tensor] = None, shs_channel_major: bool = False,
                    pair_capacity: Optional[int] = None, max_tile_hint: int = 4096):
    """Render V views in one call.  ``views`` is the (V,44) table of :func:`make_view_table`; every
    per-Gaussian tensor is either shared ``(G,...)`` or per view ``(V,G,...)``.
    Returns ``(color (V,3,H,W)|None, feature (V,C,H,W)|None, mask (V,H,W), depth (V,H,W), radii (V,G))``.
    ``means2D`` (optional, ``(V,G,3)``) only exists to receive the NDC-space mean gradient.

    Scene-level (fused) inputs, all optional: ``cov3D_precomp`` may be full ``(...,3,3)`` matrices;
    ``feature_sh (..., C, Kf)`` (instead of ``features``) makes the kernel evaluate the latent
    features ``0.5 + eval_sh(dir)`` itself (degree <= 2, C*Kf <= 120); ``shs`` may be passed in
    the stored ``(...,3,K)`` layout with ``shs_channel_major=True``; the per-view scene scale
    lives in the view table.

    Latency mode: with ``pair_capacity`` (number of (Gaussian, tile) pairs the binning workspace
    is sized for, e.g. 1.5 x the count of an earlier frame from :func:`last_forward_status`) the
    forward never waits for the device (``lsr_forward_nosync``) and can be captured in a
    ``torch.cuda.graph``.  A scene that needs more pairs renders truncated lists and raises the
    overflow flag reported by :func:`last_forward_status` — check it at a synchronisation point of
    your choice."""
    V = views.shape[0]
    feat_sh_degree = -1
    if feature_sh is not None:
        if features is not None:
            raise LsrError("provide at most one of features / feature_sh")
        from math import isqrt
        feat_sh_degree = isqrt(feature_sh.shape[-1]) - 1
        features = feature_sh
    color, feat, mask, depth, radii = _RasterizeViews.apply(
        views, means3D, means2D, cov3D_precomp, opacities, shs, colors_precomp, features,
        int(image_height), int(image_width), int(sh_degree), bool(debug), int(feat_sh_degree),
        bool(shs_channel_major), int(pair_capacity or 0), int(max_tile_hint), torch.is_grad_enabled())
    return (color if color.numel() else None, feat if feat.numel() else None, mask, depth, radii)


_LAST_PLAN = None      # plan of the most recent NO-SYNC forward (holds its workspaces), else None
_LAST_STATUS = None    # host-side counts of the most recent synchronous forward, else None


def last_forward_status() -> dict:
    """``{num_pairs, max_tile_pairs, overflow}`` of the most recent forward of this process.  For a
    synchronous forward these are the counts the host already holds; for a no-sync forward they are
    read back from its geometry workspace (synchronises the current stream) — after a graph replay
    that is the replayed forward (the workspace addresses are static)."""
    if _LAST_STATUS is not None:
        return dict(_LAST_STATUS)
    if _LAST_PLAN is None:
        raise LsrError("no forward has run yet")
    lib = _lib.load()
    plan = _LAST_PLAN
    dev = plan.geom.device
    n, mt, ov = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    with torch.cuda.device(dev):
        _lib.check(lib.lsr_forward_status(C.byref(plan.dims), _ptr(plan.geom), C.byref(n), C.byref(mt), C.byref(ov),
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "lsr_forward_status")
    return dict(num_pairs=n.value, max_tile_pairs=mt.value, overflow=bool(ov.value))


def fused_feature_sh_supported(feature_sh: Optional[Tensor]) -> bool:
    """True when the kernel can evaluate these latent SH coefficients itself."""
    if feature_sh is None:
        return False
    from math import isqrt
    return isqrt(feature_sh.shape[-1]) - 1 <= 2 and feature_sh.shape[-2] * feature_sh.shape[-1] <= _lib.MAX_SH_GROUP_FLOATS


def _covariance_from_scale_rotation(scales: Tensor, rotations: Tensor, modifier: float) -> Tensor:
    """Sigma = R S S^T R^T packed as xx,xy,xz,yy,yz,zz; quaternion (w,x,y,z) as upstream."""
    q = rotations / rotations.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)
    R = R.reshape(*q.shape[:-1], 3, 3)
    M = R * (scales * modifier)[..., None, :]
    S = M @ M.transpose(-1, -2)
    return torch.stack([S[..., 0, 0], S[..., 0, 1], S[..., 0, 2], S[..., 1, 1], S[..., 1, 2], S[..., 2, 2]], -1)


def _pack_view(rs: GaussianRasterizationSettings, dev) -> Tensor:
    """(1,44) view record of one settings tuple: one small launch (lsr_pack_view) instead of ~8 PyTorch
    ops; tan(fov) goes by value when it is a Python number and stays on the device when it is a tensor."""
    if dev.type != "cuda":
        raise LsrError("latentsplat_amd rasterizer runs on MI355X only: tensors must be on a ROCm ('cuda') device; "
                       "there is no CPU fallback")
    lib = _lib.load()
    f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    vm, pm, cp, bg = f(rs.viewmatrix), f(rs.projmatrix), f(rs.campos), f(rs.bg)
    tx, ty = rs.tanfovx, rs.tanfovy
    txd = f(tx).reshape(-1) if torch.is_tensor(tx) else None
    tyd = f(ty).reshape(-1) if torch.is_tensor(ty) else None
    out = torch.empty((1, _lib.VIEW_FLOATS), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.lsr_pack_view(_ptr(vm), _ptr(pm), _ptr(cp), _ptr(bg), 0.0 if txd is not None else float(tx),
                                     0.0 if tyd is not None else float(ty), _ptr(txd), _ptr(tyd), _ptr(out),
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "lsr_pack_view")
    return out


_NEAR_CULL = 0.2   # view-space z at or below which a Gaussian is culled (LSR_NEAR_CULL in csrc/lsr_internal.h)


class GaussianRasterizer(nn.Module):
    """Drop-in for ``diff_gaussian_rasterization.GaussianRasterizer`` (one view per call)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: Tensor) -> Tensor:
        """Published API of the upstream class (its frustum test keeps everything with view-space z > 0.2;
        the screen-bounds half of that test is commented out upstream).  ``viewmatrix`` is the transposed
        world-to-view matrix the settings tuple carries.  The reference never calls it; plain tensor ops."""
        with torch.no_grad():
            vm = torch.as_tensor(self.raster_settings.viewmatrix, dtype=positions.dtype, device=positions.device)
            z = positions @ vm[:3, 2] + vm[3, 2]
            return z > _NEAR_CULL

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, features=None,
                scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if shs is not None and colors_precomp is not None:
            raise Exception("Please provide at most one of SHs / precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if cov3D_precomp is None:
            cov3D_precomp = _covariance_from_scale_rotation(scales, rotations, float(rs.scale_modifier))
        views = _pack_view(rs, means3D.device)
        color, feat, mask, depth, radii = _RasterizeViews.apply(
            views, means3D, means2D, cov3D_precomp, opacities, shs, colors_precomp, features,
            int(rs.image_height), int(rs.image_width), int(rs.sh_degree), bool(rs.debug), -1, False, 0, 0, torch.is_grad_enabled())
        return (color[0] if color.numel() else None, feat[0] if feat.numel() else None,
                mask, depth, radii[0])
