"""CPU ORACLE bindings (test infrastructure — NOT the product path; parity unpinned, see
raster_oracle.c header).

ctypes/numpy front-end of ``oracle/raster_oracle.c`` plus a ``GaussianRasterizer``-shaped
torch wrapper (``OracleRasterizer``) so the oracle can stand in for
``diff_gaussian_rasterization`` when the reference's Python wrapper
(/root/reference/src/model/decoder/cuda_splatting.py:132-158) is imported to generate golden
fixtures, and so ``bench.py`` can time it as the ``cpu_baseline`` leg.

Only ``tests/``, ``__graft_entry__.smoke()``, ``tests/golden/make_golden.py`` and ``bench.py``'s
cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import NamedTuple, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_preprocess.restype = ctypes.c_int64
    return _LIB


def set_sh_convention(name: str) -> None:
    """Colour SH axis convention of the oracle: "3dgs" (default) or "reference" (sh_utils.py naming)."""
    lib().oracle_set_sh_convention(ctypes.c_int({"3dgs": 0, "reference": 1}[name]))


def set_fma_contraction(on: bool) -> None:
    """Arithmetic convention of the oracle's projection stage: False (default) = every float operation separate (what
    the bit-exact index tests assume); True = products fused into the sums they feed, the way a compiler with
    contraction on would build the published source (raster_oracle.c, oracle_set_fma_contraction)."""
    lib().oracle_set_fma_contraction(ctypes.c_int(1 if on else 0))


def get_fma_contraction() -> bool:
    return bool(lib().oracle_get_fma_contraction())


def _p(a: Optional[np.ndarray]):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class View(NamedTuple):
    H: int
    W: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray  # (3,)
    viewmatrix: np.ndarray  # (4,4) row-major memory = transposed world->view
    projmatrix: np.ndarray  # (4,4) row-major memory = transposed world->clip
    campos: np.ndarray  # (3,)
    sh_degree: int


def forward(view: View, means3D, cov3D, opacities, shs=None, colors_precomp=None, features=None,
            keep_intermediates: bool = True) -> dict:
    """Runs preprocess -> bin/sort -> render for ONE view. All arrays numpy float32."""
    L = lib()
    means3D, cov3D = _f32(means3D), _f32(cov3D)
    opacities = _f32(opacities).reshape(-1)
    shs, colors_precomp, features = _f32(shs), _f32(colors_precomp), _f32(features)
    G = means3D.shape[0]
    H, W = int(view.H), int(view.W)
    C = 0 if features is None else features.shape[1]
    K = 0 if shs is None else shs.shape[1]
    has_color = shs is not None or colors_precomp is not None
    vm, pm = _f32(view.viewmatrix).reshape(16), _f32(view.projmatrix).reshape(16)
    cp, bg = _f32(view.campos).reshape(3), _f32(view.bg).reshape(3)
    gx, gy = (W + 15) // 16, (H + 15) // 16

    radii = np.zeros(G, np.int32)
    rect = np.zeros((G, 4), np.int32)
    tiles = np.zeros(G, np.uint32)
    depth = np.zeros(G, np.float32)
    xy = np.zeros((G, 2), np.float32)
    co = np.zeros((G, 4), np.float32)
    rgb = np.zeros((G, 3), np.float32) if has_color else None
    clamped = np.zeros((G, 3), np.uint8) if has_color else None
    P = L.oracle_preprocess(
        ctypes.c_int(G), ctypes.c_int(H), ctypes.c_int(W), _p(means3D), _p(cov3D), _p(opacities),
        _p(shs), ctypes.c_int(view.sh_degree), ctypes.c_int(K), _p(colors_precomp), _p(vm), _p(pm),
        _p(cp), ctypes.c_float(view.tanfovx), ctypes.c_float(view.tanfovy), _p(radii), _p(rect),
        _p(tiles), _p(depth), _p(xy), _p(co), _p(rgb), _p(clamped))
    keys = np.zeros(max(P, 1), np.uint64)
    plist = np.zeros(max(P, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    L.oracle_bin_sort(ctypes.c_int(G), ctypes.c_int(H), ctypes.c_int(W), _p(rect), _p(tiles),
                      _p(depth), ctypes.c_int64(P), _p(keys), _p(plist), _p(ranges))
    out_color = np.zeros((3, H, W), np.float32) if has_color else None
    out_feat = np.zeros((C, H, W), np.float32) if C else None
    out_mask = np.zeros((H, W), np.float32)
    out_depth = np.zeros((H, W), np.float32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.uint32)
    n_considered = np.zeros((H, W), np.uint32)
    fragile = np.zeros((4096, 2), np.int32)
    fragile_count = ctypes.c_int32(0)
    L.oracle_render(ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(C), _p(ranges), _p(plist),
                    _p(xy), _p(co), _p(depth), _p(rgb), _p(features), _p(bg), _p(out_color),
                    _p(out_feat), _p(out_mask), _p(out_depth), _p(final_T), _p(n_contrib), _p(n_considered),
                    _p(fragile), ctypes.c_int(4096), ctypes.byref(fragile_count))
    res = dict(color=out_color, feature=out_feat, mask=out_mask, depth=out_depth, radii=radii, P=int(P))
    if keep_intermediates:
        res.update(rect=rect, tiles_touched=tiles, gdepth=depth, xy=xy, conic_opacity=co, rgb=rgb,
                   clamped=clamped, keys=keys[:P], point_list=plist[:P], ranges=ranges,
                   final_T=final_T, n_contrib=n_contrib, n_considered=n_considered,
                   fragile=fragile[:min(fragile_count.value, 4096)], fragile_overflow=fragile_count.value > 4096)
    return res


def backward(view: View, means3D, cov3D, opacities, shs, colors_precomp, features, fwd: dict,
             dL_dcolor=None, dL_dfeature=None, dL_dmask=None, dL_ddepth=None, f64: bool = False) -> dict:
    """f64=True: the compositing backward evaluated in double precision under the float32 forward's keep / skip /
    stop decisions (oracle_render_backward_f64: the error-budget reference, not the published float32 recurrence)."""
    L = lib()
    means3D, cov3D = _f32(means3D), _f32(cov3D)
    shs, colors_precomp, features = _f32(shs), _f32(colors_precomp), _f32(features)
    dL_dcolor, dL_dfeature = _f32(dL_dcolor), _f32(dL_dfeature)
    dL_dmask, dL_ddepth = _f32(dL_dmask), _f32(dL_ddepth)
    G = means3D.shape[0]
    H, W = int(view.H), int(view.W)
    C = 0 if features is None else features.shape[1]
    K = 0 if shs is None else shs.shape[1]
    vm, pm = _f32(view.viewmatrix).reshape(16), _f32(view.projmatrix).reshape(16)
    cp, bg = _f32(view.campos).reshape(3), _f32(view.bg).reshape(3)
    plist = np.ascontiguousarray(fwd["point_list"]) if fwd["P"] else np.zeros(1, np.uint32)
    d_xy = np.zeros((G, 2), np.float64)
    d_conic = np.zeros((G, 3), np.float64)
    d_op = np.zeros(G, np.float64)
    d_rgb = np.zeros((G, 3), np.float64)
    d_feat = np.zeros((G, max(C, 1)), np.float64)
    d_z = np.zeros(G, np.float64)
    (L.oracle_render_backward_f64 if f64 else L.oracle_render_backward)(
        ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(C), _p(fwd["ranges"]), _p(plist),
        _p(fwd["xy"]), _p(fwd["conic_opacity"]), _p(fwd["gdepth"]), _p(fwd["rgb"]), _p(features),
        _p(bg), _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(dL_dcolor), _p(dL_dfeature),
        _p(dL_dmask), _p(dL_ddepth), _p(d_xy), _p(d_conic), _p(d_op), _p(d_rgb), _p(d_feat), _p(d_z))
    g_means = np.zeros((G, 3), np.float32)
    g_cov = np.zeros((G, 6), np.float32)
    g_op = np.zeros((G, 1), np.float32)
    g_shs = np.zeros((G, K, 3), np.float32) if shs is not None else None
    g_cp = np.zeros((G, 3), np.float32) if colors_precomp is not None else None
    g_m2d = np.zeros((G, 3), np.float32)
    L.oracle_preprocess_backward(
        ctypes.c_int(G), ctypes.c_int(H), ctypes.c_int(W), _p(means3D), _p(cov3D), _p(shs),
        ctypes.c_int(view.sh_degree), ctypes.c_int(K), ctypes.c_int(colors_precomp is not None),
        _p(vm), _p(pm), _p(cp), ctypes.c_float(view.tanfovx), ctypes.c_float(view.tanfovy),
        _p(fwd["radii"]), _p(fwd["clamped"]), _p(d_xy), _p(d_conic), _p(d_op), _p(d_rgb), _p(d_z),
        _p(g_means), _p(g_cov), _p(g_op), _p(g_shs), _p(g_cp), _p(g_m2d))
    return dict(means3D=g_means, cov3D=g_cov, opacities=g_op, shs=g_shs, colors_precomp=g_cp,
                features=d_feat[:, :C].astype(np.float32) if C else None, means2D=g_m2d,
                mid=dict(xy_ndc=d_xy, conic=d_conic, opacity=d_op, rgb=d_rgb, z=d_z))


# ---------------------------------------------------------------------------------------------
# torch wrapper with the drop-in call signature (cuda_splatting.py:132-158).
# ---------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: "object"
    scale_modifier: float
    viewmatrix: "object"
    projmatrix: "object"
    sh_degree: int
    campos: "object"
    prefiltered: bool
    debug: bool


def _view_from_settings(s) -> View:
    n = lambda t: t.detach().cpu().float().numpy()
    return View(int(s.image_height), int(s.image_width), float(s.tanfovx), float(s.tanfovy), n(s.bg),
                n(s.viewmatrix), n(s.projmatrix), n(s.campos), int(s.sh_degree))


def make_function():
    torch = _torch()

    class _OracleRasterize(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, shs, colors_precomp, features, opacities, cov3D, settings):
            n = lambda t: None if t is None else t.detach().cpu().float().numpy()
            view = _view_from_settings(settings)
            fwd = forward(view, n(means3D), n(cov3D), n(opacities), n(shs), n(colors_precomp), n(features))
            ctx.view, ctx.fwd = view, fwd
            ctx.save_for_backward(means3D, cov3D, opacities,
                                  shs if shs is not None else torch.empty(0),
                                  colors_precomp if colors_precomp is not None else torch.empty(0),
                                  features if features is not None else torch.empty(0))
            ctx.flags = (shs is not None, colors_precomp is not None, features is not None)
            t = lambda a: None if a is None else torch.from_numpy(a)
            color, feat = t(fwd["color"]), t(fwd["feature"])
            mask, depth = t(fwd["mask"])[None], t(fwd["depth"])[None]
            radii = torch.from_numpy(fwd["radii"])
            outs = (color if color is not None else torch.empty(0),
                    feat if feat is not None else torch.empty(0), mask, depth, radii)
            ctx.mark_non_differentiable(radii)
            return outs

        @staticmethod
        def backward(ctx, g_color, g_feat, g_mask, g_depth, _g_radii):
            means3D, cov3D, opacities, shs, cp, feats = ctx.saved_tensors
            has_shs, has_cp, has_f = ctx.flags
            n = lambda t: t.detach().cpu().float().numpy()
            g = backward(ctx.view, n(means3D), n(cov3D), n(opacities), n(shs) if has_shs else None,
                         n(cp) if has_cp else None, n(feats) if has_f else None, ctx.fwd,
                         n(g_color) if (has_shs or has_cp) and g_color is not None else None,
                         n(g_feat) if has_f and g_feat is not None else None,
                         n(g_mask[0]) if g_mask is not None else None,
                         n(g_depth[0]) if g_depth is not None else None)
            t = lambda a: None if a is None else torch.from_numpy(a)
            return (t(g["means3D"]), t(g["means2D"]), t(g["shs"]) if has_shs else None,
                    t(g["colors_precomp"]) if has_cp else None, t(g["features"]) if has_f else None,
                    t(g["opacities"]), t(g["cov3D"]), None)

    return _OracleRasterize


_FN = None


class GaussianRasterizer:
    """Same call signature / 5-tuple return as the reference's external rasterizer."""

    def __new__(cls, raster_settings):
        torch = _torch()

        class _Mod(torch.nn.Module):
            def __init__(self, rs):
                super().__init__()
                self.raster_settings = rs

            def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None,
                        features=None, scales=None, rotations=None, cov3D_precomp=None):
                global _FN
                if _FN is None:
                    _FN = make_function()
                if shs is not None and colors_precomp is not None:
                    raise Exception("Please provide at most one of SHs / precomputed colors")
                if cov3D_precomp is None:
                    raise Exception("oracle supports cov3D_precomp only (the reference never passes scales/rotations)")
                color, feat, mask, depth, radii = _FN.apply(
                    means3D, means2D, shs, colors_precomp, features, opacities, cov3D_precomp,
                    self.raster_settings)
                return (color if color.numel() else None, feat if feat.numel() else None, mask, depth, radii)

        return _Mod(raster_settings)
