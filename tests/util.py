"""Shared helpers for the parity tests: synthetic scene -> rasterizer-boundary inputs, oracle
runs, and access to the HIP library's workspaces for bit-exact intermediate comparisons."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from latentsplat_amd.decoder import cuda_splatting as cs  # noqa: E402
from latentsplat_amd.decoder.geometry import get_fov  # noqa: E402
from latentsplat_amd.synthetic import Scene, make_scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def boundary_inputs(scene: Scene, H: int, W: int, use_sh: bool = True, bg=(0.0, 0.0, 0.0)):
    """Everything that crosses the rasterizer boundary for each view of ``scene`` (CPU tensors),
    computed by the package's own wrapper math (pinned against the reference in
    test_golden_boundary.py).  Returns dict with per-view stacked tensors."""
    V = scene.extrinsics.shape[0]
    means = scene.means[None].expand(V, -1, -1)
    covs = scene.covariances[None].expand(V, -1, -1, -1)
    ext, nr, fr, means, covs = cs._scale_scene(scene.extrinsics, scene.near, scene.far, means, covs)
    fov_x, fov_y = get_fov(scene.intrinsics).unbind(-1)
    cams = cs._cameras(ext, nr, fr, fov_x, fov_y)
    csh = None if scene.color_sh is None else scene.color_sh[None]
    fsh = None if scene.feature_sh is None else scene.feature_sh[None]
    degree, shs, colors_precomp, features = cs._payload(means, cams.campos, csh, fsh, use_sh)
    bgt = torch.tensor(bg, dtype=torch.float32)[None].expand(V, 3)
    return dict(V=V, H=H, W=W, sh_degree=degree, cams=cams, bg=bgt,
                means=means.contiguous(), cov6=cs._pack_covariances(covs).contiguous(),
                opac=scene.opacities[:, None].contiguous(),
                shs=None if shs is None else shs[0].contiguous(),
                colors_precomp=None if colors_precomp is None else colors_precomp[0].contiguous(),
                features=None if features is None else features.contiguous())


def oracle_view(bi: dict, v: int) -> orc.View:
    c = bi["cams"]
    return orc.View(bi["H"], bi["W"], float(c.tan_fov_x[v]), float(c.tan_fov_y[v]), bi["bg"][v].numpy(),
                    c.view_matrix[v].contiguous().numpy(), c.full_projection[v].contiguous().numpy(),
                    c.campos[v].contiguous().numpy(), bi["sh_degree"])


def oracle_forward(bi: dict, v: int):
    n = lambda t: None if t is None else t.detach().numpy()
    feats = None if bi["features"] is None else n(bi["features"][v])
    return orc.forward(oracle_view(bi, v), n(bi["means"][v]), n(bi["cov6"][v]), n(bi["opac"]),
                       n(bi["shs"]), n(bi["colors_precomp"]), feats)


def oracle_backward(bi: dict, v: int, fwd: dict, g_color=None, g_feat=None, g_mask=None, g_depth=None):
    n = lambda t: None if t is None else t.detach().numpy()
    feats = None if bi["features"] is None else n(bi["features"][v])
    return orc.backward(oracle_view(bi, v), n(bi["means"][v]), n(bi["cov6"][v]), n(bi["opac"]),
                        n(bi["shs"]), n(bi["colors_precomp"]), feats, fwd, g_color, g_feat, g_mask, g_depth)


def view_table(bi: dict, device) -> torch.Tensor:
    from latentsplat_amd.rasterizer import make_view_table
    c = bi["cams"]
    return make_view_table(c.view_matrix.to(device), c.full_projection.to(device), c.campos.to(device),
                           c.tan_fov_x.to(device), c.tan_fov_y.to(device), bi["bg"].to(device))


class HipRun:
    """Runs the HIP forward through the C ABI directly (no autograd) and keeps the workspaces so
    tests can read intermediates back."""

    def __init__(self, bi: dict, device="cuda", shared_means=False):
        from latentsplat_amd import _lib
        from latentsplat_amd._lib import Dims, Inputs, Layout, Outputs
        self.lib = lib = _lib.load()
        dev = torch.device(device)
        V, H, W = bi["V"], bi["H"], bi["W"]
        self.views = view_table(bi, dev)
        t = lambda x: None if x is None else x.to(dev).contiguous()
        self.means = t(bi["means"][0] if shared_means else bi["means"])
        self.cov6 = t(bi["cov6"][0] if shared_means else bi["cov6"])
        self.opac, self.features = t(bi["opac"]), t(bi["features"])
        self.color = t(bi["shs"]) if bi["shs"] is not None else t(bi["colors_precomp"])
        G = bi["means"].shape[1]
        Cf = 0 if self.features is None else self.features.shape[-1]
        mode = 1 if bi["shs"] is not None else (2 if bi["colors_precomp"] is not None else 0)
        K = bi["shs"].shape[1] if bi["shs"] is not None else 0
        self.d = d = Dims(V, G, H, W, Cf, mode, bi["sh_degree"], K, 0 if shared_means else 3 * G,
                          0 if shared_means else 6 * G, 0, 0, Cf * G if Cf else 0)
        p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        self.inp = Inputs(p(self.views), p(self.means), p(self.cov6), p(self.opac), p(self.color), p(self.features))
        u8 = dict(dtype=torch.uint8, device=dev)
        self.geom = torch.zeros(lib.lsr_geom_workspace_bytes(C.byref(d)), **u8)
        self.img = torch.zeros(lib.lsr_image_workspace_bytes(C.byref(d)), **u8)
        self.radii = torch.zeros((V, G), dtype=torch.int32, device=dev)
        npairs, maxtile = C.c_int64(0), C.c_int32(0)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.lsr_forward_prepare(C.byref(d), C.byref(self.inp), p(self.geom), p(self.radii),
                                           C.byref(npairs), C.byref(maxtile), stream), "prepare")
        self.P, self.maxtile = npairs.value, maxtile.value
        self.bin = torch.zeros(max(1, lib.lsr_binning_workspace_bytes(C.byref(d), self.P, self.maxtile)), **u8)
        f32 = dict(dtype=torch.float32, device=dev)
        self.color_out = torch.zeros((V, 3, H, W), **f32) if mode else None
        self.feat_out = torch.zeros((V, Cf, H, W), **f32) if Cf else None
        self.mask_out = torch.zeros((V, H, W), **f32)
        self.depth_out = torch.zeros((V, H, W), **f32)
        outs = Outputs(p(self.color_out), p(self.feat_out), p(self.mask_out), p(self.depth_out), p(self.radii))
        _lib.check(lib.lsr_forward_render(C.byref(d), C.byref(self.inp), p(self.geom), p(self.bin), p(self.img),
                                          self.P, self.maxtile, C.byref(outs), stream), "render")
        torch.cuda.synchronize(dev)
        self.layout = Layout()
        _lib.check(lib.lsr_get_layout(C.byref(d), self.P, C.byref(self.layout)), "layout")
        self.T = ((W + 15) // 16) * ((H + 15) // 16)

    def _view(self, ws, off, nbytes, dtype):
        return ws[off:off + nbytes].view(dtype)

    def tile_start(self):
        n = self.d.num_views * self.T + 1
        return self._view(self.geom, self.layout.geom_tile_start, n * 4, torch.int32).cpu().numpy().astype(np.int64)

    def point_list(self):
        return self._view(self.bin, self.layout.bin_point_list, max(self.P, 0) * 4, torch.int32).cpu().numpy().astype(np.int64)

    def rect(self):
        n = self.d.num_views * self.d.num_gaussians
        b = self._view(self.geom, self.layout.geom_bin, n * 16, torch.int16).cpu().numpy().astype(np.int32).reshape(
            self.d.num_views, self.d.num_gaussians, 8)
        return b[:, :, :4] & 0xFFFF

    def q(self):
        """(x,y,A,B), (C,o,z,clampbits) of every (view, Gaussian) screen-space record."""
        n = self.d.num_views * self.d.num_gaussians
        rf = self.layout.geom_rec_floats
        r = self._view(self.geom, self.layout.geom_rec, n * rf * 4, torch.float32).cpu().numpy().reshape(
            self.d.num_views, -1, rf)
        return r[:, :, 0:4], r[:, :, 4:8]

    def n_contrib(self):
        n = self.d.num_views * self.d.height * self.d.width
        return self._view(self.img, self.layout.img_n_contrib, n * 4, torch.int32).cpu().numpy().reshape(
            self.d.num_views, self.d.height, self.d.width)

    def final_T(self):
        n = self.d.num_views * self.d.height * self.d.width
        return self._view(self.img, self.layout.img_final_T, n * 4, torch.float32).cpu().numpy().reshape(
            self.d.num_views, self.d.height, self.d.width)


def oracle_rasterize_views(views, image_height, image_width, sh_degree, means3D, cov3D_precomp, opacities,
                           shs=None, colors_precomp=None, features=None, means2D=None, debug=False):
    """CPU stand-in with the signature of latentsplat_amd.rasterizer.rasterize_views, served by the
    oracle.  TESTS ONLY: lets the not-gpu suite pin the host-side wrapper logic."""
    V = views.shape[0]
    n = lambda t: None if t is None else t.detach().cpu().float().numpy()
    per_view = lambda t, base, v: None if t is None else (t[v] if t.dim() == base + 1 else t)
    cols, feats, masks, depths, radii = [], [], [], [], []
    for v in range(V):
        vw = views[v].detach().cpu()
        view = orc.View(image_height, image_width, float(vw[35]), float(vw[36]), vw[37:40].numpy(),
                        vw[0:16].numpy().reshape(4, 4), vw[16:32].numpy().reshape(4, 4), vw[32:35].numpy(), sh_degree)
        o = orc.forward(view, n(per_view(means3D, 2, v)), n(per_view(cov3D_precomp, 2, v)),
                        n(per_view(opacities, 2, v)), n(per_view(shs, 3, v)), n(per_view(colors_precomp, 2, v)),
                        n(per_view(features, 2, v)), keep_intermediates=False)
        cols.append(o["color"]); feats.append(o["feature"]); masks.append(o["mask"]); depths.append(o["depth"])
        radii.append(o["radii"])
    st = lambda xs: None if xs[0] is None else torch.from_numpy(np.stack(xs))
    return st(cols), st(feats), st(masks), st(depths), st(radii)
