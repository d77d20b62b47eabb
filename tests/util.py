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
from latentsplat_amd.decoder import geometry  # noqa: E402
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

    def __init__(self, bi: dict, device="cuda", shared_means=False, forward_flags=0):
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
        # the feature stride declared below assumes a (V,G,C) tensor; a (1,G,C) one would be read out of bounds
        assert self.features is None or self.features.shape[0] == V, \
            f"HipRun: features have leading dim {self.features.shape[0]}, expected {V} views"
        self.d = d = Dims(V, G, H, W, Cf, mode, bi["sh_degree"], K, 0 if shared_means else 3 * G,
                          0 if shared_means else 6 * G, 0, 0, Cf * G if Cf else 0, 6, 0, 0, 0, 0, 0, 0, forward_flags)
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
        if self.layout.geom_bin_stride == 12:   # narrow records: u8 tile coordinates, f32 depth, u8 span[4]
            b = self._view(self.geom, self.layout.geom_bin, n * 12, torch.uint8).cpu().numpy().astype(np.int32).reshape(
                self.d.num_views, self.d.num_gaussians, 12)
            return b[:, :, :4]
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

    def half_count(self):
        """(V*T, 2) lengths of the half-tile render lists."""
        n = self.d.num_views * self.T * 2
        return self._view(self.geom, self.layout.geom_half_count, n * 4, torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 2)

    def half_list(self):
        """Raw half-list area: [2P] words `index | sub-block bits << 24` (tile with canonical list [s, s+n)
        owns [2s, 2s+2n), half h at 2s + h*n)."""
        return self._view(self.bin, self.layout.bin_half_list, max(self.P, 0) * 8, torch.int32).cpu().numpy().view(np.uint32)

    def n_considered(self):
        """n_contrib translated from positions in the HALF-TILE render lists (what the kernels keep) to
        positions in the canonical tile lists (what the published algorithm / the oracle keeps): a pixel
        that considered all of its half's list considered the whole tile list; otherwise it stopped
        at a specific entry, whose canonical position is looked up."""
        nc = self.n_contrib()
        ts, pl, hc, hl = self.tile_start(), self.point_list(), self.half_count(), self.half_list()
        V, H, W, T = self.d.num_views, self.d.height, self.d.width, self.T
        gx = (W + 15) // 16
        out = np.zeros_like(nc)
        for v in range(V):
            for t in range(T):
                s0, s1 = ts[v * T + t], ts[v * T + t + 1]
                n = s1 - s0
                ty, tx = divmod(t, gx)
                canon = pl[s0:s1]
                pos = None
                for h in range(2):
                    y0, x0 = ty * 16 + 8 * h, tx * 16
                    if y0 >= H:
                        continue
                    blk = nc[v, y0:y0 + 8, x0:x0 + 16]
                    cnt = hc[v * T + t, h]
                    lst = hl[2 * s0 + h * n: 2 * s0 + h * n + cnt] & 0x00FFFFFF
                    res = np.full(blk.shape, n, np.int64)
                    stopped = blk < cnt
                    if stopped.any():
                        if pos is None:
                            pos = {int(g): i for i, g in enumerate(canon)}
                        res[stopped] = [pos[int(lst[k])] for k in blk[stopped]]
                    out[v, y0:y0 + 8, x0:x0 + 16] = res
        return out

    def item_flags(self):
        """((V*T, 2) flags of the half-tile items, kHdrFlagsValid) as the forward compositing kernel left them."""
        n = self.d.num_views * self.T * 2
        f = self._view(self.geom, self.layout.geom_item_flags, n * 4, torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 2)
        hdr = self._view(self.geom, self.layout.geom_header, 32, torch.int32).cpu().numpy()
        return f, int(hdr[6])

    def n_contrib(self):
        n = self.d.num_views * self.d.height * self.d.width
        return self._view(self.img, self.layout.img_n_contrib, n * 4, torch.int32).cpu().numpy().reshape(
            self.d.num_views, self.d.height, self.d.width)

    def final_T(self):
        n = self.d.num_views * self.d.height * self.d.width
        return self._view(self.img, self.layout.img_final_T, n * 4, torch.float32).cpu().numpy().reshape(
            self.d.num_views, self.d.height, self.d.width)


def psnr(ground_truth: np.ndarray, predicted: np.ndarray) -> float:
    """PSNR of one (C,H,W) image in dB, as the reference's evaluation computes it (src/evaluation/metrics.py:13-20: both
    images clipped to [0, 1], mean squared error over channels and pixels, -10 log10).  Used to report the HIP render
    against the oracle's render of the same scene (BASELINE configs[4] names "PSNR vs reference")."""
    gt = np.clip(np.asarray(ground_truth, dtype=np.float64), 0.0, 1.0)
    pr = np.clip(np.asarray(predicted, dtype=np.float64), 0.0, 1.0)
    mse = float(((gt - pr) ** 2).mean())
    return float("inf") if mse == 0.0 else -10.0 * float(np.log10(mse))


PSNR_LOG: list = []     # (what, dB) records; tests/conftest.py prints them and adds them to parity_accounting.json


# ---- accounting of what the parity assertions actually held to the bar -------------------------
# Every call appends one record; tests/conftest.py prints the table at the end of the session and
# writes it to tests/_parity_accounting.json (so a reader sees how many pixels / rows were exempt).
ACCOUNTING: list = []


def _account(kind, what, total, strict, bounded, exempt, tol, scale, worst):
    ACCOUNTING.append(dict(kind=kind, what=what, total=int(total), held_to_bar=int(strict), flip_bounded=int(bounded),
                           exempt=int(exempt), tol=float(tol), scale=float(scale), worst_strict_err=float(worst)))


def assert_close_except_fragile(got, want, oracle_fwd, atol, what="", max_fragile_frac=0.02, flip_bound=2e-2, scale=1.0):
    """|got - want| <= atol on every pixel except those where the oracle saw an evaluation within
    float rounding of one of the algorithm's discontinuities (alpha == 1/255 skip, T == 1e-4 stop):
    there two correct float implementations may legitimately take different branches, which moves
    the pixel by up to alpha*T*c ~ 4e-3 (bounded at `flip_bound` = 2e-2 for unit-range channels; depth passes the bound
    in scene units).  Only such pixels may miss `atol` (at most
    2 % of the image may even be candidates); fragile pixels that meet the bar anyway count as held to
    it.  The counts are recorded in ACCOUNTING."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want).reshape(-1, got.shape[-2], got.shape[-1]).max(0)
    frag = np.unique(oracle_fwd["fragile"][:, 0]) if len(oracle_fwd["fragile"]) else np.zeros(0, np.int64)
    assert not oracle_fwd["fragile_overflow"], f"{what}: fragile list overflow"
    assert len(frag) <= max(8, int(err.size * max_fragile_frac)), f"{what}: {len(frag)} of {err.size} pixels fragile (> {max_fragile_frac:.0%})"
    flat = err.reshape(-1)
    miss = flat > atol
    allowed = np.zeros(flat.size, bool)
    allowed[frag] = True
    stray = miss & ~allowed
    assert not stray.any(), f"{what}: {int(stray.sum())} non-fragile pixels off by more than {atol:.1e} (max abs err {flat[stray].max():.3e})"
    assert flat[miss].max(initial=0) <= flip_bound, f"{what}: fragile pixel moved more than one alpha step"
    # (`scale`: what `atol` was multiplied with — depth images are held to 1e-4 of the largest rendered depth, float32 sums
    # of values of 20-80; north_star's absolute 1e-4 is for latent / RGB — so that the accounting prints err / scale)
    _account("image", what, flat.size, flat.size - int(miss.sum()), int(miss.sum()), 0, atol / scale, scale, flat[~miss].max(initial=0))


def fragile_gaussians(oracle_fwd, W):
    """(direct, behind) for the gradient assertions.
    direct: Gaussians of the near-discontinuity evaluations themselves (either decision is correct;
            their own gradient moves by a flip-sized amount).
    behind: every OTHER Gaussian that contributes (alpha >= 1/255, power <= 0) to a pixel holding such
            an evaluation — its transmittance / accumulated-behind term changes if the decision flips.
            Gaussians of the pixel's tile list that do not reach the pixel are NOT affected and stay on
            the strict bar (round 1 exempted the whole tile list)."""
    fr = oracle_fwd["fragile"]
    if len(fr) == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    gx = (W + 15) // 16
    xy, co = oracle_fwd["xy"].astype(np.float64), oracle_fwd["conic_opacity"].astype(np.float64)
    behind = []
    for pix in np.unique(fr[:, 0]):
        px, py = float(pix % W), float(pix // W)
        tile = (pix // W // 16) * gx + (pix % W) // 16
        s0, s1 = oracle_fwd["ranges"][tile]
        lst = oracle_fwd["point_list"][s0:s1].astype(np.int64)
        dx, dy = xy[lst, 0] - px, xy[lst, 1] - py
        power = -0.5 * (co[lst, 0] * dx * dx + co[lst, 2] * dy * dy) - co[lst, 1] * dx * dy
        alpha = np.minimum(0.99, co[lst, 3] * np.exp(np.minimum(power, 0.0)))
        behind.append(lst[(power <= 1e-6) & (alpha >= 1.0 / 255.0 - 1e-5)])
    direct = np.unique(fr[:, 1]).astype(np.int64)
    behind = np.setdiff1d(np.unique(np.concatenate(behind)), direct)
    return direct, behind


def assert_grad_close_except_fragile(got, want, direct, behind, tol, what="", min_strict=0.95, clean_tol=None, row_tol=None,
                                     max_direct_frac=0.01):
    """Per-Gaussian gradient rows against the oracle.  scale = max(1, max |want|) over the tensor.

    * Rows with no fragile evaluation nearby ("clean" rows — all but a fraction of a percent) are held to
      ``clean_tol * scale`` (default: ``tol``) and, when ``row_tol`` is given, ALSO to the per-row mixed bar
      ``row_tol * max(1, |want_row|_inf)`` — a small-magnitude row is then not hidden by the tensor's largest entry.
    * A row may miss the ``tol * scale`` bar ONLY if it belongs to a fragile evaluation (`direct`: either
      decision is correct, exempt) or shares a pixel with one (`behind`: flip-sized bound of 5e-3 * scale).
    * At least `min_strict` of all rows must be within ``tol * scale``.
    ACCOUNTING records the worst CLEAN row (err / scale and per-row mixed), i.e. the real arithmetic error of the
    kernels, next to the counts.  (tools/grad_budget.py splits that error further against a float64 evaluation.)"""
    got = np.asarray(got, np.float64).reshape(np.shape(want)[0], -1)
    want = np.asarray(want, np.float64).reshape(got.shape)
    scale = max(1.0, np.abs(want).max())
    err = np.abs(got - want).max(1)
    n = err.size
    assert len(direct) <= max(8, int(n * max_direct_frac)), f"{what}: too many fragile Gaussians ({len(direct)} of {n})"
    miss = err > tol * scale
    allowed = np.zeros(n, bool)
    allowed[direct] = True
    allowed[behind] = True
    stray = miss & ~allowed
    assert not stray.any(), (f"{what}: {int(stray.sum())} rows off the {tol:.0e} bar without a fragile evaluation nearby; "
                             f"max err {err[stray].max():.3e} (scale {scale:.3e}, row {int(np.argmax(np.where(stray, err, 0)))})")
    clean = ~allowed
    ctol = tol if clean_tol is None else clean_tol
    worst_clean = err[clean].max(initial=0)
    assert worst_clean <= ctol * scale, f"{what}: clean row off by {worst_clean:.3e} > {ctol:.0e} * scale ({scale:.3e})"
    row_mixed = err / np.maximum(1.0, np.abs(want).max(1))
    worst_row = row_mixed[clean].max(initial=0)
    if row_tol is not None:
        assert worst_row <= row_tol, f"{what}: clean row off by {worst_row:.3e} of max(1, |row|) > {row_tol:.0e}"
    is_direct = np.zeros(n, bool)
    is_direct[direct] = True
    bounded = miss & ~is_direct
    assert err[bounded].max(initial=0) <= 5e-3 * scale, f"{what}: {err[bounded].max():.3e} next to a fragile evaluation"
    n_miss = int(miss.sum())
    assert n - n_miss >= min_strict * n or n_miss <= 16, \
        f"{what}: only {n - n_miss} of {n} rows within the {tol:.0e} bar (< {min_strict:.0%})"
    _account("grad", what, n, n - n_miss, int(bounded.sum()), int((miss & is_direct).sum()), tol, scale, worst_clean)
    ACCOUNTING[-1].update(clean_rows=int(clean.sum()), worst_clean_row_mixed=float(worst_row), clean_tol=float(ctol),
                          row_tol=None if row_tol is None else float(row_tol))


def to_boundary(views, v, means3D, cov3D_precomp, opacities, shs, colors_precomp, features, feature_sh,
                shs_channel_major):
    """Host statement of what the kernels fuse: (scene-level inputs of view v) -> the tensors the
    reference hands to its rasterizer (scaled means, packed scaled covariance, (G,K,3) colour SH,
    evaluated latent features).  Pure torch, differentiable; used by the CPU stand-in below and by
    the gradient tests to chain oracle gradients back to scene-level inputs."""
    V = views.shape[0]
    # leading dim: one slice per view, or per scene (group of V / S consecutive views), or shared
    pv = lambda t, base: None if t is None else (t[v * t.shape[0] // V] if t.dim() == base + 1 else t)
    vw = views[v]
    scale = vw[40]
    means = pv(means3D, 2) * scale
    cov = cov3D_precomp
    full = cov.shape[-2:] == (3, 3)
    cov = pv(cov, 3 if full else 2)
    cov6 = (cs._pack_covariances(cov) if full else cov) * (scale * scale)   # same rounding as the kernel / reference
    sh = pv(shs, 3)
    if sh is not None and shs_channel_major:
        sh = sh.transpose(-1, -2)
    feats = pv(features, 2)
    if feature_sh is not None:
        fsh = pv(feature_sh, 3)
        from math import isqrt
        d = means - vw[32:35][None]
        d = d / d.norm(dim=-1, keepdim=True)
        feats = 0.5 + geometry.eval_sh(isqrt(fsh.shape[-1]) - 1, fsh, d)
    return means, cov6, pv(opacities, 2), sh, pv(colors_precomp, 2), feats


def oracle_rasterize_views(views, image_height, image_width, sh_degree, means3D, cov3D_precomp, opacities,
                           shs=None, colors_precomp=None, features=None, means2D=None, debug=False,
                           feature_sh=None, shs_channel_major=False):
    """CPU stand-in with the signature of latentsplat_amd.rasterizer.rasterize_views, served by the
    oracle.  TESTS ONLY: lets the not-gpu suite pin the host-side wrapper logic."""
    V = views.shape[0]
    n = lambda t: None if t is None else t.detach().cpu().float().contiguous().numpy()
    cols, feats, masks, depths, radii = [], [], [], [], []
    for v in range(V):
        vw = views[v].detach().cpu()
        view = orc.View(image_height, image_width, float(vw[35]), float(vw[36]), vw[37:40].numpy(),
                        vw[0:16].numpy().reshape(4, 4), vw[16:32].numpy().reshape(4, 4), vw[32:35].numpy(), sh_degree)
        m, c6, op, sh, cp, ft = to_boundary(views.detach().cpu(), v, means3D.detach().cpu(), cov3D_precomp.detach().cpu(),
                                            opacities.detach().cpu(), None if shs is None else shs.detach().cpu(),
                                            None if colors_precomp is None else colors_precomp.detach().cpu(),
                                            None if features is None else features.detach().cpu(),
                                            None if feature_sh is None else feature_sh.detach().cpu(), shs_channel_major)
        o = orc.forward(view, n(m), n(c6), n(op), n(sh), n(cp), n(ft), keep_intermediates=False)
        cols.append(o["color"]); feats.append(o["feature"]); masks.append(o["mask"]); depths.append(o["depth"])
        radii.append(o["radii"])
    st = lambda xs: None if xs[0] is None else torch.from_numpy(np.stack(xs))
    return st(cols), st(feats), st(masks), st(depths), st(radii)
