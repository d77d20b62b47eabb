"""GPU tests of the drop-in surfaces above the C ABI: the `diff_gaussian_rasterization` module, the
decoder mirror (against vectors produced by the reference's own wrapper), edge cases, and
size-independent properties at BASELINE's full size."""
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _t(a, dev):
    return torch.from_numpy(np.asarray(a)).to(dev)


# ------------------------------------------------------------------------------------------
# decoder surface vs the reference wrapper (+ oracle) golden outputs
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,fkw,variational", [
    ("train_rgb4_feat4", {}, False),
    ("features_only", dict(return_colors=False), False),
    ("variational_8ch", {}, True),
    ("disparity_depth", dict(depth_mode="disparity"), False),
])
def test_decoder_matches_reference_wrapper_outputs(hip_device, name, fkw, variational):
    from latentsplat_amd import decoder as dec
    g = np.load(os.path.join(GOLD, f"boundary_{name}.npz"))
    want = np.load(os.path.join(GOLD, f"decoder_{name}.npz"))
    dev = hip_device
    gauss = dec.Gaussians(_t(g["in_means"], dev), _t(g["in_covariances"], dev), _t(g["in_opacities"], dev),
                          _t(g["in_color_harmonics"], dev), _t(g["in_feature_harmonics"], dev))
    d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), [float(x) for x in g["in_bg"]], variational).to(dev)
    out = d.forward(gauss, _t(g["in_extrinsics"], dev), _t(g["in_intrinsics"], dev), _t(g["in_near"], dev),
                    _t(g["in_far"], dev), tuple(int(x) for x in g["in_image_shape"]), **fkw)
    np.testing.assert_allclose(out.mask.cpu().numpy(), want["mask"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out.depth.cpu().numpy(), want["depth"], atol=1e-4 * max(1.0, np.abs(want["depth"]).max()), rtol=0)
    if "color" in want:
        np.testing.assert_allclose(out.color.cpu().numpy(), want["color"], atol=1e-4, rtol=0)
    else:
        assert out.color is None
    np.testing.assert_allclose(out.feature_posterior.mean.cpu().numpy(), want["posterior_mean"], atol=1e-4, rtol=0)
    # logvar = log(1 - mask) (or the rendered logvar half): compared where it is well conditioned through
    # what it encodes — exp(logvar) = 1 - mask must agree to the same 1e-4 as the mask itself, everywhere
    # (saturated pixels included: exp maps the clamp floor -30 to 0) — and directly away from saturation.
    lv, lw = out.feature_posterior.logvar.cpu().numpy(), want["posterior_logvar"]
    np.testing.assert_allclose(np.exp(np.minimum(lv, 20.0)), np.exp(np.minimum(lw, 20.0)), atol=1e-4 * max(1.0, float(np.exp(np.minimum(lw, 20.0)).max())), rtol=0)
    sel = lw > -4   # d logvar = d mask / (1 - mask): 1e-4 in the mask is < 6e-3 here
    np.testing.assert_allclose(lv[sel], lw[sel], atol=6e-3, rtol=0)


# ------------------------------------------------------------------------------------------
# module-name drop-in: per-view GaussianRasterizer exactly as the reference calls it
# ------------------------------------------------------------------------------------------
def _single_view_settings(bi, v, dev, debug=False):
    import diff_gaussian_rasterization as dgr
    c = bi["cams"]
    return dgr.GaussianRasterizationSettings(
        image_height=bi["H"], image_width=bi["W"], tanfovx=c.tan_fov_x[v].item(), tanfovy=c.tan_fov_y[v].item(),
        bg=bi["bg"][v].to(dev), scale_modifier=1.0, viewmatrix=c.view_matrix[v].to(dev),
        projmatrix=c.full_projection[v].to(dev), sh_degree=bi["sh_degree"], campos=c.campos[v].to(dev),
        prefiltered=False, debug=debug)


def test_dropin_rasterizer_reference_call_pattern(hip_device):
    import diff_gaussian_rasterization as dgr
    dev = hip_device
    sc = util.make_scene(3000, image_size=64, views=2, color_sh_degree=4, feature_channels=4, feature_sh_degree=2)
    bi = util.boundary_inputs(sc, 64, 64, bg=(0.1, 0.2, 0.3))
    for v in range(2):
        means = bi["means"][v].to(dev).requires_grad_(True)
        mean_gradients = torch.zeros_like(means, requires_grad=True)
        rasterizer = dgr.GaussianRasterizer(_single_view_settings(bi, v, dev, debug=(v == 1)))
        image, feature_map, mask, depth_map, radii = rasterizer(
            means3D=means, means2D=mean_gradients, shs=bi["shs"].to(dev), colors_precomp=None,
            features=bi["features"][v].to(dev), opacities=bi["opac"].to(dev), cov3D_precomp=bi["cov6"][v].to(dev))
        o = util.oracle_forward(bi, v)
        assert image.shape == (3, 64, 64) and feature_map.shape == (4, 64, 64)
        assert mask.shape == (1, 64, 64) and depth_map.shape == (1, 64, 64) and radii.shape == (3000,)
        np.testing.assert_allclose(image.detach().cpu().numpy(), o["color"], atol=1e-4)
        np.testing.assert_allclose(feature_map.detach().cpu().numpy(), o["feature"], atol=1e-4)
        np.testing.assert_array_equal(radii.cpu().numpy(), o["radii"])
        (image.sum() + feature_map.sum()).backward()
        b = util.oracle_backward(bi, v, o, np.ones_like(o["color"]), np.ones_like(o["feature"]))
        scale = max(1.0, np.abs(b["means3D"]).max())
        assert np.abs(means.grad.cpu().numpy() - b["means3D"]).max() <= 1e-4 * scale
        assert np.abs(mean_gradients.grad.cpu().numpy() - b["means2D"]).max() <= 1e-4 * max(1.0, np.abs(b["means2D"]).max())


def test_dropin_argument_combinations(hip_device):
    import diff_gaussian_rasterization as dgr
    dev = hip_device
    sc = util.make_scene(1500, image_size=48, views=1, color_sh_degree=0, feature_channels=4)
    bi = util.boundary_inputs(sc, 48, 48, use_sh=False)   # colors_precomp + raw features
    r = dgr.GaussianRasterizer(_single_view_settings(bi, 0, dev))
    kw = dict(means3D=bi["means"][0].to(dev), means2D=torch.zeros(1500, 3, device=dev), opacities=bi["opac"].to(dev),
              cov3D_precomp=bi["cov6"][0].to(dev))
    o = util.oracle_forward(bi, 0)
    # colours only / features only / both
    img, feat, mask, depth, _ = r(colors_precomp=bi["colors_precomp"].to(dev), **kw)
    assert feat is None
    np.testing.assert_allclose(img.cpu().numpy(), o["color"], atol=1e-4)
    img, feat, mask, depth, _ = r(features=bi["features"][0].to(dev), **kw)
    assert img is None
    np.testing.assert_allclose(feat.cpu().numpy(), o["feature"], atol=1e-4)
    np.testing.assert_allclose(mask[0].cpu().numpy(), o["mask"], atol=1e-4)
    # upstream argument checks
    with pytest.raises(Exception, match="at most one"):
        r(shs=torch.zeros(1500, 1, 3, device=dev), colors_precomp=bi["colors_precomp"].to(dev), **kw)
    kw2 = dict(kw); kw2.pop("cov3D_precomp")
    with pytest.raises(Exception, match="exactly one"):
        r(features=bi["features"][0].to(dev), **kw2)
    # scales / rotations instead of a precomputed covariance: identity rotation, isotropic scale s
    s = 0.02
    scales = torch.full((1500, 3), s, device=dev)
    rots = torch.tensor([1.0, 0, 0, 0], device=dev).repeat(1500, 1)
    a = r(features=bi["features"][0].to(dev), scales=scales, rotations=rots, **kw2)[1]
    iso = torch.zeros(1500, 6, device=dev); iso[:, [0, 3, 5]] = s * s
    kw3 = dict(kw); kw3["cov3D_precomp"] = iso
    b = r(features=bi["features"][0].to(dev), **kw3)[1]
    assert torch.allclose(a, b, atol=1e-6)


# ------------------------------------------------------------------------------------------
# edge cases
# ------------------------------------------------------------------------------------------
def test_empty_and_fully_culled_scenes(hip_device):
    from latentsplat_amd.rasterizer import rasterize_views
    dev = hip_device
    sc = util.make_scene(10, image_size=32, views=2, color_sh_degree=1, feature_channels=4)
    bi = util.boundary_inputs(sc, 32, 32, bg=(0.3, 0.5, 0.7))
    views = util.view_table(bi, dev)
    z = lambda *s: torch.zeros(*s, device=dev)
    color, feat, mask, depth, radii = rasterize_views(views, 32, 32, 1, z(0, 3), z(0, 6), z(0, 1), shs=z(0, 4, 3), features=z(0, 4))
    assert radii.shape == (2, 0) and float(mask.abs().max()) == 0 and float(feat.abs().max()) == 0
    assert torch.allclose(color[:, :, 5, 5], torch.tensor([0.3, 0.5, 0.7], device=dev).expand(2, 3))
    # all Gaussians behind the camera
    means = bi["means"].to(dev).clone(); means[..., 2] = -1.0
    color, feat, mask, depth, radii = rasterize_views(views, 32, 32, 1, means, bi["cov6"].to(dev), bi["opac"].to(dev),
                                                      shs=bi["shs"].to(dev), features=bi["features"].to(dev))
    assert int(radii.abs().max()) == 0 and float(mask.abs().max()) == 0
    # ... and gradients through an empty render are zeros of the right shape
    m = means.requires_grad_(True)
    out = rasterize_views(views, 32, 32, 1, m, bi["cov6"].to(dev), bi["opac"].to(dev), shs=bi["shs"].to(dev), features=bi["features"].to(dev))
    (out[0].sum() + out[1].sum()).backward()
    assert m.grad.shape == m.shape and float(m.grad.abs().max()) == 0


def test_long_tile_lists_take_the_global_merge_path(hip_device):
    """> 16384 entries on one tile: beyond the LDS sort capacity; lists must stay bit-exact."""
    G = 20_000
    gen = torch.Generator().manual_seed(3)
    sc = util.make_scene(G, image_size=32, views=1, color_sh_degree=None, feature_channels=4, sigma_px=(0.3, 0.6), opacity_scale=0.02)
    # squeeze every Gaussian onto the centre of the 32x32 image (tile (0,0)..(1,1) corner)
    z = sc.means[:, 2]
    sc.means[:, 0] = (torch.rand(G, generator=gen) * 0.1 - 0.30) / 0.8 * z
    sc.means[:, 1] = (torch.rand(G, generator=gen) * 0.1 - 0.30) / 0.8 * z
    bi = util.boundary_inputs(sc, 32, 32)
    run = util.HipRun(bi, hip_device)
    assert run.maxtile > 16384
    o = util.oracle_forward(bi, 0)
    np.testing.assert_array_equal(run.point_list()[:o["P"]], o["point_list"])
    np.testing.assert_allclose(run.feat_out[0].cpu().numpy(), o["feature"], atol=1e-4)
    np.testing.assert_allclose(run.mask_out[0].cpu().numpy(), o["mask"], atol=1e-4)


def test_every_sort_tier_in_one_call(hip_device):
    """One 64x64 view (16 tiles) whose tiles hold ~1.5k, ~6k, ~12k and ~20k entries: the per-tile launch, the two
    persistent LDS variants (lists of 4097..8192 and 8193..16384 keys, collected from both ends of one array) and
    the global merge path all run in ONE call; sorted lists bit-exact, images on the bar — through the synchronous
    forward (later tiers launched because the host knows the longest list) and the no-sync forward (launched always)."""
    from latentsplat_amd.rasterizer import last_forward_status, rasterize_views
    sizes = {0: 1500, 5: 6000, 6: 6500, 10: 12000, 15: 20000}      # tile -> Gaussians squeezed onto its centre
    G = sum(sizes.values()) + 3000
    gen = torch.Generator().manual_seed(9)
    sc = util.make_scene(G, image_size=64, views=1, color_sh_degree=None, feature_channels=4, sigma_px=(0.3, 0.6), opacity_scale=0.02)
    z = sc.means[:, 2]
    at = 0
    for tile, n in sizes.items():
        cx, cy = ((tile % 4) * 16 + 8) / 64.0 - 0.5, ((tile // 4) * 16 + 8) / 64.0 - 0.5
        sl = slice(at, at + n); at += n
        sc.means[sl, 0] = (torch.rand(n, generator=gen) * 0.05 - 0.025 + cx) / 0.8 * z[sl]
        sc.means[sl, 1] = (torch.rand(n, generator=gen) * 0.05 - 0.025 + cy) / 0.8 * z[sl]
    bi = util.boundary_inputs(sc, 64, 64)
    run = util.HipRun(bi, hip_device)
    ts = run.tile_start()
    lens = np.diff(ts)
    assert lens.max() > 16384 and ((lens > 8192) & (lens <= 16384)).any() and ((lens > 4096) & (lens <= 8192)).sum() >= 2 and (lens <= 4096).any()
    o = util.oracle_forward(bi, 0)
    np.testing.assert_array_equal(run.point_list()[:o["P"]], o["point_list"])
    np.testing.assert_allclose(run.feat_out[0].cpu().numpy(), o["feature"], atol=1e-4)
    np.testing.assert_allclose(run.mask_out[0].cpu().numpy(), o["mask"], atol=1e-4)
    # half-tile render lists of the long tiles: every half list is a subsequence of the tile's sorted list
    hc, hl, pl = run.half_count(), run.half_list(), run.point_list()
    for t in np.nonzero(lens > 4096)[0]:
        full = pl[ts[t]:ts[t + 1]]
        for h in range(2):
            n_h = int(hc[t, h])
            mine = hl[2 * ts[t] + h * lens[t]: 2 * ts[t] + h * lens[t] + n_h] & 0x00FFFFFF
            pos = {int(g): i for i, g in enumerate(full)}
            idx = np.array([pos[int(g)] for g in mine])
            assert n_h > 0 and (np.diff(idx) > 0).all(), (t, h)
    # the same scene through the autograd op, synchronous and no-sync: identical images
    dev = hip_device
    views = util.view_table(bi, dev)
    args = (views, 64, 64, 0, bi["means"].to(dev), bi["cov6"].to(dev), bi["opac"].to(dev))
    ref = rasterize_views(*args, features=bi["features"].to(dev))
    st = last_forward_status()
    out = rasterize_views(*args, features=bi["features"].to(dev), pair_capacity=st["num_pairs"] + 64, max_tile_hint=4096)
    assert last_forward_status() == st
    assert torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])
    np.testing.assert_allclose(ref[1][0].cpu().numpy(), o["feature"], atol=1e-4)


def test_identical_depths_sort_by_index(hip_device):
    """Many Gaussians at exactly the same depth: ties must resolve by ascending index."""
    G = 3000
    sc = util.make_scene(G, image_size=32, views=1, color_sh_degree=None, feature_channels=4, opacity_scale=0.05)
    ratio = 5.0 / sc.means[:, 2]
    sc.means = sc.means * ratio[:, None]          # same pixel, depth exactly 5
    sc.means[:, 2] = 5.0
    bi = util.boundary_inputs(sc, 32, 32)
    run = util.HipRun(bi, hip_device)
    o = util.oracle_forward(bi, 0)
    np.testing.assert_array_equal(run.point_list()[:o["P"]], o["point_list"])
    ts = run.tile_start()
    pl = run.point_list()
    for t in range(run.T):
        seg = pl[ts[t]:ts[t + 1]]
        assert np.all(np.diff(seg) > 0)           # equal depth -> strictly ascending indices


def test_orthographic_matches_reference_wrapper_golden(hip_device):
    """render_cuda_orthographic on the MI355X == the RenderOutput the REFERENCE's own
    render_cuda_orthographic (cuda_splatting.py:170-292) produced for the same inputs (with the oracle
    as its rasterizer; tests/golden/make_golden.py::orthographic).  The camera sits ~1.7e3 units behind
    the scene with a 0.1 degree field of view: projection entries ~1e3, depths ~1e3."""
    from latentsplat_amd.decoder import render_cuda_orthographic
    g = np.load(os.path.join(GOLD, "orthographic.npz"))
    dev = hip_device
    t = lambda k: _t(g["in_" + k], dev)
    out = render_cuda_orthographic(
        extrinsics=t("extrinsics"), width=t("width"), height=t("height"), near=t("near"), far=t("far"),
        image_shape=tuple(int(x) for x in g["in_image_shape"]), background_features=t("background_features"),
        gaussian_means=t("gaussian_means"), gaussian_covariances=t("gaussian_covariances"),
        gaussian_opacities=t("gaussian_opacities"), gaussian_color_sh_coefficients=t("gaussian_color_sh_coefficients"),
        gaussian_feature_sh_coefficients=t("gaussian_feature_sh_coefficients"))
    for k in ("color", "feature", "mask", "depth"):
        want = g["out_" + k]
        got = getattr(out, k).cpu().numpy()
        assert got.shape == want.shape
        err = np.abs(got - want)
        tol = 1e-4 * max(1.0, np.abs(want).max())
        # the 1e3-scale projection amplifies float differences of the pixel position: allow isolated
        # pixels at alpha-threshold decisions, everything else on the bar
        assert (err > tol).mean() <= 0.002, f"{k}: {(err > tol).sum()} of {err.size} values off by more than {tol:.1e} (max {err.max():.3e})"
    # the drop-in module accepts the tensor-valued tan fov the reference passes on this path (:260-261)
    import diff_gaussian_rasterization as dgr
    s = dgr.GaussianRasterizationSettings(
        image_height=int(g["in_image_shape"][0]), image_width=int(g["in_image_shape"][1]),
        tanfovx=torch.tensor(float(g["call0_tanfovx"]), device=dev), tanfovy=torch.tensor([float(g["call0_tanfovy"])], device=dev),
        bg=_t(g["call0_bg"], dev), scale_modifier=1.0, viewmatrix=_t(g["call0_viewmatrix"], dev),
        projmatrix=_t(g["call0_projmatrix"], dev), sh_degree=int(g["call0_sh_degree"]), campos=_t(g["call0_campos"], dev),
        prefiltered=False, debug=False)
    means = _t(g["call0_means3D"], dev)
    image, feature_map, mask, depth, _ = dgr.GaussianRasterizer(s)(
        means3D=means, means2D=torch.zeros_like(means), shs=_t(g["call0_shs"], dev), colors_precomp=None,
        features=_t(g["call0_features"], dev), opacities=_t(g["call0_opacities"], dev), cov3D_precomp=_t(g["call0_cov3D_precomp"], dev))
    err = np.abs(image.cpu().numpy() - g["out_color"][0])
    assert (err > 1e-4).mean() <= 0.002 and mask.shape == (1,) + tuple(g["in_image_shape"])
    # ... and those isolated pixels are exactly where the oracle (replayed on the recorded call) sees an evaluation
    # within float rounding of an alpha / transmittance threshold: nowhere else may a value miss the bar
    H, W = (int(x) for x in g["in_image_shape"])
    view = util.orc.View(H, W, float(g["call0_tanfovx"]), float(g["call0_tanfovy"]), np.asarray(g["call0_bg"], np.float32),
                         np.asarray(g["call0_viewmatrix"], np.float32), np.asarray(g["call0_projmatrix"], np.float32),
                         np.asarray(g["call0_campos"], np.float32), int(g["call0_sh_degree"]))
    o = util.orc.forward(view, g["call0_means3D"], g["call0_cov3D_precomp"], g["call0_opacities"], g["call0_shs"], None, g["call0_features"])
    fragile_px = set(int(x) for x in o["fragile"][:, 0]) if len(o["fragile"]) else set()
    for name, got, want in (("color", image, g["out_color"][0]), ("feature", feature_map, g["out_feature"][0]), ("mask", mask[0], g["out_mask"][0])):
        e = np.abs(got.cpu().numpy() - want).reshape(-1, H * W).max(0)
        off = set(int(x) for x in np.flatnonzero(e > 1e-4 * max(1.0, np.abs(want).max())))
        assert off <= fragile_px, f"orthographic {name}: {len(off - fragile_px)} pixels off the bar without a fragile evaluation"


def test_render_cuda_by_name_with_latent_sh(hip_device):
    """``render_cuda`` (reference cuda_splatting.py:56-167) called by name the way every reference caller other
    than the decoder calls it — v-fold replicated per-view inputs INCLUDING ``gaussian_feature_sh_coefficients``
    (src/visualization/validation_in_3d.py:69-82, src/scripts/render_uncertainty.py:249,265 via the decoder):
    bit-for-bit what the scene-major ``render_scenes`` gives for the same scene, and the oracle's images."""
    from latentsplat_amd.decoder import render_cuda
    from latentsplat_amd.decoder.cuda_splatting import render_scenes
    from latentsplat_amd.rasterizer import build_view_table
    dev, V, S = hip_device, 3, 80
    sc = util.make_scene(5000, image_size=S, views=V, color_sh_degree=2, feature_channels=4, feature_sh_degree=2, seed=21)
    d = sc.to(dev)
    bg = torch.tensor([0.2, 0.1, 0.4], device=dev)
    rep = lambda t: t[None].expand(V, *t.shape).contiguous()         # the reference's `repeat` (decoder_splatting_cuda.py:71-87)
    out = render_cuda(d.extrinsics, d.intrinsics, d.near, d.far, (S, S), bg[None].expand(V, 3), rep(d.means), rep(d.covariances),
                      rep(d.opacities), rep(d.color_sh), rep(d.feature_sh))
    ref = render_scenes(d.extrinsics[None], d.intrinsics[None], d.near[None], d.far[None], (S, S), bg, d.means[None],
                        d.covariances[None], d.opacities[None], d.color_sh[None], d.feature_sh[None])
    for k in ("color", "feature", "mask", "depth"):
        assert torch.equal(getattr(out, k), getattr(ref, k)), k
    assert out.color.shape == (V, 3, S, S) and out.feature.shape == (V, 4, S, S)
    views_cpu = build_view_table(d.extrinsics, d.intrinsics, d.near, d.far, bg, True).cpu()
    n = lambda t: None if t is None else t.detach().contiguous().numpy()
    for v in range(V):
        m, c6, op, sh, cp, ft = util.to_boundary(views_cpu, v, sc.means, sc.covariances, sc.opacities[:, None], sc.color_sh, None, None,
                                                 sc.feature_sh, True)
        vw = views_cpu[v]
        view = util.orc.View(S, S, float(vw[35]), float(vw[36]), vw[37:40].numpy(), vw[0:16].numpy().reshape(4, 4),
                             vw[16:32].numpy().reshape(4, 4), vw[32:35].numpy(), 2)
        o = util.orc.forward(view, n(m), n(c6), n(op), n(sh), None, n(ft))
        util.assert_close_except_fragile(out.color[v].cpu().numpy(), o["color"], o, 1e-4, f"render_cuda colour[view {v}]")
        util.assert_close_except_fragile(out.feature[v].cpu().numpy(), o["feature"], o, 1e-4, f"render_cuda latent[view {v}]")
        util.assert_close_except_fragile(out.mask[v].cpu().numpy(), o["mask"], o, 1e-4, f"render_cuda mask[view {v}]")


def test_orthographic_and_depth_modes_run(hip_device):
    from latentsplat_amd.decoder import render_cuda_orthographic, render_depth_cuda
    dev = hip_device
    sc = util.make_scene(2000, image_size=48, views=2, color_sh_degree=1, feature_channels=4).to(dev)
    rep = lambda t: t[None].expand(2, *t.shape).contiguous()
    out = render_cuda_orthographic(sc.extrinsics, torch.full((2,), 4.0, device=dev), torch.full((2,), 4.0, device=dev),
                                   sc.near, sc.far, (48, 48), torch.zeros(2, 3, device=dev), rep(sc.means),
                                   rep(sc.covariances), rep(sc.opacities), rep(sc.color_sh), rep(sc.feature_sh))
    assert out.color.shape == (2, 3, 48, 48) and out.feature.shape == (2, 4, 48, 48)
    assert torch.isfinite(out.color).all() and float(out.mask.max()) <= 1.0
    for mode in ("depth", "disparity", "relative_disparity", "log"):
        dm = render_depth_cuda(sc.extrinsics, sc.intrinsics, sc.near, sc.far, (48, 48), rep(sc.means),
                               rep(sc.covariances), rep(sc.opacities), mode=mode)
        assert dm.shape == (2, 48, 48) and torch.isfinite(dm).all()


def _exempt_fragile_pixels(err_hw, ofw, what="full-size image"):
    """Zero the error of pixels where the oracle saw an evaluation within float rounding of one of
    the algorithm's discontinuities (alpha == 1/255, T == 1e-4): either decision is correct there.
    The number of exempt pixels is bounded (< 0.3 % of the image) and recorded."""
    assert not ofw["fragile_overflow"] and len(ofw["fragile"]) < 200
    pixels = np.unique(ofw["fragile"][:, 0])
    for pix in pixels:
        err_hw[pix // err_hw.shape[1], pix % err_hw.shape[1]] = 0
    util._account("image", what, err_hw.size, err_hw.size - len(pixels), len(pixels), 0, 1e-4, 1.0, err_hw.max())


# ------------------------------------------------------------------------------------------
# full-size, size-independent properties (BASELINE configs[1]/[2]: 300k Gaussians, 256x256, C=4)
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_run(hip_device):
    sc = util.make_scene(300_000, image_size=256, views=2, color_sh_degree=None, feature_channels=4)
    bi = util.boundary_inputs(sc, 256, 256)
    return bi, util.HipRun(bi, hip_device, shared_means=False)


def test_full_size_structure(full_run):
    bi, run = full_run
    ts, pl = run.tile_start(), run.point_list()
    q0, q1 = run.q()
    radii = run.radii.cpu().numpy()
    rect = run.rect()
    assert ts[0] == 0 and ts[-1] == run.P and np.all(np.diff(ts) >= 0)
    area = (rect[..., 2] - rect[..., 0]) * (rect[..., 3] - rect[..., 1])
    assert int(area.sum()) == run.P                                   # checksum of tile rectangles
    assert np.array_equal(area > 0, radii > 0)
    T = run.T
    for v in range(2):
        depth_bits = q1[v][:, 2].view(np.uint32).astype(np.uint64)
        for t in (0, 17, 100, 255, 120, 136):                          # sortedness + membership
            seg = pl[ts[v * T + t]:ts[v * T + t + 1]]
            key = (depth_bits[seg] << np.uint64(32)) | seg.astype(np.uint64)
            assert np.all(np.diff(key.astype(np.float64)) >= 0) and np.all(key[1:] > key[:-1])
            tx, ty = t % 16, t // 16
            r = rect[v][seg]
            assert np.all((r[:, 0] <= tx) & (tx < r[:, 2]) & (r[:, 1] <= ty) & (ty < r[:, 3]))
        # every (Gaussian, tile) pair appears exactly once
        seg_all = pl[ts[v * T]:ts[(v + 1) * T]]
        counts = np.bincount(seg_all, minlength=radii.shape[1])
        assert np.array_equal(counts, area[v])
    m = run.mask_out.cpu().numpy()
    assert m.min() >= 0 and m.max() <= 1 and np.isfinite(run.feat_out.cpu().numpy()).all()
    assert np.allclose(1 - m, run.final_T(), atol=1e-6)


def test_full_size_linearity_and_determinism(hip_device, full_run):
    """Output features are linear in the input features (alpha does not depend on them), the
    render is deterministic, and equals the oracle on one view."""
    from latentsplat_amd.rasterizer import rasterize_views
    bi, run = full_run
    dev = hip_device
    views = util.view_table(bi, dev)
    m, c, o, f = (bi[k].to(dev) for k in ("means", "cov6", "opac", "features"))
    f2 = torch.randn(f.shape, generator=torch.Generator().manual_seed(5)).to(dev)
    r = lambda feats: rasterize_views(views, 256, 256, 0, m, c, o, features=feats)
    a, b, ab = r(f), r(f2), r(2.0 * f - 3.0 * f2)
    assert torch.equal(a[1], run.feat_out) and torch.equal(a[2], run.mask_out)        # deterministic
    assert torch.equal(a[2], b[2]) and torch.equal(a[4], b[4])                          # geometry only
    assert float((ab[1] - (2.0 * a[1] - 3.0 * b[1])).abs().max()) < 2e-4
    ofw = util.oracle_forward(bi, 1)
    err = np.abs(a[1][1].cpu().numpy() - ofw["feature"]).max(0)
    _exempt_fragile_pixels(err, ofw)
    assert err.max() <= 1e-4
    np.testing.assert_array_equal(run.point_list()[run.tile_start()[run.T]:], ofw["point_list"])


def test_full_size_permutation_invariance(hip_device, full_run):
    """Relabelling the Gaussians (a random permutation of the input rows) changes every index the binning stages
    carry — scatter order, sort keys, half-tile render lists — but not what is composited where: the images must come
    out the same, and the gradients are the permuted gradients.  Only Gaussians of EXACTLY equal depth on a common
    pixel may swap their blend order (ties break by index, as published) — a real, if small, change of the result —
    so equality is asserted bit for bit on every pixel outside the radius of such a pair."""
    from latentsplat_amd.rasterizer import rasterize_views
    bi, run = full_run
    dev = hip_device
    views = util.view_table(bi, dev)
    G = bi["means"].shape[1]
    perm = torch.randperm(G, generator=torch.Generator().manual_seed(11)).to(dev)
    t = {k: bi[k].to(dev) for k in ("means", "cov6", "opac", "features")}
    gdim = {k: (0 if k == "opac" else 1) for k in t}          # opacities are (G, 1), shared by the views; the rest (V, G, ...)
    tp = {k: v.index_select(gdim[k], perm).contiguous() for k, v in t.items()}
    leaf = lambda d: {k: v.clone().requires_grad_(True) for k, v in d.items()}
    a, b = leaf(t), leaf(tp)
    oa = rasterize_views(views, 256, 256, 0, a["means"], a["cov6"], a["opac"], features=a["features"])
    ob = rasterize_views(views, 256, 256, 0, b["means"], b["cov6"], b["opac"], features=b["features"])
    assert torch.equal(oa[1], run.feat_out)
    # pixels that may legitimately change: inside the radius of a visible Gaussian whose depth bits occur twice in the view
    q0, q1 = run.q()
    radii = run.radii.cpu().numpy()
    ys, xs = np.mgrid[0:256, 0:256]
    may = np.zeros((2, 256, 256), bool)
    for v in range(2):
        vis = np.nonzero(radii[v] > 0)[0]
        bits = np.ascontiguousarray(q1[v][vis, 2]).view(np.uint32)
        order = np.argsort(bits, kind="stable")
        sb, sg = bits[order], vis[order]
        for k in range(1, 4):                                                   # equal-depth groups are pairs, rarely more
            same = np.nonzero(sb[k:] == sb[:-k])[0]
            for i in same:
                g1, g2 = sg[i], sg[i + k]
                r = radii[v][g1] + radii[v][g2] + 2
                if abs(q0[v][g1, 0] - q0[v][g2, 0]) <= r and abs(q0[v][g1, 1] - q0[v][g2, 1]) <= r:      # footprints can overlap
                    for gi in (g1, g2):
                        x, y, rr = q0[v][gi, 0], q0[v][gi, 1], radii[v][gi] + 1
                        may[v] |= (np.abs(xs - x) <= rr) & (np.abs(ys - y) <= rr)
    assert may.mean() < 0.10                                                   # the exemption stays an exception
    for i in (1, 2, 3):      # features, mask, depth
        diff = (oa[i] - ob[i]).detach().abs().cpu().numpy()
        diff = diff.max(1) if diff.ndim == 4 else diff
        assert not (diff[~may] > 0).any(), i                                   # bit for bit wherever no tie can reach
        assert diff.max() <= 1e-3 * max(1.0, float(oa[i].abs().max())), i     # a swapped pair changes alpha_1 alpha_2 (c_1 - c_2) T
    assert torch.equal(oa[4][:, perm], ob[4])                                  # radii
    g = torch.randn(oa[1].shape, generator=torch.Generator().manual_seed(12)).to(dev)
    (oa[1] * g).sum().backward()
    (ob[1] * g).sum().backward()
    for k in t:
        want, got = a[k].grad.index_select(gdim[k], perm), b[k].grad
        scale = max(1.0, float(want.abs().max()))
        err = (got - want).abs().reshape(-1, got.shape[-1]).max(-1).values / scale
        # the order of the atomic sums differs (<= 2e-5 of scale); rows that share a pixel with a swapped pair change for real
        assert float((err > 2e-5).float().mean()) < 1e-3 and float(err.max()) <= 2e-3, (k, float(err.max()))


def test_full_size_backward_against_oracle(hip_device, full_run):
    """configs[2]: 300k Gaussians, forward+backward, gradient parity within 1e-4 (of the scale)."""
    from latentsplat_amd.rasterizer import rasterize_views
    bi, _ = full_run
    dev = hip_device
    views = util.view_table(bi, dev)
    req = lambda k: bi[k].to(dev).clone().requires_grad_(True)
    m, c, o, f = req("means"), req("cov6"), req("opac"), req("features")
    out = rasterize_views(views[:1], 256, 256, 0, m[:1], c[:1], o, features=f[:1])
    g = torch.randn(out[1].shape, generator=torch.Generator().manual_seed(11))
    grads = torch.autograd.grad((out[1] * g.to(dev)).sum(), (m, c, o, f))
    ofw = util.oracle_forward(bi, 0)
    b = util.oracle_backward(bi, 0, ofw, None, g[0].numpy())
    # A decision flip at a fragile evaluation changes alpha by 1/255 for ONE (pixel, Gaussian) and
    # therefore the transmittance of everything that contributes to that pixel: the Gaussian itself
    # is exempt, the other contributors of that pixel get the flip-sized bound, everything else
    # (>= 95 % of the rows, asserted and recorded by the helper) is held to 1e-4 of the scale.
    direct, behind = util.fragile_gaussians(ofw, 256)
    for name, got, want in (("means3D", grads[0][0], b["means3D"]), ("cov3D", grads[1][0], b["cov3D"]),
                            ("opacities", grads[2], b["opacities"]), ("features", grads[3][0], b["features"])):
        got = got.cpu().numpy()
        # clean rows (no fragile evaluation nearby): 2e-5 of the tensor's scale AND 1e-4 of max(1, |row|) — measured
        # 1.8e-6 / 4.4e-5 (tools/grad_budget.py, which also shows the float32 oracle itself 0.7e-6 / 1.9e-5 away from
        # a float64 evaluation).  dL/dcov3D rows mix entries of very different magnitude through the float32
        # conic -> covariance chain (the oracle runs that stage in double): per-row bar 2e-3 there (measured 7.7e-4).
        util.assert_grad_close_except_fragile(got, want, direct, behind, 1e-4, f"full-size dL/d{name}", clean_tol=2e-5,
                                              row_tol=2e-3 if name == "cov3D" else 1e-4)
        err = np.abs(got - want).reshape(got.shape[0], -1).max(1)
        assert np.median(err) <= 1e-6 * max(1.0, np.abs(want).max())


def test_device_camera_table_matches_host_math(hip_device):
    """lsr_build_views == the reference's per-call camera math (restated in cuda_splatting._scaled_cameras)."""
    from latentsplat_amd.decoder import cuda_splatting as cs
    from latentsplat_amd.rasterizer import build_view_table, make_view_table
    sc = util.make_scene(10, image_size=64, views=7, color_sh_degree=0)
    near = sc.near * torch.linspace(0.8, 1.7, 7)
    far = sc.far * torch.linspace(1.0, 0.5, 7)
    intr = sc.intrinsics.clone()
    intr[:, 0, 0] = torch.linspace(0.6, 1.4, 7); intr[:, 1, 1] = torch.linspace(1.3, 0.7, 7)
    intr[:, 0, 2] = torch.linspace(0.45, 0.55, 7)
    bg = torch.rand(7, 3)
    for si in (True, False):
        cams, scale = cs._scaled_cameras(sc.extrinsics, intr, near, far, si)   # float32, as the reference
        want = make_view_table(cams.view_matrix, cams.full_projection, cams.campos, cams.tan_fov_x, cams.tan_fov_y, bg, scale)
        got = build_view_table(sc.extrinsics.to(hip_device), intr.to(hip_device), near.to(hip_device), far.to(hip_device),
                               bg.to(hip_device), si).cpu()
        assert got.shape == (7, 44)
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    got1 = build_view_table(sc.extrinsics.to(hip_device), intr.to(hip_device), near.to(hip_device), far.to(hip_device),
                            bg[0].to(hip_device), True).cpu()
    assert torch.equal(got1[:, 37:40], bg[0][None].expand(7, 3))


def test_large_image_uses_global_histogram_paths(hip_device):
    """More than 4096 / 8192 tiles: the LDS-privatised tile histograms of k_preprocess / k_scatter
    fall back to global atomics; results must not change.
    The last shape has 257 tile columns: tile rectangles no longer fit the 8-byte bin records (byte
    coordinates) and the 16-byte form is used."""
    for H, W, G in ((1040, 1024, 1500), (1552, 1408, 1500), (48, 4112, 20000)):       # 65*64 = 4160 tiles, 97*88 = 8536 tiles, 3*257
        sc = util.make_scene(G, image_size=max(H, W), views=1, color_sh_degree=1, feature_channels=4,
                             sigma_px=(2.0, 40.0), opacity_scale=1.0)
        bi = util.boundary_inputs(sc, H, W, bg=(0.1, 0.2, 0.3))
        run = util.HipRun(bi, hip_device)
        o = util.oracle_forward(bi, 0)
        assert run.layout.geom_bin_stride == (16 if W > 4080 else 12)
        np.testing.assert_array_equal(run.rect()[0], o["rect"])
        np.testing.assert_array_equal(run.radii[0].cpu().numpy(), o["radii"])
        ts = run.tile_start()
        np.testing.assert_array_equal(np.diff(ts), o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0])
        np.testing.assert_array_equal(run.point_list()[:o["P"]], o["point_list"])
        util.assert_close_except_fragile(run.color_out[0].cpu().numpy(), o["color"], o, 1e-4, "colour")
        util.assert_close_except_fragile(run.feat_out[0].cpu().numpy(), o["feature"], o, 1e-4, "feature")


def test_single_gaussian_and_single_pixel_images(hip_device):
    from latentsplat_amd.rasterizer import rasterize_views
    dev = hip_device
    sc = util.make_scene(1, image_size=16, views=1, color_sh_degree=0, feature_channels=4, sigma_px=(2.0, 2.0), opacity_scale=3.0)
    sc.means[:] = torch.tensor([[0.0, 0.0, 2.0]])
    for H, W in ((16, 16), (1, 1), (3, 40)):
        bi = util.boundary_inputs(sc, H, W)
        run = util.HipRun(bi, dev)
        o = util.oracle_forward(bi, 0)
        np.testing.assert_array_equal(run.radii[0].cpu().numpy(), o["radii"])
        np.testing.assert_allclose(run.feat_out[0].cpu().numpy(), o["feature"], atol=1e-4)
        np.testing.assert_allclose(run.mask_out[0].cpu().numpy(), o["mask"], atol=1e-4)


def test_shape_mismatches_are_rejected_before_any_launch(hip_device):
    """A shared (stride-0) tensor whose G / trailing dims disagree with means3D must raise instead of
    letting the kernels read out of bounds."""
    from latentsplat_amd._lib import LsrError
    from latentsplat_amd.rasterizer import rasterize_views
    dev = hip_device
    sc = util.make_scene(500, image_size=32, views=2, color_sh_degree=1, feature_channels=4)
    bi = util.boundary_inputs(sc, 32, 32)
    views = util.view_table(bi, dev)
    m, c, o, sh, f = (bi[k].to(dev) for k in ("means", "cov6", "opac", "shs", "features"))
    ok = rasterize_views(views, 32, 32, 1, m, c, o, shs=sh, features=f)
    assert ok[0].shape == (2, 3, 32, 32)
    for kw in (dict(o=o[:-1]), dict(sh=sh[:-3]), dict(f=f[:, :-1]), dict(c=c[:, :-2]), dict(o=o[:, 0]),
               dict(sh=sh[:, :, :2]), dict(c=c[..., :5])):
        a = dict(m=m, c=c, o=o, sh=sh, f=f); a.update(kw)
        with pytest.raises(LsrError):
            rasterize_views(views, 32, 32, 1, a["m"], a["c"], a["o"], shs=a["sh"], features=a["f"])


def test_odd_gaussian_count_unaligned_sh_slices(hip_device):
    """G = 2001: per-view SH slices (G*K*3 floats apart) and a caller's shs[i] view are not 16-byte
    aligned, so the SH kernels must leave the 16-byte LDS-DMA staging path."""
    from latentsplat_amd.rasterizer import rasterize_views
    dev = hip_device
    sc = util.make_scene(2001, image_size=48, views=3, color_sh_degree=2, feature_channels=4, feature_sh_degree=1)
    bi = util.boundary_inputs(sc, 48, 48)
    views = util.view_table(bi, dev)
    V, G = 3, 2001
    shs_pv = bi["shs"][None].expand(V, -1, -1, -1).contiguous().to(dev).requires_grad_(True)     # (V,G,9,3): odd slice stride
    m, c, o, f = (bi[k].to(dev) for k in ("means", "cov6", "opac", "features"))
    color, feat, mask, depth, radii = rasterize_views(views, 48, 48, 2, m, c, o, shs=shs_pv, features=f)
    g = torch.randn(color.shape, generator=torch.Generator().manual_seed(3))
    (color * g.to(dev)).sum().backward()
    for v in range(V):
        ov = util.oracle_forward(bi, v)
        util.assert_close_except_fragile(color[v].detach().cpu().numpy(), ov["color"], ov, 1e-4, "colour (odd G)")
        b = util.oracle_backward(bi, v, ov, g[v].numpy(), None)
        want = b["shs"]
        assert np.abs(shs_pv.grad[v].cpu().numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    # drop-in per-view pattern with a slice that starts at an odd float offset of its storage
    import diff_gaussian_rasterization as dgr
    big = torch.zeros(G * 9 * 3 + 1, device=dev)
    big[1:] = bi["shs"].to(dev).reshape(-1)
    shs_off = big[1:].view(G, 9, 3)
    assert shs_off.data_ptr() % 16 != 0
    c0 = bi["cams"]
    s = dgr.GaussianRasterizationSettings(image_height=48, image_width=48, tanfovx=float(c0.tan_fov_x[0]), tanfovy=float(c0.tan_fov_y[0]),
                                          bg=bi["bg"][0].to(dev), scale_modifier=1.0, viewmatrix=c0.view_matrix[0].to(dev),
                                          projmatrix=c0.full_projection[0].to(dev), sh_degree=2, campos=c0.campos[0].to(dev),
                                          prefiltered=False, debug=False)
    img = dgr.GaussianRasterizer(s)(means3D=m[0], means2D=torch.zeros_like(m[0]), shs=shs_off, colors_precomp=None,
                                    features=None, opacities=o, cov3D_precomp=c[0])[0]
    o0 = util.oracle_forward(bi, 0)
    util.assert_close_except_fragile(img.cpu().numpy(), o0["color"], o0, 1e-4, "colour (unaligned shs view)")


def test_more_than_2_to_24_gaussians_run_without_footprint_culling(hip_device):
    """Scenes beyond 2^24 Gaussians (round 3 returned LSR_EUNSUPPORTED: the sort key packs `index << 8 | sub-block code`)
    keep the full 32-bit index in keys and list entries and give up the footprint culling instead — every entry of a
    tile's list goes to both half lists and is evaluated on every sub-block: the published algorithm's work.  16 780 216
    Gaussians, the visible ones at the END of the array (indices above 2^24), everything in front of them behind the
    camera: radii, pair count, the sorted canonical lists bit for bit, images and gradients against the oracle."""
    from latentsplat_amd.synthetic import Scene
    G_vis, pad, size = 6000, (1 << 24) - 3000, 64
    sc = util.make_scene(G_vis, image_size=size, views=1, color_sh_degree=None, feature_channels=4, seed=21)
    behind = torch.zeros(pad, 3); behind[:, 2] = -5.0                    # view z < 0.2: culled
    big = Scene(torch.cat([behind, sc.means]), torch.cat([torch.eye(3).expand(pad, 3, 3) * 1e-4, sc.covariances]),
                torch.cat([torch.full((pad,), 0.5), sc.opacities]), None,
                torch.cat([torch.zeros(pad, 4, 1), sc.feature_sh]), sc.extrinsics, sc.intrinsics, sc.near, sc.far)
    assert big.means.shape[0] > (1 << 24)
    bi = util.boundary_inputs(big, size, size)
    run = util.HipRun(bi, hip_device, shared_means=True)
    o = util.oracle_forward(bi, 0)
    np.testing.assert_array_equal(run.radii[0].cpu().numpy(), o["radii"])
    assert run.P == o["P"] and (o["radii"][pad:] > 0).sum() > 1000 and (o["radii"][:pad] > 0).sum() == 0
    ts, pl = run.tile_start(), run.point_list()
    np.testing.assert_array_equal(np.diff(ts), o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0])
    np.testing.assert_array_equal(pl[:o["P"]], o["point_list"])
    assert int(pl[:o["P"]].min()) >= pad                                  # every listed index needs more than 24 bits
    util.assert_close_except_fragile(run.feat_out[0].cpu().numpy(), o["feature"], o, 1e-4, "feature (> 2^24 Gaussians)")
    util.assert_close_except_fragile(run.mask_out[0].cpu().numpy(), o["mask"], o, 1e-4, "mask (> 2^24 Gaussians)")
    # and the backward through the autograd op
    from latentsplat_amd.rasterizer import rasterize_views
    dev = hip_device
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)
    means, cov6, opac, feats = leaf(bi["means"][0]), leaf(bi["cov6"][0]), leaf(bi["opac"]), leaf(bi["features"][0])
    color, feat, mask, depth, radii = rasterize_views(util.view_table(bi, dev), size, size, 0, means, cov6, opac, features=feats)
    gf = torch.randn(feat.shape, generator=torch.Generator().manual_seed(2))
    (feat * gf.to(dev)).sum().backward()
    b = util.oracle_backward(bi, 0, o, None, gf[0].numpy())
    direct, behind_px = util.fragile_gaussians(o, size)
    util.assert_grad_close_except_fragile(feats.grad.cpu().numpy(), b["features"], direct, behind_px, 1e-4, "dL/dfeatures (> 2^24 Gaussians)", clean_tol=2e-5)
    util.assert_grad_close_except_fragile(opac.grad.cpu().numpy(), b["opacities"][:, None], direct, behind_px, 1e-4, "dL/dopacities (> 2^24 Gaussians)", clean_tol=2e-5)
    assert float(means.grad[:pad].abs().max()) == 0.0
