"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): tile / sort indices bit-exact; rendered colour / latent / mask /
depth within 1e-4 abs; gradients within 1e-4 (relative to the gradient scale, see below).
"""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-4


def _check_forward(bi, run, views=None):
    V = bi["V"]
    ts = run.tile_start()
    pl = run.point_list()
    T = run.T
    rect = run.rect()
    q0, q1 = run.q()
    ncontrib = run.n_contrib()
    total_P = 0
    for v in (range(V) if views is None else views):
        o = util.oracle_forward(bi, v)
        total_P += o["P"]
        # --- integer / index work: bit exact ---
        np.testing.assert_array_equal(run.radii[v].cpu().numpy(), o["radii"], err_msg="radii")
        np.testing.assert_array_equal(rect[v], o["rect"], err_msg="tile rectangles")
        vis = o["radii"] > 0
        np.testing.assert_array_equal(q1[v][vis, 2].view(np.uint32), o["gdepth"][vis].view(np.uint32), err_msg="depth bits")
        np.testing.assert_array_equal(q0[v][vis, :2].view(np.uint32), o["xy"][vis].view(np.uint32), err_msg="pixel means")
        np.testing.assert_array_equal(q0[v][vis, 2:].view(np.uint32), o["conic_opacity"][vis, :2].view(np.uint32), err_msg="conic")
        base = ts[v * T]
        starts = ts[v * T:(v + 1) * T] - base
        ends = ts[v * T + 1:(v + 1) * T + 1] - base
        nonempty = o["ranges"][:, 1] > o["ranges"][:, 0]
        np.testing.assert_array_equal((ends - starts), (o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]), err_msg="tile counts")
        np.testing.assert_array_equal(starts[nonempty], o["ranges"][nonempty, 0], err_msg="tile starts")
        np.testing.assert_array_equal(pl[base:base + o["P"]], o["point_list"], err_msg="sorted tile lists")
        # --- float outputs: 1e-4 abs ---
        if o["color"] is not None:
            np.testing.assert_allclose(run.color_out[v].cpu().numpy(), o["color"], atol=ABS_TOL, rtol=0)
        if o["feature"] is not None:
            np.testing.assert_allclose(run.feat_out[v].cpu().numpy(), o["feature"], atol=ABS_TOL, rtol=0)
        np.testing.assert_allclose(run.mask_out[v].cpu().numpy(), o["mask"], atol=ABS_TOL, rtol=0)
        dscale = max(1.0, float(np.abs(o["depth"]).max()))
        np.testing.assert_allclose(run.depth_out[v].cpu().numpy(), o["depth"], atol=ABS_TOL * dscale, rtol=0)
        # the per-pixel list prefix kept for the backward pass may differ only where exp rounding
        # flips the transmittance-termination decision
        mism = (ncontrib[v] != o["n_considered"].astype(np.int32)).mean()
        assert mism < 2e-3, f"n_considered mismatch fraction {mism}"
    if views is None:
        assert run.P == total_P


CASES = {
    # BASELINE config #1: 10k Gaussians, 64x64, RGB SH degree 0
    "cfg1_rgb_deg0": dict(G=10_000, size=64, views=1, color_sh_degree=0, feature_channels=None),
    "feat4": dict(G=5_000, size=64, views=1, color_sh_degree=None, feature_channels=4),
    "rgb_deg4_feat4_deg2_v3": dict(G=6_000, size=96, views=3, color_sh_degree=4, feature_channels=4, feature_sh_degree=2),
    "feat8": dict(G=3_000, size=48, views=2, color_sh_degree=None, feature_channels=8),
    "rgb_deg3_feat8": dict(G=3_000, size=64, views=1, color_sh_degree=3, feature_channels=8),
    "feat32": dict(G=2_000, size=40, views=1, color_sh_degree=1, feature_channels=32),
    "ragged_image": dict(G=4_000, size=(50, 70), views=2, color_sh_degree=2, feature_channels=4),
    "big_splats": dict(G=1_500, size=64, views=1, color_sh_degree=1, feature_channels=4, sigma_px=(4.0, 30.0), opacity_scale=1.0),
}


def _scene(case):
    case = dict(case)
    size = case.pop("size")
    H, W = (size, size) if isinstance(size, int) else size
    G = case.pop("G")
    sc = util.make_scene(G, image_size=max(H, W), **case)
    return sc, H, W


@pytest.mark.parametrize("name", list(CASES))
def test_forward_parity(hip_device, name):
    sc, H, W = _scene(CASES[name])
    bi = util.boundary_inputs(sc, H, W, bg=(0.2, 0.4, 0.6))
    run = util.HipRun(bi, hip_device)
    _check_forward(bi, run)


@pytest.mark.parametrize("pxl", ["1", "2", "4"])
def test_forward_parity_all_wave_shapes(hip_device, monkeypatch, pxl):
    monkeypatch.setenv("LSR_PXL", pxl)
    sc, H, W = _scene(CASES["rgb_deg4_feat4_deg2_v3"])
    bi = util.boundary_inputs(sc, H, W, bg=(0.2, 0.4, 0.6))
    run = util.HipRun(bi, hip_device)
    _check_forward(bi, run)


def test_forward_shared_scene_equals_per_view(hip_device):
    """Stride-0 (scene shared by all views) must give exactly what per-view copies give."""
    sc = util.make_scene(4000, image_size=64, views=1, color_sh_degree=2, feature_channels=4)
    sc.extrinsics = sc.extrinsics.repeat(3, 1, 1); sc.intrinsics = sc.intrinsics.repeat(3, 1, 1)
    sc.near = sc.near.repeat(3); sc.far = sc.far.repeat(3)
    sc.extrinsics[1, 0, 3] = 0.1; sc.extrinsics[2, 1, 3] = -0.1
    bi = util.boundary_inputs(sc, 64, 64)
    a = util.HipRun(bi, hip_device, shared_means=False)
    b = util.HipRun(bi, hip_device, shared_means=True)
    assert torch.equal(a.color_out, b.color_out) and torch.equal(a.feat_out, b.feat_out)
    assert torch.equal(a.radii, b.radii)
    _check_forward(bi, b)


def _grad_case(hip_device, case, with_aux, pxl_env=None, monkeypatch=None):
    from latentsplat_amd.rasterizer import rasterize_views
    if pxl_env is not None:
        monkeypatch.setenv("LSR_PXL_BWD", pxl_env)
        monkeypatch.setenv("LSR_PXL", pxl_env)
    sc, H, W = _scene(case)
    bi = util.boundary_inputs(sc, H, W, bg=(0.3, 0.1, 0.5))
    V = bi["V"]
    dev = hip_device
    req = lambda t: None if t is None else t.to(dev).clone().requires_grad_(True)
    means, cov6, opac = req(bi["means"]), req(bi["cov6"]), req(bi["opac"])
    shs, feats = req(bi["shs"]), req(bi["features"])
    m2d = torch.zeros((V, means.shape[1], 3), device=dev, requires_grad=True)
    views = util.view_table(bi, dev)
    color, feat, mask, depth, radii = rasterize_views(views, H, W, bi["sh_degree"], means, cov6, opac,
                                                      shs=shs, features=feats, means2D=m2d)
    gen = torch.Generator().manual_seed(7)
    loss = 0
    gs = {}
    for name, out in (("color", color), ("feat", feat), ("mask", mask), ("depth", depth)):
        if out is None or (name in ("mask", "depth") and not with_aux):
            gs[name] = None
            continue
        gs[name] = torch.randn(out.shape, generator=gen)
        loss = loss + (out * gs[name].to(dev)).sum()
    loss.backward()
    # oracle, view by view; shared inputs (opac, shs) sum over views
    exp = dict(means=[], cov=[], feat=[], m2d=[])
    exp_opac = np.zeros((means.shape[1], 1), np.float64)
    exp_shs = None if shs is None else np.zeros(tuple(shs.shape), np.float64)
    for v in range(V):
        o = util.oracle_forward(bi, v)
        n = lambda g: None if g is None else g[v].numpy()
        b = util.oracle_backward(bi, v, o, n(gs["color"]), n(gs["feat"]), n(gs["mask"]), n(gs["depth"]))
        exp["means"].append(b["means3D"]); exp["cov"].append(b["cov3D"]); exp["m2d"].append(b["means2D"])
        if b["features"] is not None:
            exp["feat"].append(b["features"])
        exp_opac += b["opacities"]
        if exp_shs is not None:
            exp_shs += b["shs"]

    def close(got, want, what):
        got, want = got.detach().cpu().numpy().astype(np.float64), np.asarray(want, np.float64)
        scale = max(1.0, np.abs(want).max())  # 1e-4 of the gradient's own scale
        err = np.abs(got - want).max()
        assert err <= ABS_TOL * scale, f"{what}: max err {err:.3e} (scale {scale:.3e})"

    close(means.grad, np.stack(exp["means"]), "dL/dmeans3D")
    close(cov6.grad, np.stack(exp["cov"]), "dL/dcov3D")
    close(opac.grad, exp_opac, "dL/dopacities")
    close(m2d.grad, np.stack(exp["m2d"]), "dL/dmeans2D")
    if feats is not None:
        close(feats.grad, np.stack(exp["feat"]), "dL/dfeatures")
    if shs is not None:
        close(shs.grad, exp_shs, "dL/dshs")


GRAD_CASES = {
    "feat4": dict(G=3_000, size=64, views=1, color_sh_degree=None, feature_channels=4),
    "rgb_deg4_feat4_v2": dict(G=3_000, size=64, views=2, color_sh_degree=4, feature_channels=4, feature_sh_degree=2),
    "rgb_deg0": dict(G=3_000, size=48, views=1, color_sh_degree=0, feature_channels=None),
    "feat8_rgb_deg2": dict(G=2_000, size=48, views=1, color_sh_degree=2, feature_channels=8),
    "ragged_feat4": dict(G=2_500, size=(40, 56), views=2, color_sh_degree=1, feature_channels=4),
}


@pytest.mark.parametrize("name", list(GRAD_CASES))
@pytest.mark.parametrize("with_aux", [False, True])
def test_backward_parity(hip_device, name, with_aux):
    _grad_case(hip_device, GRAD_CASES[name], with_aux)


@pytest.mark.parametrize("pxl", ["1", "2", "4"])
def test_backward_parity_all_wave_shapes(hip_device, monkeypatch, pxl):
    _grad_case(hip_device, GRAD_CASES["feat4"], False, pxl_env=pxl, monkeypatch=monkeypatch)
