"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): tile / sort indices bit-exact; rendered colour / latent / mask /
depth within 1e-4 abs; gradients within 1e-4 (relative to the gradient scale, see below).
"""
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-4
CLEAN_TOL = 2e-5   # gradient rows with no fragile evaluation nearby (tests/util.py assert_grad_close_except_fragile)


def _check_forward(bi, run, views=None):
    V = bi["V"]
    ts = run.tile_start()
    pl = run.point_list()
    T = run.T
    rect = run.rect()
    q0, q1 = run.q()
    ncontrib = run.n_considered()
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
            util.assert_close_except_fragile(run.color_out[v].cpu().numpy(), o["color"], o, ABS_TOL, "colour")
        if o["feature"] is not None:
            util.assert_close_except_fragile(run.feat_out[v].cpu().numpy(), o["feature"], o, ABS_TOL, "feature")
        util.assert_close_except_fragile(run.mask_out[v].cpu().numpy(), o["mask"], o, ABS_TOL, "mask")
        dscale = max(1.0, float(np.abs(o["depth"]).max()))
        # (a flipped alpha >= 1/255 decision moves the depth image by alpha * T * z of THAT Gaussian: the flip bound is in
        # units of the deepest visible Gaussian, not of the blended image — tests/fuzz_parity.py found the difference in a
        # low-opacity scene)
        zmax = float(o["gdepth"][vis].max(initial=1.0))
        util.assert_close_except_fragile(run.depth_out[v].cpu().numpy(), o["depth"], o, ABS_TOL * dscale, "depth (tol 1e-4 of the largest depth)", flip_bound=2e-2 * max(dscale, zmax), scale=dscale)
        # the per-pixel list prefix kept for the backward pass may differ only where exp rounding
        # flips a decision: every mismatching pixel must be one the oracle flagged as fragile
        mism = np.flatnonzero(ncontrib[v].reshape(-1) != o["n_considered"].astype(np.int64).reshape(-1))
        fragile_px = set(int(x) for x in o["fragile"][:, 0]) if len(o["fragile"]) else set()
        assert set(int(x) for x in mism) <= fragile_px, f"n_considered differs at {len(set(mism) - fragile_px)} non-fragile pixels"
        assert len(mism) < 2e-3 * ncontrib[v].size, f"n_considered mismatch fraction {len(mism) / ncontrib[v].size}"
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


@pytest.mark.parametrize("name", ["cfg1_rgb_deg0", "rgb_deg4_feat4_deg2_v3", "ragged_image", "big_splats"])
def test_forward_parity_contracted_projection(hip_device, name):
    """The OTHER arithmetic convention of the projection stage: products fused into the sums they feed, as a
    compiler with contraction on (nvcc's default -fmad=true, which the real fork was built with) may build the published
    source.  The kernel's switch (lsr_set_projection_contraction) against the oracle's (oracle_set_fma_contraction): radii,
    rectangles, depth bits, pixel means, conics, tile offsets and sorted lists bit for bit, images at the usual bar —
    so that the default can be flipped in one commit the day vectors of the real fork say which convention it follows
    (tools/contraction_census.py counts what flips between the two: profiles/r04_contraction_census.json)."""
    from latentsplat_amd import _lib
    from oracle import oracle as orc
    lib = _lib.load()
    sc, H, W = _scene(CASES[name])
    bi = util.boundary_inputs(sc, H, W, bg=(0.2, 0.4, 0.6))
    try:
        lib.lsr_set_projection_contraction(1)
        orc.set_fma_contraction(True)
        assert lib.lsr_get_projection_contraction() == 1
        run = util.HipRun(bi, hip_device)
        _check_forward(bi, run)
        o_fused = util.oracle_forward(bi, bi["V"] - 1)
    finally:
        lib.lsr_set_projection_contraction(0)
        orc.set_fma_contraction(False)
    # and the two conventions really are different arithmetic: some depth bits move (1-2 ulp) — except in view 0 of the
    # synthetic scenes, whose camera looks down the z axis (its products are exact)
    o_plain = util.oracle_forward(bi, bi["V"] - 1)
    vis = (o_plain["radii"] > 0) & (o_fused["radii"] > 0)
    if bi["V"] > 1:
        assert (o_plain["gdepth"][vis].view(np.uint32) != o_fused["gdepth"][vis].view(np.uint32)).any()


def test_pixel_aligned_means_tile_rectangles(hip_device):
    """Gaussians sitting on pixel centres (what the reference's encoder emits: one Gaussian per context-view ray)
    project to x = k + 0.99998-type floats, where the published tile-rectangle expression
    `(int)((p.x + max_radius + BLOCK_X - 1) / BLOCK_X)` depends on its left-to-right float evaluation order:
    ((p + r) + 16) - 1 rounds up across a tile boundary where (p + r) + 15 does not.  Found by the chained path
    test (tests/test_path_gpu.py) in round 3; rectangles, pair counts and lists must be bit-exact here too."""
    S, G = 128, 128 * 128
    sc = util.make_scene(G, image_size=S, views=2, color_sh_degree=0, feature_channels=4, seed=9, sigma_px=(0.3, 2.5))
    ys, xs = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    u, v = (xs.reshape(-1).float() + 0.5) / S, (ys.reshape(-1).float() + 0.5) / S
    z = sc.means[:, 2]
    sc.means[:, 0] = (u - 0.5) / 0.8 * z
    sc.means[:, 1] = (v - 0.5) / 0.8 * z
    bi = util.boundary_inputs(sc, S, S)
    run = util.HipRun(bi, hip_device)
    o = util.oracle_forward(bi, 0)
    frac = np.abs(o["xy"][o["radii"] > 0] % 1.0 - 0.5)          # distance from an integer pixel coordinate, 0.5 = on it
    assert (frac > 0.4999).mean() > 0.5                          # the scene really is pixel aligned in view 0
    _check_forward(bi, run)


def _box_masks(xy, co, lst, tx0, ty0):
    """float64 restatement of the kernels' footprint box (lsr_blend.h footprint_cells) -> 16-bit sub-block masks."""
    A, B, C, o = (co[lst, k].astype(np.float64) for k in range(4))
    x, y = xy[lst, 0].astype(np.float64), xy[lst, 1].astype(np.float64)
    det = A * C - B * B
    tau = np.log(np.maximum(255.0 * o, 1e-300)) * 1.0001 + 1e-4
    s2 = 2.0 * np.maximum(tau, 0.0) / det
    ex, ey = np.sqrt(s2 * C) * 1.001 + 0.05, np.sqrt(s2 * A) * 1.001 + 0.05
    m = np.zeros(len(lst), np.int64)
    for r in range(4):
        for c in range(4):
            hit = (x - ex <= tx0 + 4 * c + 3) & (x + ex >= tx0 + 4 * c) & (y - ey <= ty0 + 4 * r + 3) & (y + ey >= ty0 + 4 * r)
            m |= np.where(hit & (o >= 1.0 / 255.0), 1 << (4 * r + c), 0)
    return m


@pytest.mark.parametrize("name", ["feat4", "ragged_image", "big_splats"])
def test_half_tile_render_lists(hip_device, name):
    """The per-half render lists the compositing kernels walk (k_sort_tiles, ABI v6 layout) against the
    canonical, bit-exact tile lists: every half list is an order-preserving sub-list of its tile's
    canonical list; LOSSLESS — an entry that reaches alpha >= 1/255 on any pixel of a 4x4 sub-block (dense
    float64 evaluation) is in that half's list with the sub-block's bit set; and tight — the listed
    (entry, sub-block) pairs are those of the footprint box (+ at most 2 % from float rounding)."""
    sc, H, W = _scene(CASES[name])
    bi = util.boundary_inputs(sc, H, W)
    run = util.HipRun(bi, hip_device)
    ts, pl, hc, hl, T = run.tile_start(), run.point_list(), run.half_count(), run.half_list(), run.T
    gx = (W + 15) // 16
    listed = boxed = 0
    for v in range(bi["V"]):
        o = util.oracle_forward(bi, v)
        xy, co = o["xy"], o["conic_opacity"]
        for t in range(T):
            s0, s1 = ts[v * T + t], ts[v * T + t + 1]
            n = s1 - s0
            canon = pl[s0:s1]
            ty, tx = divmod(t, gx)
            got = np.zeros(n, np.int64)            # 16-bit sub-block masks reassembled from the two lists
            pos = {int(g): i for i, g in enumerate(canon)}
            for h in range(2):
                cnt = hc[v * T + t, h]
                assert 0 <= cnt <= n
                words = hl[2 * s0 + h * n: 2 * s0 + h * n + cnt]
                idx, bits = (words & 0x00FFFFFF).astype(np.int64), (words >> 24).astype(np.int64)
                where = np.array([pos[int(g)] for g in idx], np.int64)
                assert (np.diff(where) > 0).all(), "half list is not an ordered sub-list of the canonical list"
                assert (bits != 0).all(), "list entry without a reachable sub-block"
                got[where] |= bits << (8 * h)
            if n == 0:
                continue
            # dense truth: alpha >= 1/255 (and power <= 0) anywhere in the sub-block
            px = tx * 16 + np.arange(16, dtype=np.float64)[None, None, :]
            py = ty * 16 + np.arange(16, dtype=np.float64)[None, :, None]
            dx = xy[canon, 0].astype(np.float64)[:, None, None] - px
            dy = xy[canon, 1].astype(np.float64)[:, None, None] - py
            A, B, C, op = (co[canon, k].astype(np.float64)[:, None, None] for k in range(4))
            power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
            reach = (power <= 0) & (np.minimum(0.99, op * np.exp(np.minimum(power, 0))) >= 1.0 / 255.0)
            need = np.zeros(n, np.int64)
            for r in range(4):
                for c in range(4):
                    need |= np.where(reach[:, 4 * r:4 * r + 4, 4 * c:4 * c + 4].any((1, 2)), 1 << (4 * r + c), 0)
            assert ((need & ~got) == 0).all(), f"view {v} tile {t}: a reachable sub-block is missing from the render lists"
            box = _box_masks(xy, co, canon, tx * 16, ty * 16)
            listed += sum(bin(int(m)).count("1") for m in got)
            boxed += sum(bin(int(m)).count("1") for m in box)
    assert listed <= 1.02 * boxed + 16, f"render lists hold {listed} (entry, sub-block) pairs, the footprint boxes {boxed}"
    assert listed >= 0.98 * boxed - 16


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


def _grad_case(hip_device, case, with_aux, edit_scene=None):
    from latentsplat_amd.rasterizer import rasterize_views
    sc, H, W = _scene(case)
    if edit_scene is not None:
        edit_scene(sc)
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
    frag = []
    for v in range(V):
        o = util.oracle_forward(bi, v)
        frag.append(util.fragile_gaussians(o, W))
        n = lambda g: None if g is None else g[v].numpy()
        b = util.oracle_backward(bi, v, o, n(gs["color"]), n(gs["feat"]), n(gs["mask"]), n(gs["depth"]))
        exp["means"].append(b["means3D"]); exp["cov"].append(b["cov3D"]); exp["m2d"].append(b["means2D"])
        if b["features"] is not None:
            exp["feat"].append(b["features"])
        exp_opac += b["opacities"]
        if exp_shs is not None:
            exp_shs += b["shs"]

    all_direct = np.unique(np.concatenate([f[0] for f in frag]))
    all_behind = np.unique(np.concatenate([f[1] for f in frag]))

    def close_per_view(got, want, what):   # (V,G,...) tensors: exemptions of the view itself
        for v in range(V):
            util.assert_grad_close_except_fragile(got[v].detach().cpu().numpy(), want[v], frag[v][0], frag[v][1], ABS_TOL, f"{what}[view {v}]", clean_tol=CLEAN_TOL)

    def close_shared(got, want, what):     # (G,...) tensors summed over views: union of exemptions
        util.assert_grad_close_except_fragile(got.detach().cpu().numpy(), want, all_direct, all_behind, ABS_TOL, what, clean_tol=CLEAN_TOL)

    close_per_view(means.grad, np.stack(exp["means"]), "dL/dmeans3D")
    close_per_view(cov6.grad, np.stack(exp["cov"]), "dL/dcov3D")
    close_shared(opac.grad, exp_opac, "dL/dopacities")
    close_per_view(m2d.grad, np.stack(exp["m2d"]), "dL/dmeans2D")
    if feats is not None:
        close_per_view(feats.grad, np.stack(exp["feat"]), "dL/dfeatures")
    if shs is not None:
        close_shared(shs.grad, exp_shs, "dL/dshs")


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


# ------------------------------------------------------------------------------------------
# fused scene-level inputs: scene scale in the view table, 3x3 covariances, stored-layout colour SH,
# latent SH coefficients evaluated in-kernel (what the reference does in PyTorch per view)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("cfg", [
    dict(G=3000, size=64, views=3, color_sh_degree=4, feature_channels=4, feature_sh_degree=2),
    dict(G=2000, size=48, views=2, color_sh_degree=None, feature_channels=8, feature_sh_degree=1),
    dict(G=2000, size=48, views=2, color_sh_degree=2, feature_channels=4, feature_sh_degree=0),
    dict(G=1500, size=32, views=6, color_sh_degree=3, feature_channels=4, feature_sh_degree=2),   # > 1 view chunk in sh.hip
])
def test_fused_scene_inputs_match_oracle(hip_device, cfg, shared):
    from latentsplat_amd.decoder import cuda_splatting as cs
    from latentsplat_amd.rasterizer import rasterize_views
    cfg = dict(cfg)
    G, size = cfg.pop("G"), cfg.pop("size")
    sc = util.make_scene(G, image_size=size, **cfg)
    V = sc.extrinsics.shape[0]
    near = sc.near * torch.linspace(1.0, 1.3, V)          # per-view scene scales differ
    cams, scale = cs._scaled_cameras(sc.extrinsics, sc.intrinsics, near, sc.far, True)
    from latentsplat_amd.rasterizer import make_view_table
    bg = torch.tensor([[0.2, 0.1, 0.4]]).expand(V, 3)
    views_cpu = make_view_table(cams.view_matrix, cams.full_projection, cams.campos, cams.tan_fov_x, cams.tan_fov_y, bg, scale)
    rep = (lambda t: t) if shared else (lambda t: None if t is None else t[None].expand(V, *t.shape).contiguous())
    scene = dict(means=rep(sc.means), cov=rep(sc.covariances), opac=rep(sc.opacities[:, None]),
                 shs=rep(sc.color_sh), fsh=rep(sc.feature_sh))
    dev = hip_device
    gpu = {k: (None if t is None else t.to(dev).clone().requires_grad_(True)) for k, t in scene.items()}
    deg = 0 if sc.color_sh is None else int(round(sc.color_sh.shape[-1] ** 0.5)) - 1
    color, feat, mask, depth, radii = rasterize_views(views_cpu.to(dev), size, size, deg, gpu["means"], gpu["cov"], gpu["opac"],
                                                      shs=gpu["shs"], feature_sh=gpu["fsh"], shs_channel_major=True)
    gen = torch.Generator().manual_seed(3)
    g_color = None if color is None else torch.randn(color.shape, generator=gen)
    g_feat = torch.randn(feat.shape, generator=gen)
    loss = (feat * g_feat.to(dev)).sum() + (0 if color is None else (color * g_color.to(dev)).sum())
    loss.backward()

    cpu = {k: (None if t is None else t.clone().requires_grad_(True)) for k, t in scene.items()}
    n = lambda t: None if t is None else t.detach().contiguous().numpy()
    frag = []
    for v in range(V):
        m, c6, op, sh, cp, ft = util.to_boundary(views_cpu, v, cpu["means"], cpu["cov"], cpu["opac"], cpu["shs"], None, None,
                                                 cpu["fsh"], True)
        vw = views_cpu[v]
        view = util.orc.View(size, size, float(vw[35]), float(vw[36]), vw[37:40].numpy(), vw[0:16].numpy().reshape(4, 4),
                             vw[16:32].numpy().reshape(4, 4), vw[32:35].numpy(), deg)
        o = util.orc.forward(view, n(m), n(c6), n(op), n(sh), None, n(ft))
        np.testing.assert_array_equal(radii[v].cpu().numpy(), o["radii"])
        frag.append(util.fragile_gaussians(o, size))
        if color is not None:
            util.assert_close_except_fragile(color[v].detach().cpu().numpy(), o["color"], o, ABS_TOL, "colour")
        util.assert_close_except_fragile(feat[v].detach().cpu().numpy(), o["feature"], o, ABS_TOL, "feature")
        util.assert_close_except_fragile(mask[v].detach().cpu().numpy(), o["mask"], o, ABS_TOL, "mask")
        b = util.orc.backward(view, n(m), n(c6), n(op), n(sh), None, n(ft), o,
                              None if g_color is None else g_color[v].numpy(), g_feat[v].numpy())
        outs, grads = [m, c6, op, ft], [b["means3D"], b["cov3D"], b["opacities"], b["features"]]
        if sh is not None:
            outs.append(sh); grads.append(b["shs"])
        torch.autograd.backward(outs, [torch.from_numpy(np.ascontiguousarray(x)) for x in grads])
    all_direct = np.unique(np.concatenate([f[0] for f in frag]))
    all_behind = np.unique(np.concatenate([f[1] for f in frag]))
    for k in scene:
        if scene[k] is None:
            continue
        got, want = gpu[k].grad.cpu().numpy(), cpu[k].grad.numpy()
        if shared:
            util.assert_grad_close_except_fragile(got, want, all_direct, all_behind, ABS_TOL, f"dL/d{k}", clean_tol=CLEAN_TOL)
        else:
            for v in range(V):
                util.assert_grad_close_except_fragile(got[v], want[v], frag[v][0], frag[v][1], ABS_TOL, f"dL/d{k}[view {v}]", clean_tol=CLEAN_TOL)


@pytest.mark.parametrize("cfg", [
    dict(G=3001, size=64, b=1, v=3, color_sh_degree=4, feature_channels=4, feature_sh_degree=2),    # shared scene, odd G (unaligned rows)
    dict(G=2000, size=(40, 72), b=3, v=2, color_sh_degree=4, feature_channels=4, feature_sh_degree=2),   # view groups: the decoder's shape
    dict(G=2500, size=48, b=1, v=6, color_sh_degree=3, feature_channels=None),                       # colour only, > 4 views per workgroup
    dict(G=1800, size=48, b=2, v=1, color_sh_degree=None, feature_channels=8, feature_sh_degree=1),  # latent harmonics only, 8 channels
    dict(G=900, size=32, b=1, v=2, color_sh_degree=1, feature_channels=20, feature_sh_degree=1),     # 23 channels: 64-float records
    # the largest coefficient rows (degree-4 colour + 13 latent channels at degree 2: 50 KB of LDS) next to 8 x 1024 tile
    # counters: the launch must not ask for more dynamic LDS than it may without a function attribute
    dict(G=1100, size=512, b=1, v=8, color_sh_degree=4, feature_channels=13, feature_sh_degree=2),
])
def test_fused_projection_and_sh_kernel_equals_the_two_kernel_path(hip_device, cfg):
    """Calls whose payload is harmonics only and whose views share their inputs run the projection and the SH payload
    pass as ONE kernel (sh.hip k_preprocess_sh); everything else runs k_preprocess + k_sh_fwd.  Same arithmetic, so
    every output, the radii and every gradient must be bitwise identical between the two (LSR_FUSE_SH, switched in
    process), in the synchronous and the no-sync forward."""
    from latentsplat_amd import _lib
    from latentsplat_amd.decoder import cuda_splatting as cs
    from latentsplat_amd.rasterizer import make_view_table, rasterize_views
    cfg = dict(cfg)
    G, size, b, v = cfg.pop("G"), cfg.pop("size"), cfg.pop("b"), cfg.pop("v")
    H, W = (size, size) if isinstance(size, int) else size
    dev = hip_device
    scenes = [util.make_scene(G, image_size=max(H, W), views=v, seed=70 + s, **cfg) for s in range(b)]
    tables = []
    for sc in scenes:
        cams, scale = cs._scaled_cameras(sc.extrinsics, sc.intrinsics, sc.near * torch.linspace(1.0, 1.2, v), sc.far, True)
        tables.append(make_view_table(cams.view_matrix, cams.full_projection, cams.campos, cams.tan_fov_x,
                                      cams.tan_fov_y, torch.tensor([[0.3, 0.2, 0.1]]).expand(v, 3), scale))
    views = torch.cat(tables).to(dev)
    stack = lambda name: None if getattr(scenes[0], name) is None else torch.stack([getattr(s, name) for s in scenes])
    sq = (lambda t: None if t is None else t[0]) if b == 1 else (lambda t: t)          # b = 1: shared (G, ...) tensors
    fields = dict(means=sq(stack("means")), cov=sq(stack("covariances")), opac=sq(stack("opacities")[..., None]),
                  shs=sq(stack("color_sh")), fsh=sq(stack("feature_sh")))
    deg = 0 if fields["shs"] is None else int(round(fields["shs"].shape[-1] ** 0.5)) - 1
    gen = torch.Generator().manual_seed(5)
    results = {}
    try:
        for fuse in (1, 0):
            _lib.set_knob("LSR_FUSE_SH", fuse)
            for mode, kw in (("sync", {}), ("nosync", dict(pair_capacity=4 * b * v * G + 16, max_tile_hint=4096))):
                leaf = {k: (None if x is None else x.to(dev).clone().requires_grad_(True)) for k, x in fields.items()}
                out = rasterize_views(views, H, W, deg, leaf["means"], leaf["cov"], leaf["opac"], shs=leaf["shs"],
                                      feature_sh=leaf["fsh"], shs_channel_major=True, **kw)
                results[(fuse, mode)] = [None if o is None else o.detach().clone() for o in out]
    finally:
        _lib.set_knob("LSR_FUSE_SH", 1)
    ref = results[(0, "sync")]
    for key, res in results.items():
        for name, a, r in zip(("colour", "feature", "mask", "depth", "radii"), res, ref):
            assert (a is None) == (r is None), (key, name)
            if a is not None:
                assert torch.equal(a, r), f"{key}: {name} differs between the fused and the two-kernel path"


@pytest.mark.parametrize("cfg,direct", [
    (dict(color_sh_degree=4, feature_channels=4, feature_sh_degree=2), False),      # the reference's experiment payload
    (dict(color_sh_degree=None, feature_channels=4, feature_sh_degree=0), True),    # direct features only
    (dict(color_sh_degree=1, feature_channels=8, feature_sh_degree=0), False),
])
def test_view_groups_equal_per_scene_calls(hip_device, cfg, direct):
    """b scenes x v views in ONE call (inputs keep their leading scene dimension, lsr_dims
    views_per_group) == b separate shared-scene calls: images, radii and every gradient."""
    from latentsplat_amd.decoder import cuda_splatting as cs
    from latentsplat_amd.rasterizer import make_view_table, rasterize_views
    dev, b, v, size, G = hip_device, 3, 2, 64, 1500
    scenes = [util.make_scene(G, image_size=size, views=v, seed=50 + s, **cfg) for s in range(b)]
    tables = []
    for sc in scenes:
        cams, scale = cs._scaled_cameras(sc.extrinsics, sc.intrinsics, sc.near * torch.linspace(1.0, 1.2, v), sc.far, True)
        tables.append(make_view_table(cams.view_matrix, cams.full_projection, cams.campos, cams.tan_fov_x,
                                      cams.tan_fov_y, torch.tensor([[0.3, 0.2, 0.1]]).expand(v, 3), scale))
    views = torch.cat(tables).to(dev)
    stack = lambda name: None if getattr(scenes[0], name) is None else torch.stack([getattr(s, name) for s in scenes])
    fields = dict(means=stack("means"), cov=stack("covariances"), opac=stack("opacities")[..., None],
                  shs=stack("color_sh"), feats=stack("feature_sh")[..., 0].contiguous() if direct else stack("feature_sh"))
    deg = 0 if fields["shs"] is None else int(round(fields["shs"].shape[-1] ** 0.5)) - 1
    gen = torch.Generator().manual_seed(9)

    def run(t, vw):
        leaf = {k: (None if x is None else x.to(dev).clone().requires_grad_(True)) for k, x in t.items()}
        kw = dict(features=leaf["feats"]) if direct else dict(feature_sh=leaf["feats"])
        out = rasterize_views(vw, size, size, deg, leaf["means"], leaf["cov"], leaf["opac"], shs=leaf["shs"],
                              shs_channel_major=True, **kw)
        return out, leaf

    (color, feat, mask, depth, radii), leaf = run(fields, views)
    g_feat = torch.randn(feat.shape, generator=gen).to(dev)
    g_color = None if color is None else torch.randn(color.shape, generator=gen).to(dev)
    g_depth = torch.randn(depth.shape, generator=gen).to(dev)
    ((feat * g_feat).sum() + (depth * g_depth).sum() + (0 if color is None else (color * g_color).sum())).backward()
    for s in range(b):
        one = {k: (None if x is None else x[s]) for k, x in fields.items()}          # (G, ...): shared by the scene's views
        sl = slice(s * v, (s + 1) * v)
        (c1, f1, m1, d1, r1), leaf1 = run(one, views[sl])
        assert torch.equal(radii[sl], r1)
        assert torch.equal(feat[sl], f1) and torch.equal(mask[sl], m1) and torch.equal(depth[sl], d1)
        assert color is None or torch.equal(color[sl], c1)
        ((f1 * g_feat[sl]).sum() + (d1 * g_depth[sl]).sum() + (0 if c1 is None else (c1 * g_color[sl]).sum())).backward()
        for k in leaf:
            if leaf[k] is None:
                continue
            a, r = leaf[k].grad[s], leaf1[k].grad
            assert a.shape == r.shape
            assert float((a - r).abs().max()) <= 1e-5 * max(1e-6, float(r.abs().max())), k


def test_view_group_argument_errors(hip_device):
    from latentsplat_amd.rasterizer import LsrError, rasterize_views
    sc = util.make_scene(300, image_size=32, views=4, color_sh_degree=None, feature_channels=4, seed=1)
    bi = util.boundary_inputs(sc, 32, 32)
    vt = util.view_table(bi, hip_device)
    m, c, f = (bi[k][0].to(hip_device) for k in ("means", "cov6", "features"))
    o = bi["opac"].to(hip_device)                      # (G, 1): shared
    two = lambda t: torch.stack([t, t])
    with pytest.raises(LsrError):      # 3 slices do not divide 4 views
        rasterize_views(vt, 32, 32, 0, torch.stack([m, m, m]), c, o, features=f)
    with pytest.raises(LsrError):      # per-scene means but shared covariances
        rasterize_views(vt, 32, 32, 0, two(m), c, o, features=f)
    out = rasterize_views(vt, 32, 32, 0, two(m), two(c), two(o), features=two(f))
    assert out[1].shape == (4, 4, 32, 32)


def test_color_sh_reference_axis_convention(hip_device):
    """lsr_dims.color_sh_convention = LSR_SH_AXES_REFERENCE (rasterizer.set_color_sh_convention):
    colours and the gradients w.r.t. SH coefficients and means equal the oracle in the same
    convention, degree 4; and the default convention is restored / still differs."""
    from latentsplat_amd import rasterizer as rz
    from oracle import oracle as orc
    dev = hip_device
    sc = util.make_scene(2000, image_size=64, views=2, color_sh_degree=4, feature_channels=4, feature_sh_degree=0)
    bi = util.boundary_inputs(sc, 64, 64, bg=(0.2, 0.1, 0.3))
    views = util.view_table(bi, dev)
    gen = torch.Generator().manual_seed(13)
    g_color = torch.randn((2, 3, 64, 64), generator=gen)

    def run():
        req = lambda t: t.to(dev).clone().requires_grad_(True)
        means, cov6, opac, shs, feats = map(req, (bi["means"], bi["cov6"], bi["opac"], bi["shs"], bi["features"]))
        color, feat, mask, depth, radii = rz.rasterize_views(views, 64, 64, 4, means, cov6, opac, shs=shs, features=feats)
        (color * g_color.to(dev)).sum().backward()
        return color.detach().cpu().numpy(), means.grad.cpu().numpy(), shs.grad.cpu().numpy()

    default_color = run()[0]
    try:
        rz.set_color_sh_convention("reference")
        orc.set_sh_convention("reference")
        assert rz.get_color_sh_convention() == "reference"
        color, g_means, g_shs = run()
        want_shs = np.zeros(tuple(bi["shs"].shape), np.float64)
        for v in range(2):
            o = util.oracle_forward(bi, v)
            util.assert_close_except_fragile(color[v], o["color"], o, ABS_TOL, "colour (reference SH axes)")
            b = util.oracle_backward(bi, v, o, g_color[v].numpy(), None)
            direct, behind = util.fragile_gaussians(o, 64)
            util.assert_grad_close_except_fragile(g_means[v], b["means3D"], direct, behind, ABS_TOL, "dL/dmeans3D (reference SH axes)", clean_tol=CLEAN_TOL)
            want_shs += b["shs"]
        assert np.abs(g_shs - want_shs).max() <= ABS_TOL * max(1.0, np.abs(want_shs).max())
    finally:
        rz.set_color_sh_convention("3dgs")
        orc.set_sh_convention("3dgs")
    assert np.abs(color - default_color).max() > 1e-3
    assert np.array_equal(run()[0], default_color)


def test_single_pass_binning_equals_two_phase_and_falls_back_on_overflow(hip_device):
    """Round 5: k_preprocess writes the sort keys into fixed-capacity per-tile segments itself (no k_scatter).  The
    canonical lists, the half-tile render lists and the images must be bit for bit those of the two-phase path
    (LSR_SEGMENTS=0); tiles whose lists outgrow their segments (forced here with LSR_SEG_CAP = 64: most tiles of these
    scenes, but not all) are binned a second time by the overfull-tiles-only scatter — same results, in the synchronous
    forward (launched because the host knows the longest list) and in the no-sync forward (launched always)."""
    from latentsplat_amd import _lib
    from latentsplat_amd.rasterizer import last_forward_status, rasterize_views
    for case in (dict(G=9_000, size=(80, 112), views=3, color_sh_degree=1, feature_channels=4),
                 dict(G=2_500, size=64, views=5, color_sh_degree=None, feature_channels=8, sigma_px=(3.0, 20.0), opacity_scale=1.0)):
        case = dict(case)
        size = case.pop("size")
        H, W = size if isinstance(size, tuple) else (size, size)
        sc = util.make_scene(case.pop("G"), image_size=max(H, W), **case)
        bi = util.boundary_inputs(sc, H, W, bg=(0.1, 0.3, 0.5))
        runs = {}
        try:
            small = 64
            for name, seg in (("two_phase", 0), ("segments", 1), ("overflow_fallback", 1)):
                _lib.set_knob("LSR_SEGMENTS", seg)
                if name == "overflow_fallback":      # a capacity at the median list length: half of the tiles overfull
                    lens = np.diff(runs["two_phase"]["ts"])
                    small = int(-(-int(np.median(lens[lens > 0])) // 64) * 64)
                _lib.set_knob("LSR_SEG_CAP", small if name == "overflow_fallback" else 0)
                r = util.HipRun(bi, hip_device)
                runs[name] = dict(P=r.P, maxtile=r.maxtile, ts=r.tile_start(), pl=r.point_list(), hc=r.half_count(), hl=r.half_list(),
                                  img=[None if t is None else t.clone() for t in (r.color_out, r.feat_out, r.mask_out, r.depth_out)],
                                  nc=r.n_contrib(), radii=r.radii.clone())
            assert runs["two_phase"]["maxtile"] > small, "the forced capacity must be exceeded for the fallback leg to mean anything"
            a = runs["two_phase"]
            for name in ("segments", "overflow_fallback"):
                b = runs[name]
                assert (a["P"], a["maxtile"]) == (b["P"], b["maxtile"])
                np.testing.assert_array_equal(a["ts"], b["ts"], err_msg=name + ": tile offsets")
                np.testing.assert_array_equal(a["pl"], b["pl"], err_msg=name + ": canonical lists")
                np.testing.assert_array_equal(a["hc"], b["hc"], err_msg=name + ": half-list lengths")
                # (the half-list area is only defined inside the counted prefixes)
                T = a["hc"].shape[0]
                for vt in range(T):
                    s0, n = a["ts"][vt], a["ts"][vt + 1] - a["ts"][vt]
                    for h in range(2):
                        c = a["hc"][vt, h]
                        np.testing.assert_array_equal(a["hl"][2 * s0 + h * n: 2 * s0 + h * n + c], b["hl"][2 * s0 + h * n: 2 * s0 + h * n + c])
                np.testing.assert_array_equal(a["nc"], b["nc"], err_msg=name + ": n_contrib")
                assert torch.equal(a["radii"], b["radii"])
                for x, y in zip(a["img"], b["img"]):
                    assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), name + ": images"
            lens = np.diff(a["ts"])
            assert (lens > small).any() and ((lens > 0) & (lens <= small)).any(), "the fallback leg needs overfull AND fitting tiles"
            _lib.set_knob("LSR_SEGMENTS", 1)
            from latentsplat_amd import rasterizer as rz
            rz.set_reached_only(False)        # (the autograd op bins the reachable pairs only by default; the counts compared below are the published ones)
            views = util.view_table(bi, hip_device)
            t = lambda x: x.to(hip_device).contiguous()
            kw = dict(features=t(bi["features"]), pair_capacity=2 * a["P"] + 64, max_tile_hint=int(a["maxtile"]))
            if bi["shs"] is not None:
                kw["shs"] = t(bi["shs"])
            for cap in (0, small):
                _lib.set_knob("LSR_SEG_CAP", cap)
                with torch.no_grad():
                    out = rasterize_views(views, H, W, bi["sh_degree"], t(bi["means"]), t(bi["cov6"]), t(bi["opac"]), **kw)
                st = last_forward_status()
                assert not st["overflow"] and st["num_pairs"] == a["P"] and st["max_tile_pairs"] == a["maxtile"], (cap, st)
                assert torch.equal(out[1], a["img"][1]) and torch.equal(out[2], a["img"][2]) and torch.equal(out[3], a["img"][3])
        finally:
            from latentsplat_amd import rasterizer as rz
            rz.set_reached_only(True)
            _lib.set_knob("LSR_SEGMENTS", 1)
            _lib.set_knob("LSR_SEG_CAP", 0)


@pytest.mark.parametrize("case", [
    dict(G=30_000, size=96, views=4, color_sh_degree=None, feature_channels=4),                      # 4 payload channels
    dict(G=14_000, size=(80, 112), views=3, color_sh_degree=2, feature_channels=4),                  # 7 channels, ragged image
    dict(G=4_000, size=64, views=4, color_sh_degree=None, feature_channels=4, sigma_px=(3.0, 25.0), opacity_scale=1.0),   # long lists, pixels that run out of transmittance
])
@pytest.mark.parametrize("rows", [0, 1])
def test_forward_for_backward_narrows_the_render_lists_losslessly(hip_device, case, rows):
    """Round 5: a forward that a backward will follow (lsr_dims.forward_flags, set by the autograd op) records on which
    sub-blocks every list entry contributed and narrows the entry's sub-block bits in the half-tile render list to those.
    The images are untouched, the narrowed bits are a subset of the footprint-box bits, and the gradients the backward
    computes from the narrowed lists are those from the original lists (up to the order of float sums).  rows = 1 (round 6):
    the row items of small view batches narrow their own nibble of every entry's bits."""
    from latentsplat_amd import _lib
    from latentsplat_amd.rasterizer import rasterize_views
    case = dict(case)
    size = case.pop("size")
    H, W = size if isinstance(size, tuple) else (size, size)
    sc = util.make_scene(case.pop("G"), image_size=max(H, W), **case)
    bi = util.boundary_inputs(sc, H, W, bg=(0.3, 0.2, 0.1))
    dev = hip_device
    try:
        _lib.set_knob("LSR_FWD_ROWS", rows)       # the half-tile kernel / the row items whatever the batch size
        _lib.set_knob("LSR_FWD_QUAD", 0)
        plain = util.HipRun(bi, dev)
        rec = util.HipRun(bi, dev, forward_flags=_lib.FWD_FOR_BACKWARD)
        for a, b in ((plain.color_out, rec.color_out), (plain.feat_out, rec.feat_out), (plain.mask_out, rec.mask_out), (plain.depth_out, rec.depth_out)):
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
        np.testing.assert_array_equal(plain.n_contrib(), rec.n_contrib())
        np.testing.assert_array_equal(plain.point_list(), rec.point_list())
        ts, hc, hp, hr = plain.tile_start(), plain.half_count(), plain.half_list(), rec.half_list()
        np.testing.assert_array_equal(hc, rec.half_count())
        before = after = 0
        for vt in range(hc.shape[0]):
            s0, n = ts[vt], ts[vt + 1] - ts[vt]
            for h in range(2):
                a = hp[2 * s0 + h * n: 2 * s0 + h * n + hc[vt, h]]
                b = hr[2 * s0 + h * n: 2 * s0 + h * n + hc[vt, h]]
                np.testing.assert_array_equal(a & 0x00FFFFFF, b & 0x00FFFFFF)            # the same entries in the same order
                assert not ((b >> 24) & ~(a >> 24)).any(), "narrowed bits must be a subset of the footprint-box bits"
                before += int(sum(bin(int(x)).count("1") for x in (a >> 24)))
                after += int(sum(bin(int(x)).count("1") for x in (b >> 24)))
        assert after < 0.97 * before, (before, after)
        if rows:   # a row stops when ITS 64 pixels are finished and clears what lies behind: never less narrowing than the half-tile kernel's
            _lib.set_knob("LSR_FWD_ROWS", 0)
            hh = util.HipRun(bi, dev, forward_flags=_lib.FWD_FOR_BACKWARD).half_list()
            _lib.set_knob("LSR_FWD_ROWS", 1)
            assert not ((hr >> 24) & ~(hh >> 24)).any(), "row-item narrowing must be at least the half-tile kernel's"
        # gradients through the autograd op, with and without the narrowing
        views = util.view_table(bi, dev)
        t = {k: (None if bi[k] is None else bi[k].to(dev)) for k in ("means", "cov6", "opac", "shs", "features")}
        gen = torch.Generator().manual_seed(13)
        grads = {}
        for record in (1, 0):
            _lib.set_knob("LSR_FWD_RECORD", record)
            leaves = {k: (None if v is None else v.clone().requires_grad_(True)) for k, v in t.items()}
            out = rasterize_views(views, H, W, bi["sh_degree"], leaves["means"], leaves["cov6"], leaves["opac"], shs=leaves["shs"], features=leaves["features"])
            outs = [o for o in out[:2] if o is not None]
            gen.manual_seed(13)
            torch.autograd.backward(outs, [torch.randn(o.shape, generator=gen).to(dev) for o in outs])
            grads[record] = {k: v.grad.clone() for k, v in leaves.items() if v is not None}
            assert torch.equal(out[2], plain.mask_out) and torch.equal(out[3], plain.depth_out)
        for k in grads[1]:
            scale = max(1.0, float(grads[0][k].abs().max()))
            assert float((grads[1][k] - grads[0][k]).abs().max()) <= 2e-5 * scale, k
    finally:
        _lib.set_knob("LSR_FWD_ROWS", -1)
        _lib.set_knob("LSR_FWD_QUAD", -1)
        _lib.set_knob("LSR_FWD_RECORD", 1)
