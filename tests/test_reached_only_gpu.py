"""ABI v9 LSR_FWD_REACHED_ONLY: the binning keeps only the (Gaussian, tile) pairs whose alpha >= 1/255 footprint box reaches the
tile.  The published algorithm pairs a Gaussian with every tile of the 3-sigma square around it; the pairs that cannot touch a
pixel were never in the half-tile render lists the compositing kernels walk, so dropping them before they are counted, keyed
and sorted changes nothing a consumer of the path sees: images, final_T, n_contrib, radii and the render lists bit for bit,
gradients up to the order of the float atomics.  What does change — pair count, tile offsets, the canonical sorted list — is
checked against its definition: the published list with the unreachable pairs removed, order kept."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

CASES = {
    "cloud_4ch_v3": dict(G=20_000, size=(96, 128), views=3, color_sh_degree=None, feature_channels=4),
    "rgb_feat4_big_splats": dict(G=4_000, size=64, views=2, color_sh_degree=2, feature_channels=4, sigma_px=(2.0, 20.0), opacity_scale=1.0),
    "tiny_splats_low_opacity": dict(G=6_000, size=80, views=1, color_sh_degree=None, feature_channels=8, sigma_px=(0.05, 0.5), opacity_scale=0.05),
}


def _lists(run):
    ts, pl, hc, hl = run.tile_start(), run.point_list(), run.half_count(), run.half_list()
    tiles = []
    for vt in range(hc.shape[0]):
        s0, n = ts[vt], ts[vt + 1] - ts[vt]
        halves = [hl[2 * s0 + h * n: 2 * s0 + h * n + hc[vt, h]].copy() for h in range(2)]
        tiles.append((pl[s0:s0 + n].copy(), halves))
    return tiles


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("segments", [1, 0])
def test_reached_only_binning_equals_the_published_lists_minus_the_unreachable_pairs(hip_device, name, segments):
    from latentsplat_amd import _lib
    case = dict(CASES[name])
    size = case.pop("size")
    H, W = size if isinstance(size, tuple) else (size, size)
    sc = util.make_scene(case.pop("G"), image_size=max(H, W), **case)
    bi = util.boundary_inputs(sc, H, W, bg=(0.2, 0.1, 0.3))
    try:
        _lib.set_knob("LSR_SEGMENTS", segments)       # single-pass binning (key emission in the projection kernel) / k_scatter
        full = util.HipRun(bi, hip_device)
        red = util.HipRun(bi, hip_device, forward_flags=_lib.FWD_REACHED_ONLY)
    finally:
        _lib.set_knob("LSR_SEGMENTS", 1)
    assert 0 < red.P < full.P
    compare_runs(full, red)


def compare_runs(full, red):
    """`red` (LSR_FWD_REACHED_ONLY) against `full` (published lists), both tests.util.HipRun of the same inputs."""
    for a, b in ((full.color_out, red.color_out), (full.feat_out, red.feat_out), (full.mask_out, red.mask_out), (full.depth_out, red.depth_out)):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), "reached-only: images differ"
    assert torch.equal(full.radii, red.radii), "reached-only: radii differ"
    np.testing.assert_array_equal(full.n_contrib(), red.n_contrib(), err_msg="reached-only: n_contrib")
    np.testing.assert_array_equal(full.final_T(), red.final_T(), err_msg="reached-only: final_T")
    np.testing.assert_array_equal(full.half_count(), red.half_count(), err_msg="reached-only: half-list lengths")
    assert red.P <= full.P
    lf, lr = _lists(full), _lists(red)
    dropped = 0
    for (canon, hf), (kept, hr) in zip(lf, lr):
        for h in range(2):
            np.testing.assert_array_equal(hf[h], hr[h], err_msg="reached-only: sorted tile lists (half-tile render lists)")   # bit for bit
        reach = np.zeros(0, np.int64) if len(canon) == 0 else np.unique(np.concatenate([x & 0x00FFFFFF for x in hf]).astype(np.int64))
        want = canon[np.isin(canon, reach)]                                   # the published list minus the pairs no half list holds
        np.testing.assert_array_equal(kept, want, err_msg="reached-only: sorted tile lists (canonical minus unreachable)")
        dropped += len(canon) - len(kept)
    assert dropped == full.P - red.P


def test_reached_only_gradients_and_the_autograd_default(hip_device):
    """The autograd op sets the bit by default (`set_reached_only(False)` / LSR_REACHED_ONLY=0: the published lists): same
    images bit for bit, gradients up to the order of the float atomics, fewer pairs."""
    from latentsplat_amd import rasterizer as rz
    dev = hip_device
    sc = util.make_scene(30_000, image_size=128, views=8, color_sh_degree=None, feature_channels=4)
    bi = util.boundary_inputs(sc, 128, 128)
    vt = util.view_table(bi, dev)
    t = {k: bi[k].to(dev) for k in ("means", "cov6", "opac", "features")}
    g = torch.randn((8, 4, 128, 128), generator=torch.Generator().manual_seed(4)).to(dev)
    res = {}
    assert rz._REACHED_ONLY
    try:
        for on in (True, False):
            rz.set_reached_only(on)
            leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
            out = rz.rasterize_views(vt, 128, 128, 0, leaves["means"], leaves["cov6"], leaves["opac"], features=leaves["features"])
            out[1].backward(g)
            res[on] = (out, {k: v.grad for k, v in leaves.items()}, rz.last_forward_status()["num_pairs"])
    finally:
        rz.set_reached_only(True)
    for a, b in zip(res[True][0][1:], res[False][0][1:]):
        assert torch.equal(a, b)
    assert res[True][2] < 0.9 * res[False][2]
    for k in res[True][1]:
        scale = max(1.0, float(res[False][1][k].abs().max()))
        assert float((res[True][1][k] - res[False][1][k]).abs().max()) <= 2e-5 * scale, k
