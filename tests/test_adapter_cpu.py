"""Gaussian adapter tail, CPU side: the oracle restatement is pinned against
vectors produced by the reference's own GaussianAdapter (tests/golden/make_golden_adapter.py), and the
host mirror's non-kernel logic matches them."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import adapter_oracle as ao

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "adapter_*.npz")))


def flat_case(path):
    """Fixture -> the flattened (cam, rays, samples) arguments of the oracle / C ABI."""
    z = np.load(path)
    b, v, r, srf, spp = z["depths"].shape
    raw = z["raw"]                                             # (b v r srf 2+d_in): Linear output
    return z, dict(
        extrinsics=z["extrinsics"].reshape(b * v, 4, 4), intrinsics=z["intrinsics"].reshape(b * v, 3, 3),
        coordinates=z["coordinates"].reshape(b * v, r * srf, 2), depths=z["depths"].reshape(b * v, r * srf, spp),
        raw_scales=raw[..., 2:5].reshape(b * v, r * srf, 3), raw_rotations=raw[..., 5:9].reshape(b * v, r * srf, 4),
        image_shape=z["image_shape"], scale_min=z["scale_range"][0], scale_max=z["scale_range"][1])


def assert_cov_close(got, ref, tol=5e-6):
    """Covariances: error relative to each matrix's own magnitude (near-zero entries are sums that
    cancel, so an elementwise relative test is meaningless)."""
    mag = np.abs(ref).reshape(ref.shape[:-2] + (9,)).max(-1)[..., None, None]
    assert (np.abs(got - ref) / mag).max() <= tol


def test_fixtures_present():
    assert len(GOLDEN) >= 2


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[8:-4])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_oracle_matches_reference_forward(path, dtype):
    z, inp = flat_case(path)
    out, _ = ao.adapter_forward_backward(inp, {}, dtype)
    b, v, r, srf, spp = z["depths"].shape
    np.testing.assert_allclose(out["means"].reshape(z["means"].shape), z["means"], rtol=2e-5, atol=2e-6)
    assert_cov_close(out["covariances"].reshape(z["covariances"].shape), z["covariances"])
    np.testing.assert_allclose(out["scales"].reshape(z["scales"].shape), z["scales"], rtol=2e-5)
    ref_rot = z["rotations"][..., 0, :].reshape(b * v, r * srf, 4)      # broadcast over spp in the reference
    np.testing.assert_allclose(out["rotations"], ref_rot, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[8:-4])
@pytest.mark.parametrize("which", ["m", "mc", "mcs"])
def test_oracle_matches_reference_backward(path, which):
    z, inp = flat_case(path)
    b, v, r, srf, spp = z["depths"].shape
    flat = lambda a, tail: a.reshape((b * v, r * srf, spp) + tail)
    grads = dict(means=flat(z["g_means"], (3,)))
    if "c" in which:
        grads["covariances"] = flat(z["g_covariances"], (3, 3))
    if "s" in which:
        grads["scales"] = flat(z["g_scales"], (3,))
    _, din = ao.adapter_forward_backward(inp, grads, torch.float64)
    ref_raw = z[f"d_raw_{which}"]
    scale = lambda a: max(1e-6, float(np.abs(a).max()))
    for got, ref in ((din["coordinates"], z[f"d_coordinates_{which}"].reshape(b * v, r * srf, 2)),
                     (din["depths"], z[f"d_depths_{which}"].reshape(b * v, r * srf, spp)),
                     (din["raw_scales"], ref_raw[..., 2:5].reshape(b * v, r * srf, 3)),
                     (din["raw_rotations"], ref_raw[..., 5:9].reshape(b * v, r * srf, 4))):
        assert np.abs(got - ref).max() <= 2e-4 * scale(ref)
    # nothing else of the Linear output receives geometry gradient
    assert np.abs(ref_raw[..., :2]).max() == 0 and np.abs(ref_raw[..., 9:]).max() == 0


def test_mirror_host_logic_matches_reference():
    from latentsplat_amd.gaussian_adapter import GaussianAdapter, GaussianAdapterCfg
    z = np.load(GOLDEN[0])
    cdeg, fdeg, fch = (int(x) for x in z["sh_degrees"])
    ad = GaussianAdapter(GaussianAdapterCfg(float(z["scale_range"][0]), float(z["scale_range"][1]), cdeg, fdeg), fch,
                         rotate_sh=lambda sh, rot: sh)
    assert ad.d_in == z["raw"].shape[-1] - 2
    assert ad.d_color_sh == (cdeg + 1) ** 2 and ad.d_feature_sh == (fdeg + 1) ** 2
    h, w = (int(x) for x in z["image_shape"])
    K = torch.tensor(z["intrinsics"])
    mult = ad.get_scale_multiplier(K, 1 / torch.tensor((w, h), dtype=torch.float32))
    np.testing.assert_allclose(mult.numpy(), ao.scale_multiplier(K, h, w).numpy(), rtol=1e-6)
    # SH masks: harmonics of the fixture (identity rotation) = raw coefficients * mask
    raw = torch.tensor(z["raw"])[..., None, 2:]
    color = raw[..., 7:7 + 3 * ad.d_color_sh].reshape(*raw.shape[:-1], 3, ad.d_color_sh) * ad.color_sh_mask
    np.testing.assert_allclose(color.expand(z["color_harmonics"].shape).numpy(), z["color_harmonics"], rtol=1e-6)


def test_no_cpu_fallback():
    from latentsplat_amd import _lib
    from latentsplat_amd.gaussian_adapter import adapter_geometry
    with pytest.raises(_lib.LsrError):
        adapter_geometry(torch.eye(4)[None], torch.eye(3)[None], torch.zeros(1, 2, 2), torch.ones(1, 2, 1),
                         torch.zeros(1, 2, 7), (4, 4), 0.5, 15.0)
