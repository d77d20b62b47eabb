"""Latent epilogue, CPU side: oracle pinned to vectors produced with the
reference's own posterior classes; the filter weights restated in the HIP kernel equal ATen's."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import latent_oracle as lo

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "latent_*.npz")))


def load_case(path):
    z = np.load(path)
    b, v = (int(x) for x in z["bv"])
    t = lambda k: torch.tensor(z[k])
    return z, b, v, dict(features=t("feature").unflatten(0, (b, v)), mask=t("mask").unflatten(0, (b, v)),
                         noise=t("noise"), color=t("color").unflatten(0, (b, v)), factor=int(z["factor"]),
                         variational=bool(z["variational"]))


def test_fixtures_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[7:-4])
def test_oracle_matches_reference(path):
    z, b, v, inp = load_case(path)
    inp["features"].requires_grad_()
    out = lo.latent_epilogue(**inp)
    np.testing.assert_allclose(out["logvar"].detach().numpy(), z["logvar"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["sample"].detach().numpy(), z["sample"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["skip"].detach().numpy(), z["skip"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["z"].detach().numpy(), z["z"], rtol=1e-5, atol=1e-6)
    ((out["z"] * torch.tensor(z["g_z"])).sum() + (out["skip"] * torch.tensor(z["g_skip"])).sum()).backward()
    np.testing.assert_allclose(inp["features"].grad.flatten(0, 1).numpy(), z["d_feature_all"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n_in,n_out", [(256, 32), (64, 8), (40, 10), (72, 18), (30, 7), (17, 17), (100, 33), (9, 1)])
def test_restated_filter_weights_equal_aten(n_in, n_out):
    """The weight formula documented in csrc/latent_epilogue.hip reproduces ATen's
    _upsample_bilinear2d_aa (the operator behind torchvision's resize(antialias=True))."""
    import torch.nn.functional as F
    W = lo.aa_weight_matrix(n_in, n_out)
    x = torch.randn(3, 5, n_in, generator=torch.Generator().manual_seed(n_in))
    ref = F.interpolate(x[None], size=(5, n_out), mode="bilinear", align_corners=False, antialias=True)[0]
    np.testing.assert_allclose((x @ W.T).numpy(), ref.numpy(), rtol=1e-5, atol=2e-6)
    assert np.allclose(W.sum(1).numpy(), 1.0, atol=1e-6)


def test_no_cpu_fallback():
    from latentsplat_amd import _lib
    from latentsplat_amd.decoder.latent_epilogue import get_scaled_size, rescale, sample_rescale_skip
    from fractions import Fraction
    assert get_scaled_size(Fraction(1, 8), (256, 64)) == (32, 8)
    with pytest.raises(ValueError):
        get_scaled_size(Fraction(1, 8), (100, 64))
    with pytest.raises(_lib.LsrError):
        rescale(torch.zeros(2, 3, 16, 16), Fraction(1, 4))
    with pytest.raises(_lib.LsrError):
        sample_rescale_skip(torch.zeros(1, 1, 4, 16, 16), torch.zeros(1, 1, 16, 16))
