"""Latent epilogue on the MI355X through the C ABI (lsr_latent_forward / _backward): parity with
the reference-generated vectors and with the oracle at the full 16 x 4 x 256 x 256 size."""
import os
from fractions import Fraction

import numpy as np
import pytest
import torch

from oracle import latent_oracle as lo
from tests.test_latent_cpu import GOLDEN, load_case

pytestmark = pytest.mark.gpu


def run(inp, dev, g_z=None, g_skip=None, with_color=True):
    from latentsplat_amd.decoder.latent_epilogue import sample_rescale_skip
    feats = inp["features"].to(dev).requires_grad_()
    out = sample_rescale_skip(feats, inp["mask"].to(dev), inp["color"].to(dev) if with_color else None,
                              inp["factor"], inp["variational"], noise=inp["noise"].to(dev))
    grad = None
    if g_z is not None or g_skip is not None:
        loss = 0
        if g_z is not None:
            loss = loss + (out.z * g_z.to(dev)).sum()
        if g_skip is not None:
            loss = loss + ((out.skip_z if with_color else out.latent_sample) * g_skip.to(dev)).sum()
        loss.backward()
        grad = feats.grad.cpu()
    return out, grad


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[7:-4])
def test_matches_reference_vectors(path, hip_device):
    z, b, v, inp = load_case(path)
    out, grad = run(inp, hip_device, torch.tensor(z["g_z"]), torch.tensor(z["g_skip"]))
    np.testing.assert_allclose(out.latent_sample.detach().cpu().numpy(), z["sample"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out.skip_z.detach().cpu().numpy(), z["skip"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out.z.detach().cpu().numpy(), z["z"], rtol=2e-5, atol=2e-5)
    lv = out.logvar.cpu().numpy()
    np.testing.assert_allclose(np.broadcast_to(lv, z["logvar"].shape), z["logvar"], rtol=2e-5, atol=2e-5)
    scale = float(np.abs(z["d_feature_all"]).max())
    assert np.abs(grad.flatten(0, 1).numpy() - z["d_feature_all"]).max() <= 1e-4 * scale
    _, grad_z = run(inp, hip_device, torch.tensor(z["g_z"]), None)
    assert np.abs(grad_z.flatten(0, 1).numpy() - z["d_feature_z"]).max() <= 1e-4 * float(np.abs(z["d_feature_z"]).max())


@pytest.mark.parametrize("shape", [(4, 4, 4, 256, 256, 8, False), (1, 3, 8, 128, 96, 4, True), (2, 2, 1, 24, 1000, 2, False),
                                   (1, 1, 2, 16, 16, 1, False), (1, 2, 4, 64, 64, 64, False)])
def test_matches_oracle_at_size(shape, hip_device):
    """BASELINE configs[4]'s per-GPU batch (16 views x 4 latent channels x 256^2, factor 8) and
    ragged / extreme shapes (factor 1 = identity resize, factor = whole image)."""
    b, v, C, H, W, f, variational = shape
    g = torch.Generator().manual_seed(H * W + C)
    fch = 2 * C if variational else C
    inp = dict(features=torch.randn(b, v, fch, H, W, generator=g), mask=torch.rand(b, v, H, W, generator=g) ** 0.2,
               noise=torch.randn(b, v, C, H, W, generator=g), color=torch.rand(b, v, 3, H, W, generator=g), factor=f,
               variational=variational)
    g_z = torch.randn(b, v, C, H // f, W // f, generator=g)
    g_skip = torch.randn(b, v, 3 + C, H, W, generator=g)
    out, grad = run(inp, hip_device, g_z, g_skip)
    ref_in = dict(inp)
    ref_in["features"] = inp["features"].clone().requires_grad_()
    ref = lo.latent_epilogue(**ref_in)
    ((ref["z"] * g_z).sum() + (ref["skip"] * g_skip).sum()).backward()
    assert torch.allclose(out.skip_z.detach().cpu(), ref["skip"].detach(), rtol=2e-5, atol=2e-5)
    assert torch.allclose(out.z.detach().cpu(), ref["z"].detach(), rtol=2e-5, atol=2e-5)
    assert torch.allclose(grad, ref_in["features"].grad, rtol=1e-4, atol=1e-4)
    # colour-less variant: the sample alone
    out2, grad2 = run(inp, hip_device, g_z, g_skip[:, :, 3:], with_color=False)
    assert out2.skip_z is None and torch.equal(out2.latent_sample, out.latent_sample)
    assert torch.allclose(grad2, grad)


def test_rescale_and_deterministic(hip_device):
    from latentsplat_amd.decoder.latent_epilogue import rescale, sample_rescale_skip
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 5, 48, 80, generator=g)
    xd = x.to(hip_device).requires_grad_()
    y = rescale(xd, Fraction(1, 8))
    ref_x = x.clone().requires_grad_()
    ref = lo.rescale(ref_x, 8)
    assert y.shape == ref.shape and torch.allclose(y.detach().cpu(), ref.detach(), rtol=2e-5, atol=2e-5)
    gy = torch.randn(ref.shape, generator=g)
    (y * gy.to(hip_device)).sum().backward()
    (ref * gy).sum().backward()
    assert torch.allclose(xd.grad.cpu(), ref_x.grad, rtol=1e-4, atol=1e-5)
    # deterministic = posterior mean; opaque pixels (mask = 1) have vanishing variance
    feats, mask = torch.randn(1, 2, 4, 32, 32, generator=g).to(hip_device), torch.ones(1, 2, 32, 32, device=hip_device)
    det = sample_rescale_skip(feats, mask, None, 8, deterministic=True)
    assert torch.equal(det.latent_sample, feats)
    noisy = sample_rescale_skip(feats, mask, None, 8)
    assert float((noisy.latent_sample - feats).abs().max()) < 1e-5 and float(noisy.logvar.max()) == -30.0


def test_decoder_output_epilogue_chain(hip_device):
    """DecoderSplattingCUDA.forward -> epilogue, gradients reaching the Gaussians."""
    from latentsplat_amd import decoder as dec
    from latentsplat_amd.decoder.latent_epilogue import decoder_output_epilogue
    from latentsplat_amd.synthetic import make_scene
    sc = make_scene(2000, image_size=64, views=2, color_sh_degree=1, feature_channels=4, feature_sh_degree=1, seed=3).to(hip_device)
    leaf = lambda t: t[None].contiguous().requires_grad_(True)
    gauss = dec.Gaussians(leaf(sc.means), leaf(sc.covariances), leaf(sc.opacities), leaf(sc.color_sh), leaf(sc.feature_sh))
    d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), [0.0, 0.0, 0.0]).to(hip_device)
    out = d.forward(gauss, sc.extrinsics[None], sc.intrinsics[None], sc.near[None], sc.far[None], (64, 64))
    torch.manual_seed(1)
    ep = decoder_output_epilogue(out, 8)
    torch.manual_seed(1)
    ref_sample = out.feature_posterior.sample()            # host mirror of the reference class
    assert torch.allclose(ep.latent_sample, ref_sample, rtol=1e-5, atol=1e-5)
    assert ep.z.shape == (1, 2, 4, 8, 8) and ep.skip_z.shape == (1, 2, 7, 64, 64)
    assert torch.equal(ep.skip_z[:, :, :3], out.color)
    ep.z.square().sum().backward()
    assert gauss.feature_harmonics.grad is not None and float(gauss.feature_harmonics.grad.abs().sum()) > 0
    assert gauss.color_harmonics.grad is None or float(gauss.color_harmonics.grad.abs().sum()) == 0


def test_latent_abi_errors(hip_device):
    import ctypes as C
    from latentsplat_amd import _lib
    lib = _lib.load()
    d = _lib.LatentDims(1, 4, 16, 16, 2, 2, 5, 0, -30.0, 20.0, 0, 0)      # bad logvar mode
    assert lib.lsr_latent_forward(C.byref(d), C.byref(_lib.LatentInputs()), C.byref(_lib.LatentOutputs()), None) == -1
    d.logvar_mode = 0
    assert lib.lsr_latent_forward(C.byref(d), C.byref(_lib.LatentInputs()), C.byref(_lib.LatentOutputs()), None) == -2
    d.width = 2000
    assert lib.lsr_latent_forward(C.byref(d), C.byref(_lib.LatentInputs()), C.byref(_lib.LatentOutputs()), None) == -5


def test_extreme_masks_and_nan_propagation(hip_device):
    """mask = 1 (opaque: logvar clamps to -30), mask = 0 (empty: unit variance), and a mask slightly
    above 1 (cannot come out of the rasterizer, but the reference would give NaN there: so do we)."""
    from latentsplat_amd.decoder.latent_epilogue import sample_rescale_skip
    feats = torch.zeros(1, 1, 2, 16, 16, device=hip_device)
    mask = torch.zeros(1, 1, 16, 16, device=hip_device)
    mask[0, 0, 0, 0], mask[0, 0, 0, 1], mask[0, 0, 0, 2] = 1.0, 0.0, 1.0 + 1e-3
    noise = torch.ones(1, 1, 2, 16, 16, device=hip_device)
    ep = sample_rescale_skip(feats, mask, None, 8, noise=noise)
    ref = lo.latent_epilogue(feats.cpu(), mask.cpu(), noise.cpu(), None, 8)
    s, r = ep.latent_sample.cpu(), ref["sample"]
    assert torch.isnan(s[0, 0, :, 0, 2]).all() and torch.isnan(r[0, 0, :, 0, 2]).all()
    ok = ~torch.isnan(r)
    assert torch.allclose(s[ok], r[ok], rtol=2e-5, atol=1e-6)
    assert float(ep.logvar[0, 0, 0, 0, 0]) == -30.0 and abs(float(ep.logvar[0, 0, 0, 0, 1])) < 1e-6
    assert abs(float(s[0, 0, 0, 0, 0]) - float(torch.exp(torch.tensor(-15.0)))) < 1e-9     # std = e^-15
