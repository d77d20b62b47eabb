"""Gaussian adapter tail on the MI355X, through the C ABI (lsr_adapter_forward / _backward):
parity with the reference-generated vectors, with the float64 oracle at full encoder size, and
the GaussianAdapter mirror end to end."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import adapter_oracle as ao
from tests.test_adapter_cpu import GOLDEN, assert_cov_close, flat_case

pytestmark = pytest.mark.gpu


def run_kernel(inp, grads, dev, packed=False, wide_raw=None):
    """adapter_geometry on the device.  `wide_raw`: embed the 7 raw columns in a wider matrix and
    hand the kernel a strided view, as the encoder does."""
    from latentsplat_amd.gaussian_adapter import adapter_geometry
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    raw7 = np.concatenate([inp["raw_scales"], inp["raw_rotations"]], -1)
    if wide_raw:
        full = torch.zeros(raw7.shape[:-1] + (wide_raw,), device=dev)
        full[..., 2:9] = t(raw7)
        full.requires_grad_()
        raw = full[..., 2:]
    else:
        full = t(raw7).requires_grad_()
        raw = full
    coords, depths = t(inp["coordinates"]).requires_grad_(), t(inp["depths"]).requires_grad_()
    out = adapter_geometry(t(inp["extrinsics"]), t(inp["intrinsics"]), coords, depths, raw,
                           tuple(int(x) for x in inp["image_shape"]), float(inp["scale_min"]),
                           float(inp["scale_max"]), packed_covariance=packed)
    names = ("means", "covariances", "scales", "rotations")
    loss = sum((o * t(grads[n])).sum() for n, o in zip(names, out) if grads.get(n) is not None)
    din = {}
    if torch.is_tensor(loss):
        loss.backward()
        graw = full.grad[..., 2:9] if wide_raw else full.grad
        din = dict(coordinates=coords.grad.cpu().numpy(), depths=depths.grad.cpu().numpy(),
                   raw_scales=graw[..., :3].cpu().numpy(), raw_rotations=graw[..., 3:].cpu().numpy())
        if wide_raw:
            assert float(full.grad[..., :2].abs().max()) == 0 and float(full.grad[..., 9:].abs().max()) == 0
    return {n: o.detach().cpu().numpy() for n, o in zip(names, out)}, din


def rel_err(got, ref):
    return float(np.abs(got - ref).max() / max(1e-6, np.abs(ref).max()))


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[8:-4])
def test_forward_matches_reference_vectors(path, hip_device):
    z, inp = flat_case(path)
    out, _ = run_kernel(inp, {}, hip_device, wide_raw=z["raw"].shape[-1])
    b, v, r, srf, spp = z["depths"].shape
    np.testing.assert_allclose(out["means"].reshape(z["means"].shape), z["means"], rtol=2e-5, atol=2e-6)
    assert_cov_close(out["covariances"].reshape(z["covariances"].shape), z["covariances"])
    np.testing.assert_allclose(out["scales"].reshape(z["scales"].shape), z["scales"], rtol=2e-5)
    np.testing.assert_allclose(out["rotations"], z["rotations"][..., 0, :].reshape(b * v, r * srf, 4), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[8:-4])
@pytest.mark.parametrize("which", ["m", "mc", "mcs"])
def test_backward_matches_reference_vectors(path, which, hip_device):
    z, inp = flat_case(path)
    b, v, r, srf, spp = z["depths"].shape
    flat = lambda a, tail: a.reshape((b * v, r * srf, spp) + tail)
    grads = dict(means=flat(z["g_means"], (3,)))
    if "c" in which:
        grads["covariances"] = flat(z["g_covariances"], (3, 3))
    if "s" in which:
        grads["scales"] = flat(z["g_scales"], (3,))
    _, din = run_kernel(inp, grads, hip_device)
    ref_raw = z[f"d_raw_{which}"]
    assert rel_err(din["coordinates"], z[f"d_coordinates_{which}"].reshape(b * v, r * srf, 2)) <= 1e-4
    assert rel_err(din["depths"], z[f"d_depths_{which}"].reshape(b * v, r * srf, spp)) <= 1e-4
    assert rel_err(din["raw_scales"], ref_raw[..., 2:5].reshape(b * v, r * srf, 3)) <= 1e-4
    assert rel_err(din["raw_rotations"], ref_raw[..., 5:9].reshape(b * v, r * srf, 4)) <= 1e-4


def random_case(cams, rays, samples, seed, h=64, w=48):
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(cams, 4)); q /= np.linalg.norm(q, axis=-1, keepdims=True)
    x, y, z, s = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * s), 2 * (x * z + y * s),
                  2 * (x * y + z * s), 1 - 2 * (x * x + z * z), 2 * (y * z - x * s),
                  2 * (x * z - y * s), 2 * (y * z + x * s), 1 - 2 * (x * x + y * y)], -1).reshape(cams, 3, 3)
    E = np.tile(np.eye(4), (cams, 1, 1)); E[:, :3, :3] = R; E[:, :3, 3] = rng.normal(size=(cams, 3))
    K = np.tile(np.eye(3), (cams, 1, 1))
    K[:, 0, 0] = 0.6 + rng.random(cams); K[:, 1, 1] = 0.6 + rng.random(cams)
    K[:, 0, 2] = 0.45 + 0.1 * rng.random(cams); K[:, 1, 2] = 0.45 + 0.1 * rng.random(cams)
    return dict(extrinsics=E.astype(np.float32), intrinsics=K.astype(np.float32),
                coordinates=rng.random((cams, rays, 2)).astype(np.float32),
                depths=(0.5 + 20 * rng.random((cams, rays, samples))).astype(np.float32),
                raw_scales=(2 * rng.normal(size=(cams, rays, 3))).astype(np.float32),
                raw_rotations=rng.normal(size=(cams, rays, 4)).astype(np.float32),
                image_shape=np.array([h, w]), scale_min=0.5, scale_max=15.0)


@pytest.mark.parametrize("cams,rays,samples,packed", [(2, 65536, 3, False), (3, 1000, 1, True), (1, 1, 4, False), (5, 257, 2, True)])
def test_matches_float64_oracle(cams, rays, samples, packed, hip_device):
    """Encoder-size rows (2 context views x 256x256 rays x 3 samples = configs[3]'s 393 216
    Gaussians) and ragged sizes, against the oracle evaluated in float64."""
    inp = random_case(cams, rays, samples, seed=cams * 7 + samples)
    rng = np.random.default_rng(3)
    gfull = rng.normal(size=(cams, rays, samples, 3, 3)).astype(np.float32)
    iu = np.triu_indices(3)
    grads = dict(means=rng.normal(size=(cams, rays, samples, 3)).astype(np.float32),
                 covariances=gfull[..., iu[0], iu[1]] if packed else gfull,
                 scales=rng.normal(size=(cams, rays, samples, 3)).astype(np.float32),
                 rotations=rng.normal(size=(cams, rays, 4)).astype(np.float32))
    out, din = run_kernel(inp, grads, hip_device, packed=packed)
    ograds = dict(grads)
    if packed:   # packed upper triangle == the caller's triu gather of the full matrix
        g = np.zeros_like(gfull); g[..., iu[0], iu[1]] = grads["covariances"]; ograds["covariances"] = g
    ref, dref = ao.adapter_forward_backward(inp, ograds, torch.float64)
    np.testing.assert_allclose(out["means"], ref["means"], rtol=1e-5, atol=1e-5)
    cov = ref["covariances"][..., iu[0], iu[1]] if packed else ref["covariances"]
    mag = np.abs(ref["covariances"]).reshape(cams, rays, samples, 9).max(-1)
    assert (np.abs(out["covariances"] - cov).reshape(cams, rays, samples, -1).max(-1) / mag).max() <= 5e-6
    np.testing.assert_allclose(out["scales"], ref["scales"], rtol=1e-5)
    np.testing.assert_allclose(out["rotations"], ref["rotations"], rtol=1e-5, atol=1e-6)
    # gradients: per-row error relative to the row's own gradient magnitude
    for k in ("coordinates", "depths", "raw_scales", "raw_rotations"):
        a, r = din[k].reshape(cams * rays, -1), dref[k].reshape(cams * rays, -1)
        mag = np.maximum(np.abs(r).max(-1), 1e-3 * np.abs(r).max())
        assert (np.abs(a - r).max(-1) / mag).max() <= 2e-4, k


def test_mirror_class_matches_reference_vectors(hip_device):
    """GaussianAdapter.forward called exactly like encoder_epipolar.py:184-193 does."""
    from latentsplat_amd.gaussian_adapter import GaussianAdapter, GaussianAdapterCfg
    for path in GOLDEN:
        z = np.load(path)
        cdeg, fdeg, fch = (int(x) for x in z["sh_degrees"])
        ad = GaussianAdapter(GaussianAdapterCfg(float(z["scale_range"][0]), float(z["scale_range"][1]), cdeg, fdeg),
                             fch, rotate_sh=lambda sh, rot: sh).to(hip_device)
        t = lambda k: torch.tensor(z[k], device=hip_device)
        raw = t("raw").requires_grad_()
        coords, depths = t("coordinates").requires_grad_(), t("depths").requires_grad_()
        g = ad.forward(t("extrinsics")[:, :, None, None, None], t("intrinsics")[:, :, None, None, None],
                       coords[..., None, :], depths, t("opacities"), raw[..., None, 2:],
                       tuple(int(x) for x in z["image_shape"]))
        for name in ("means", "scales", "rotations", "color_harmonics", "feature_harmonics"):
            got = getattr(g, name).detach().cpu().numpy()
            assert got.shape == z[name].shape, name
            np.testing.assert_allclose(got, z[name], rtol=2e-5, atol=2e-6, err_msg=name)
        assert_cov_close(g.covariances.detach().cpu().numpy(), z["covariances"])
        assert torch.equal(g.opacities, t("opacities"))
        ((g.means * t("g_means")).sum() + (g.covariances * t("g_covariances")).sum() + (g.scales * t("g_scales")).sum()).backward()
        assert rel_err(raw.grad.cpu().numpy(), z["d_raw_mcs"]) <= 1e-4
        assert rel_err(coords.grad.cpu().numpy(), z["d_coordinates_mcs"]) <= 1e-4
        assert rel_err(depths.grad.cpu().numpy(), z["d_depths_mcs"]) <= 1e-4


def test_adapter_feeds_rasterizer_packed(hip_device):
    """cov_elems=6 output is exactly the `triu` gather the reference performs per view
    (cuda_splatting.py:148,157) of the 3x3 output."""
    inp = random_case(2, 500, 3, seed=11)
    full, _ = run_kernel(inp, {}, hip_device, packed=False)
    packed, _ = run_kernel(inp, {}, hip_device, packed=True)
    iu = np.triu_indices(3)
    # two template instantiations: identical up to FMA contraction choices of the compiler
    mag = np.abs(full["covariances"]).reshape(2, 500, 3, 9).max(-1, keepdims=True)
    assert (np.abs(full["covariances"][..., iu[0], iu[1]] - packed["covariances"]) / mag).max() <= 1e-6
    assert np.array_equal(full["means"], packed["means"])


def test_adapter_abi_errors(hip_device):
    import ctypes as C
    from latentsplat_amd import _lib
    lib = _lib.load()
    d = _lib.AdapterDims(1, 4, 1, 8, 8, 7, 0.5, 15.0, 1e-8, 7, 0, 0)       # cov_elems 7
    inp = _lib.AdapterInputs()
    out = _lib.AdapterOutputs()
    assert lib.lsr_adapter_forward(C.byref(d), C.byref(inp), C.byref(out), None) == -1
    d.cov_elems = 9
    assert lib.lsr_adapter_forward(C.byref(d), C.byref(inp), C.byref(out), None) == -2   # NULL inputs
    d.raw_stride = 5
    assert lib.lsr_adapter_forward(C.byref(d), C.byref(inp), C.byref(out), None) == -1


def test_degenerate_rows_match_oracle(hip_device):
    """Zero quaternion (normalised to 0 -> identity rotation), saturated scale logits, tiny / huge
    depths: the kernel follows the reference's arithmetic there too (no NaN, same gradients)."""
    inp = random_case(1, 8, 2, seed=4)
    inp["raw_rotations"][0, 0] = 0.0
    inp["raw_rotations"][0, 1] = [1e-12, 0, 0, 0]
    inp["raw_scales"][0, 2] = [40.0, -40.0, 0.0]
    inp["depths"][0, 3] = [1e-4, 1e4]
    inp["coordinates"][0, 4] = [0.0, 1.0]
    rng = np.random.default_rng(1)
    grads = dict(means=rng.normal(size=(1, 8, 2, 3)).astype(np.float32),
                 covariances=rng.normal(size=(1, 8, 2, 3, 3)).astype(np.float32))
    out, din = run_kernel(inp, grads, hip_device)
    ref, dref = ao.adapter_forward_backward(inp, grads, torch.float32)
    for k in out:
        assert np.isfinite(out[k]).all(), k
        np.testing.assert_allclose(out[k], ref[k], rtol=2e-5, atol=1e-6 * max(1.0, float(np.abs(ref[k]).max())), err_msg=k)
    for k in din:
        assert np.isfinite(din[k]).all(), k
        scale = max(1e-6, float(np.abs(dref[k]).max()))
        assert np.abs(din[k] - dref[k]).max() <= 2e-4 * scale, k
