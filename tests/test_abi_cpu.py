"""The C-ABI shared library loads without a GPU and exports every symbol the public header
declares; host-only entry points (sizing, validation) behave as documented."""
import ctypes as C
import os
import re

import pytest

from latentsplat_amd import _lib
from latentsplat_amd._lib import Dims, Inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = ""
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text += open(os.path.join(ROOT, "include", h)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lsr_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert set(names) == set(_lib.EXPORTS)
    assert lib.lsr_abi_version() == 10


def _dims(**kw):
    base = dict(num_views=2, num_gaussians=1000, height=64, width=64, feat_channels=4, color_mode=0,
                sh_degree=0, sh_coeffs=0, vs_means=0, vs_cov=0, vs_opac=0, vs_color=0, vs_feat=0,
                cov_elems=6, feat_mode=0, feat_sh_degree=0, feat_sh_coeffs=0)
    base.update(kw)
    return Dims(**base)


def test_workspace_sizes():
    lib = _lib.load()
    d = _dims()
    g, i = lib.lsr_geom_workspace_bytes(C.byref(d)), lib.lsr_image_workspace_bytes(C.byref(d))
    assert g >= 2 * 1000 * (16 + 16 + 8) and i >= 2 * 64 * 64 * 8
    assert lib.lsr_binning_workspace_bytes(C.byref(d), 5000, 100) >= 5000 * 12
    # lists longer than the LDS capacity need the merge scratch
    assert lib.lsr_binning_workspace_bytes(C.byref(d), 50000, 20000) > lib.lsr_binning_workspace_bytes(C.byref(d), 50000, 100)
    d2 = _dims(num_views=4)
    assert lib.lsr_geom_workspace_bytes(C.byref(d2)) > g
    assert lib.lsr_grad_workspace_bytes(C.byref(d)) >= 2 * 1000 * (8 + 16 + 4)


def test_key_segments_in_the_geometry_workspace():
    """Round 5 (ABI v8): calls that qualify for single-pass binning (byte tile coordinates, <= 1024 tiles) carry one
    fixed-capacity key segment per (view, tile) at the end of the geometry workspace — 8 bytes x min(8192, G rounded up to
    64) keys each — and the size is a pure function of the dims (what `query size, then run` relies on)."""
    lib = _lib.load()
    small, big = _dims(num_gaussians=1000), _dims(num_gaussians=100_000)
    tiles = 2 * 4 * 4                                     # 2 views of 64 x 64
    g_small, g_big = lib.lsr_geom_workspace_bytes(C.byref(small)), lib.lsr_geom_workspace_bytes(C.byref(big))
    assert g_small == lib.lsr_geom_workspace_bytes(C.byref(small))
    assert g_small >= tiles * 1024 * 8                    # capacity = 1000 rounded up to 64 = 1024 keys
    assert g_big >= tiles * 8192 * 8                      # ... capped at the second sort tier
    # images beyond 1024 tiles (here 2048 x 2048 = 16 384) fall back to the two-phase binning: no segments
    huge = _dims(num_gaussians=1000, height=2048, width=2048)
    per_tile = (lib.lsr_geom_workspace_bytes(C.byref(huge)) - g_small) / (2 * (128 * 128 - 16))
    assert per_tile < 1024 * 8 / 4                        # far less than a segment per tile
    # round 6 (ABI v9): the caller's hint (the longest tile list it expects) sizes the segments: real device memory
    hinted = _dims(num_gaussians=100_000, seg_cap_hint=3000)
    g_hint = lib.lsr_geom_workspace_bytes(C.byref(hinted))
    assert g_big - g_hint == tiles * (8192 - 3008) * 8    # 3000 rounded up to 64 keys
    assert lib.lsr_geom_workspace_bytes(C.byref(_dims(num_gaussians=100_000, seg_cap_hint=10))) == g_big - tiles * (8192 - 256) * 8
    assert lib.lsr_geom_workspace_bytes(C.byref(_dims(num_gaussians=100_000, seg_cap_hint=10 ** 6))) == g_big


def test_speculative_sort_tier_hint():
    """The sort-tier boundary the speculative forward asks for: the first power-of-two tier at or above 1.1 x the longest list."""
    from latentsplat_amd.rasterizer import _tier_hint
    assert [_tier_hint(x) for x in (0, 900, 931, 3289, 3724, 7000, 7500, 20000)] == [1024, 1024, 2048, 4096, 8192, 8192, 16384, 32768]


@pytest.mark.parametrize("bad", [
    dict(num_views=0), dict(height=0), dict(feat_channels=33), dict(feat_channels=0, color_mode=0),
    dict(color_mode=1, sh_degree=5, sh_coeffs=36), dict(color_mode=1, sh_degree=2, sh_coeffs=4),
    dict(vs_means=7), dict(vs_feat=5), dict(cov_elems=7), dict(feat_mode=2),
    dict(feat_mode=1, feat_sh_degree=1, feat_sh_coeffs=3), dict(color_sh_convention=2),
    dict(forward_flags=16), dict(seg_cap_hint=-1),
])
def test_invalid_dims_are_rejected(bad):
    lib = _lib.load()
    d = _dims(**bad)
    assert lib.lsr_geom_workspace_bytes(C.byref(d)) == 0
    npairs, maxtile = C.c_int64(0), C.c_int32(0)
    rc = lib.lsr_forward_prepare(C.byref(d), C.byref(Inputs()), None, None, C.byref(npairs), C.byref(maxtile), None)
    assert rc == -1 and b"invalid" in lib.lsr_error_string(rc)


def test_unsupported_fused_sh_shapes_are_reported():
    lib = _lib.load()
    for bad in (dict(feat_mode=1, feat_sh_degree=3, feat_sh_coeffs=16), dict(feat_channels=32, feat_mode=1, feat_sh_degree=2, feat_sh_coeffs=9),
                dict(num_views=1 << 14, height=4096, width=4096),       # view*tile ids must fit the 28-bit work-item field
                dict(num_views=1 << 12, num_gaussians=1 << 20)):        # 32-bit pair counts / tile offsets
        d = _dims(**bad)
        npairs, maxtile = C.c_int64(0), C.c_int32(0)
        assert lib.lsr_forward_prepare(C.byref(d), C.byref(Inputs()), None, None, C.byref(npairs), C.byref(maxtile), None) == -5


def test_null_pointers_are_rejected_before_any_gpu_work():
    lib = _lib.load()
    d = _dims()
    npairs, maxtile = C.c_int64(0), C.c_int32(0)
    assert lib.lsr_forward_prepare(C.byref(d), C.byref(Inputs()), None, None, C.byref(npairs), C.byref(maxtile), None) == -2
    assert lib.lsr_forward_render(C.byref(d), C.byref(Inputs()), None, None, None, 0, 0, None, None) == -2


def test_product_path_refuses_cpu_tensors():
    import torch
    from latentsplat_amd.rasterizer import rasterize_views
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rasterize_views(torch.zeros(1, 40), 16, 16, 0, torch.zeros(4, 3), torch.zeros(4, 6), torch.zeros(4, 1),
                        features=torch.zeros(4, 4))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_SO", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LsrError, match="no CPU"):
        _lib.load()


def test_mark_visible_matches_the_oracle_near_plane_cull():
    """GaussianRasterizer.markVisible (upstream API, unused by the reference): view-space z > 0.2, with the
    transposed view matrix of the settings tuple — the same Gaussians the oracle's preprocess keeps past its
    near-plane test."""
    import numpy as np
    import torch
    import diff_gaussian_rasterization as dgr
    from tests import util
    sc = util.make_scene(4000, image_size=64, views=1, color_sh_degree=0, feature_channels=None)
    sc.means[:500, 2] -= 2.0          # push some behind / next to the camera
    bi = util.boundary_inputs(sc, 64, 64)
    c = bi["cams"]
    rs = dgr.GaussianRasterizationSettings(
        image_height=64, image_width=64, tanfovx=float(c.tan_fov_x[0]), tanfovy=float(c.tan_fov_y[0]),
        bg=bi["bg"][0], scale_modifier=1.0, viewmatrix=c.view_matrix[0], projmatrix=c.full_projection[0],
        sh_degree=0, campos=c.campos[0], prefiltered=False, debug=False)
    vis = dgr.GaussianRasterizer(rs).markVisible(bi["means"][0])
    vm = c.view_matrix[0].numpy().astype(np.float32).reshape(16)
    m = bi["means"][0].numpy().astype(np.float32)
    z = vm[2] * m[:, 0] + vm[6] * m[:, 1] + vm[10] * m[:, 2] + vm[14]
    want = z > 0.2
    assert vis.dtype == torch.bool and vis.shape == (4000,)
    assert (vis.numpy() != want).sum() <= 2 and 0 < want.sum() < 4000     # (float summation order at the threshold)
