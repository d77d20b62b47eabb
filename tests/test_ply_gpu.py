"""PLY export on the MI355X (lsr_ply_pack + host writer) against the reference's record table."""
import os

import numpy as np
import pytest
import torch

from oracle import ply_oracle as po
from tests.test_ply_cpu import GOLDEN

pytestmark = pytest.mark.gpu


def assert_table_close(got, ref):
    np.testing.assert_allclose(got[:, :13], ref[:, :13], rtol=2e-5, atol=2e-5)
    # quaternions: same rotation; the sign is expected to agree as well
    dq = np.minimum(np.abs(got[:, 13:] - ref[:, 13:]).max(1), np.abs(got[:, 13:] + ref[:, 13:]).max(1))
    assert dq.max() <= 1e-5
    flips = (np.abs(got[:, 13:] - ref[:, 13:]).max(1) > 1e-3).sum()
    assert flips <= max(1, got.shape[0] // 10000), flips


def test_pack_matches_reference_table(hip_device, tmp_path):
    from latentsplat_amd.ply_export import export_ply, pack_vertices
    z = np.load(GOLDEN)
    args = [torch.tensor(z[k], device=hip_device) for k in ("extrinsics", "means", "scales", "rotations", "harmonics", "opacities")]
    assert_table_close(pack_vertices(*args).cpu().numpy(), z["vertices"])
    path = tmp_path / "sub" / "scene.ply"
    export_ply(*args, path)
    names, data = po.read_ply(path)
    assert names == list(z["names"])
    assert_table_close(data, z["vertices"])


def test_pack_matches_oracle_at_size(hip_device):
    from latentsplat_amd.ply_export import pack_vertices
    g = torch.Generator().manual_seed(8)
    n = 393_216
    E = torch.eye(4)
    Q = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    E[:3, :3] = Q * torch.sign(torch.linalg.det(Q))          # proper rotation
    E[:3, 3] = torch.randn(3, generator=g)
    means = torch.randn(n, 3, generator=g) * 3 + 1
    scales = torch.rand(n, 3, generator=g) * 0.1 + 1e-3
    rot = torch.randn(n, 4, generator=g)
    sh = torch.randn(n, 3, 25, generator=g)
    op = torch.rand(n, generator=g)
    got = pack_vertices(*[t.to(hip_device) for t in (E, means, scales, rot, sh, op)]).cpu().numpy()
    ref = po.ply_vertices(E.numpy(), means.numpy(), scales.numpy(), rot.numpy(), sh.numpy(), op.numpy())
    assert_table_close(got, ref)
