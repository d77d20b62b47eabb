"""CPU oracle of the .ply export — TEST INFRASTRUCTURE ONLY (tests/ only).

numpy / scipy restatement of /root/reference/src/model/ply_export.py:26-92 (recentre by the
median :35, rescale by the 0.95-quantile :38-41, viewer rotation :43-66, quaternion rotation
through scipy :68-73, DC band :77, vertex record :79-90) and a minimal reader for the binary
little-endian PLY layout plyfile writes.

Parity status: PINNED — tests/golden/ply_export.npz holds the `elements` array the reference's own
`export_ply` handed to plyfile (captured with a stand-in `plyfile` module by
tests/golden/make_golden_ply.py, plyfile itself is not installed in the build image).
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R


def ply_vertices(extrinsics, means, scales, rotations, harmonics, opacities) -> np.ndarray:
    t = lambda a: torch.as_tensor(a, dtype=torch.float32)
    extrinsics, means, scales = t(extrinsics), t(means), t(scales)
    means = means - means.median(dim=0).values
    scale_factor = means.abs().quantile(0.95, dim=0).max()
    means, scales = means / scale_factor, scales / scale_factor
    rotation = torch.tensor([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=torch.float32)
    rotation = torch.tensor(R.from_rotvec([0, 0, -45], True).as_matrix(), dtype=torch.float32) @ rotation
    rotation = rotation @ extrinsics[:3, :3].inverse()
    means = means @ rotation.T
    rot = rotation.numpy() @ R.from_quat(np.asarray(rotations)).as_matrix()
    x, y, z, w = R.from_matrix(rot).as_quat().T
    cols = (means.numpy(), np.zeros_like(means.numpy()), np.asarray(harmonics)[..., 0],
            np.asarray(opacities)[..., None], scales.log().numpy(), np.stack((w, x, y, z), -1))
    return np.concatenate(cols, axis=1).astype(np.float32)


def read_ply(path):
    """-> (property names, (n, len(names)) float32 array) of a binary little-endian float PLY."""
    with open(path, "rb") as f:
        assert f.readline() == b"ply\n"
        assert f.readline() == b"format binary_little_endian 1.0\n"
        n, names = None, []
        while True:
            line = f.readline().decode().strip()
            if line == "end_header":
                break
            parts = line.split()
            if parts[0] == "element":
                assert parts[1] == "vertex" and n is None
                n = int(parts[2])
            elif parts[0] == "property":
                assert parts[1] == "float"
                names.append(parts[2])
        data = np.frombuffer(f.read(), dtype="<f4")
    return names, data.reshape(n, len(names))
