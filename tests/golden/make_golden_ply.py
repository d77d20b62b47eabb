#!/usr/bin/env python
"""Generate tests/golden/ply_export.npz by running the reference's own ``export_ply``
(/root/reference/src/model/ply_export.py) in this container on seeded inputs, with a stand-in
``plyfile`` module (not installed here) that captures the structured ``elements`` array the
reference would have written.  Only the vectors travel."""
from __future__ import annotations

import importlib
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
CAPTURED = {}


def main():
    from make_golden import _install_stubs, _recording_module
    _install_stubs(_recording_module())
    pf = types.ModuleType("plyfile")

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            CAPTURED["elements"], CAPTURED["name"] = elements.copy(), name
            return (elements, name)

    class PlyData:
        def __init__(self, elements):
            self.elements = elements

        def write(self, path):
            CAPTURED["path"] = str(path)

    pf.PlyElement, pf.PlyData = PlyElement, PlyData
    sys.modules["plyfile"] = pf
    mod = importlib.import_module("src.model.ply_export")
    g = torch.Generator().manual_seed(31)
    n = 700
    q = torch.randn(4, generator=g)
    q = q / q.norm()
    x, y, z, w = q.tolist()
    E = torch.eye(4)
    E[:3, :3] = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    E[:3, 3] = torch.randn(3, generator=g)
    means = torch.randn(n, 3, generator=g) * torch.tensor([2.0, 0.5, 4.0]) + torch.tensor([1.0, -2.0, 6.0])
    scales = torch.rand(n, 3, generator=g) * 0.2 + 0.01
    rotations = torch.randn(n, 4, generator=g)
    rotations = rotations / rotations.norm(dim=-1, keepdim=True)
    harmonics = torch.randn(n, 3, 9, generator=g)
    opac = torch.rand(n, generator=g)
    mod.export_ply(E, means, scales, rotations, harmonics, opac, Path("/tmp/lsr_golden_unused/x.ply"))
    el = CAPTURED["elements"]
    names = list(el.dtype.names)
    assert names == mod.construct_list_of_attributes(0)
    table = np.stack([el[k] for k in names], -1).astype(np.float32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ply_export.npz"), extrinsics=E.numpy(),
                        means=means.numpy(), scales=scales.numpy(), rotations=rotations.numpy(),
                        harmonics=harmonics.numpy(), opacities=opac.numpy(), vertices=table,
                        names=np.array(names), element=np.array(CAPTURED["name"]))
    print(table.shape, names)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("the reference is only available in the build container")
    sys.path.insert(0, REF)
    main()
