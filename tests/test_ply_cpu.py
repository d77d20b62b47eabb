"""PLY export, CPU side: oracle pinned to the record table the reference's own
export_ply produced; the library's host writer emits the layout plyfile writes."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ply_oracle as po

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ply_export.npz")


def test_oracle_matches_reference_table():
    z = np.load(GOLDEN)
    got = po.ply_vertices(z["extrinsics"], z["means"], z["scales"], z["rotations"], z["harmonics"], z["opacities"])
    assert got.shape == z["vertices"].shape
    np.testing.assert_allclose(got, z["vertices"], rtol=1e-6, atol=1e-6)


def test_host_writer_layout(tmp_path):
    from latentsplat_amd import _lib
    from latentsplat_amd.ply_export import construct_list_of_attributes
    z = np.load(GOLDEN)
    assert construct_list_of_attributes(0) == list(z["names"])
    assert len(construct_list_of_attributes(5)) == 22
    lib = _lib.load()
    table = np.ascontiguousarray(z["vertices"], np.float32)
    path = tmp_path / "g.ply"
    assert lib.lsr_ply_write_host(os.fsencode(str(path)), table.ctypes.data_as(C.c_void_p), table.shape[0]) == 0
    names, data = po.read_ply(path)
    assert names == list(z["names"]) and np.array_equal(data, table)
    raw = open(path, "rb").read()
    header = raw[: raw.index(b"end_header\n") + len(b"end_header\n")]
    assert header.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 700\nproperty float x\n")
    assert len(raw) == len(header) + 700 * 17 * 4
    # empty export and unwritable path
    assert lib.lsr_ply_write_host(os.fsencode(str(tmp_path / "e.ply")), None, 0) == 0
    assert po.read_ply(tmp_path / "e.ply")[1].shape == (0, 17)
    assert lib.lsr_ply_write_host(os.fsencode(str(tmp_path / "no_dir" / "x.ply")), table.ctypes.data_as(C.c_void_p), 1) == -1


def test_no_cpu_fallback():
    import torch
    from latentsplat_amd import _lib
    from latentsplat_amd.ply_export import pack_vertices
    with pytest.raises(_lib.LsrError):
        pack_vertices(torch.eye(4), torch.zeros(2, 3), torch.ones(2, 3), torch.ones(2, 4), torch.zeros(2, 3, 1), torch.ones(2))
