"""3DGS ``.ply`` export — mirror of /root/reference/src/model/ply_export.py (``export_ply`` :26-92,
``construct_list_of_attributes`` :13-23) on the MI355X: the per-Gaussian transform and the 17-float
vertex packing run as one HIP kernel (csrc/ply.hip, C ABI include/lsr_ply.h), the file is written
by the library's host writer.  ROCm float32 tensors only; no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch
from torch import Tensor

from . import _lib
from ._lib import PLY_VERTEX_FLOATS, PlyInputs


def construct_list_of_attributes(num_rest: int) -> list[str]:
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(num_rest)] + ["opacity"]
    return names + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]


def pack_vertices(extrinsics: Tensor, means: Tensor, scales: Tensor, rotations: Tensor, harmonics: Tensor,
                  opacities: Tensor) -> Tensor:
    """(gaussian, 17) float32 vertex records on the device, in the order of
    ``construct_list_of_attributes(0)``."""
    tensors = [extrinsics, means, scales, rotations, harmonics, opacities]
    for t in tensors:
        if not t.is_cuda or t.dtype != torch.float32:
            raise _lib.LsrError("export_ply needs float32 ROCm tensors (no CPU fallback)")
    lib = _lib.load()
    extrinsics, means, scales, rotations, harmonics, opacities = (t.detach().contiguous() for t in tensors)
    n = means.shape[0]
    if harmonics.dim() != 3 or harmonics.shape[1] != 3:
        raise _lib.LsrError("harmonics must be (gaussian, 3, d_sh)")
    # the two global statistics of ply_export.py:35-41 (torch reductions on the device)
    center = means.median(dim=0).values.contiguous()
    scale_factor = (means - center).abs().quantile(0.95, dim=0).max().reshape(1).contiguous()
    out = torch.empty((n, PLY_VERTEX_FLOATS), device=means.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    inp = PlyInputs(p(extrinsics), p(means), p(scales), p(rotations), p(harmonics), p(opacities), p(center),
                    p(scale_factor))
    stream = C.c_void_p(torch.cuda.current_stream(means.device).cuda_stream)
    _lib.check(lib.lsr_ply_pack(n, harmonics.shape[2], C.byref(inp), p(out), stream), "lsr_ply_pack")
    return out


def export_ply(extrinsics: Tensor, means: Tensor, scales: Tensor, rotations: Tensor, harmonics: Tensor,
               opacities: Tensor, path: Path) -> None:
    vertices = pack_vertices(extrinsics, means, scales, rotations, harmonics, opacities).cpu().contiguous()
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    _lib.check(_lib.load().lsr_ply_write_host(os.fsencode(str(path)), C.c_void_p(vertices.data_ptr()),
                                              vertices.shape[0]), "lsr_ply_write_host")
