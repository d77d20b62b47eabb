"""ctypes binding of ``liblsr_hip.so`` (C ABI: include/lsr_rasterizer.h, include/lsr_adapter.h, include/lsr_latent.h, include/lsr_ply.h).

The library is built in-tree by ``latentsplat_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no CPU fallback: if the shared object is missing or not loadable this module raises, and
every product entry point above it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# LSR_LIB selects an alternative build of the same library (kernel A/B experiments only)
_SO = os.environ.get("LSR_LIB") or os.path.join(_CSRC, "liblsr_hip.so")

VIEW_FLOATS = 44
COLOR_NONE, COLOR_SH, COLOR_PRECOMP = 0, 1, 2
FEAT_DIRECT, FEAT_SH = 0, 1
SH_AXES_3DGS, SH_AXES_REFERENCE = 0, 1   # lsr_dims.color_sh_convention
FWD_FOR_BACKWARD = 1                      # lsr_dims.forward_flags
FWD_CLEARS_GRAD = 2
FWD_REACHED_ONLY = 4
FWD_FRONT_DONE = 8      # (ABI v10) lsr_forward_front has launched the front half of this forward
MAX_SH_GROUP_FLOATS = 120   # C*Kf the fused latent-SH path supports (LDS budget of sh.hip)
MAX_FEAT_CHANNELS = 32


class LsrError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [("num_views", C.c_int32), ("num_gaussians", C.c_int32), ("height", C.c_int32),
                ("width", C.c_int32), ("feat_channels", C.c_int32), ("color_mode", C.c_int32),
                ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32), ("vs_means", C.c_int64),
                ("vs_cov", C.c_int64), ("vs_opac", C.c_int64), ("vs_color", C.c_int64),
                ("vs_feat", C.c_int64), ("cov_elems", C.c_int32), ("feat_mode", C.c_int32),
                ("feat_sh_degree", C.c_int32), ("feat_sh_coeffs", C.c_int32),
                ("color_sh_channel_major", C.c_int32), ("views_per_group", C.c_int32),
                ("color_sh_convention", C.c_int32), ("forward_flags", C.c_int32), ("seg_cap_hint", C.c_int32)]


class Inputs(C.Structure):
    _fields_ = [("views", C.c_void_p), ("means3D", C.c_void_p), ("cov3D", C.c_void_p),
                ("opacities", C.c_void_p), ("color", C.c_void_p), ("features", C.c_void_p)]


class Outputs(C.Structure):
    _fields_ = [("color", C.c_void_p), ("feature", C.c_void_p), ("mask", C.c_void_p),
                ("depth", C.c_void_p), ("radii", C.c_void_p), ("grad_ws", C.c_void_p)]


class OutGrads(C.Structure):
    _fields_ = [("color", C.c_void_p), ("feature", C.c_void_p), ("mask", C.c_void_p),
                ("depth", C.c_void_p)]


class InGrads(C.Structure):
    _fields_ = [("means3D", C.c_void_p), ("cov3D", C.c_void_p), ("opacities", C.c_void_p),
                ("color", C.c_void_p), ("features", C.c_void_p), ("means2D", C.c_void_p)]


class Layout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in
                ("geom_rec", "geom_rec_floats", "geom_bin", "geom_tile_count", "geom_tile_start",
                 "geom_header", "bin_keys", "bin_point_list", "img_final_T", "img_n_contrib", "geom_bin_stride",
                 "bin_half_list", "geom_half_count", "geom_item_flags")]


class AdapterDims(C.Structure):      # lsr_adapter_dims (include/lsr_adapter.h, include/lsr_latent.h, include/lsr_ply.h)
    _fields_ = [("num_cameras", C.c_int32), ("rays", C.c_int32), ("samples", C.c_int32),
                ("height", C.c_int32), ("width", C.c_int32), ("cov_elems", C.c_int32),
                ("scale_min", C.c_float), ("scale_max", C.c_float), ("eps", C.c_float),
                ("raw_stride", C.c_int32), ("reserved0", C.c_int32), ("reserved1", C.c_int32)]


class AdapterInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("extrinsics", "intrinsics", "coordinates", "depths", "raw")]


class AdapterOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("means", "covariances", "scales", "rotations")]


class AdapterOutGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("means", "covariances", "scales", "rotations")]


class AdapterInGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("coordinates", "depths", "raw")]


class LatentDims(C.Structure):       # lsr_latent_dims (include/lsr_latent.h, include/lsr_ply.h)
    _fields_ = [("num_views", C.c_int32), ("channels", C.c_int32), ("height", C.c_int32),
                ("width", C.c_int32), ("out_height", C.c_int32), ("out_width", C.c_int32),
                ("logvar_mode", C.c_int32), ("color_channels", C.c_int32),
                ("logvar_min", C.c_float), ("logvar_max", C.c_float),
                ("reserved0", C.c_int32), ("reserved1", C.c_int32)]


class LatentInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("features", "mask", "noise", "color")]


class LatentOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("skip", "z", "logvar")]


class LatentOutGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("skip", "z")]


class PlyInputs(C.Structure):        # lsr_ply_inputs (include/lsr_ply.h)
    _fields_ = [(n, C.c_void_p) for n in ("extrinsics", "means", "scales", "rotations", "harmonics",
                                          "opacities", "center", "scale_factor")]


PLY_VERTEX_FLOATS = 17

EXPORTS = (
    "lsr_abi_version", "lsr_error_string", "lsr_last_hip_error", "lsr_geom_workspace_bytes",
    "lsr_image_workspace_bytes", "lsr_binning_workspace_bytes", "lsr_grad_workspace_bytes",
    "lsr_get_layout", "lsr_build_views", "lsr_pack_view", "lsr_forward_prepare", "lsr_forward_render", "lsr_forward_nosync", "lsr_forward_speculative", "lsr_forward_front",
    "lsr_forward_status", "lsr_forward_abandon", "lsr_backward",
    "lsr_profile_enable", "lsr_profile_num_stages", "lsr_profile_stage_name", "lsr_profile_read",
    "lsr_debug_set_knob", "lsr_set_projection_contraction", "lsr_get_projection_contraction",
    "lsr_adapter_forward", "lsr_adapter_backward", "lsr_latent_forward", "lsr_latent_backward",
    "lsr_ply_pack", "lsr_ply_write_host",
)

_lib = None
ABI_VERSION = 10    # include/lsr_rasterizer.h LSR_ABI_VERSION


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return _SO


def so_path() -> str:
    return _SO


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; it has to be the HIP runtime of the process, so make sure it
    # is loaded before our library pulls in the system copy (two runtimes = "no device" at launch)
    import torch  # noqa: F401
    if not os.path.exists(_SO):
        raise LsrError(
            f"{_SO} not found: the MI355X rasterizer extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU "
            "fallback for the product path.")
    try:
        lib = C.CDLL(_SO)
    except OSError as e:  # e.g. libamdhip64 missing on a machine without ROCm
        raise LsrError(f"cannot load {_SO}: {e}") from e
    P, I32, I64, SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    lib.lsr_abi_version.restype = C.c_int
    # checked before any other prototype is bound: the structs below are this version's (an older or newer build would
    # be handed structs of the wrong size; there is no compatibility mode)
    abi = lib.lsr_abi_version()
    if abi != ABI_VERSION:
        raise LsrError(f"{_SO}: ABI version {abi}, this package needs {ABI_VERSION}; rebuild it "
                       "(`python -c 'import __graft_entry__ as g; g.build()'`)")
    lib.lsr_error_string.restype = C.c_char_p
    lib.lsr_error_string.argtypes = [C.c_int]
    lib.lsr_last_hip_error.restype = C.c_int
    for name in ("lsr_geom_workspace_bytes", "lsr_image_workspace_bytes", "lsr_grad_workspace_bytes"):
        getattr(lib, name).restype = SZ
        getattr(lib, name).argtypes = [C.POINTER(Dims)]
    lib.lsr_binning_workspace_bytes.restype = SZ
    lib.lsr_binning_workspace_bytes.argtypes = [C.POINTER(Dims), I64, I32]
    lib.lsr_get_layout.restype = C.c_int
    lib.lsr_get_layout.argtypes = [C.POINTER(Dims), I64, C.POINTER(Layout)]
    lib.lsr_build_views.restype = C.c_int
    lib.lsr_build_views.argtypes = [I32, P, P, P, P, P, I32, I32, P, P]
    lib.lsr_pack_view.restype = C.c_int
    lib.lsr_pack_view.argtypes = [P, P, P, P, C.c_float, C.c_float, P, P, P, P]
    lib.lsr_forward_prepare.restype = C.c_int
    lib.lsr_forward_prepare.argtypes = [C.POINTER(Dims), C.POINTER(Inputs), P, P, C.POINTER(I64),
                                        C.POINTER(I32), P]
    lib.lsr_forward_render.restype = C.c_int
    lib.lsr_forward_render.argtypes = [C.POINTER(Dims), C.POINTER(Inputs), P, P, P, I64, I32,
                                       C.POINTER(Outputs), P]
    lib.lsr_forward_nosync.restype = C.c_int
    lib.lsr_forward_nosync.argtypes = [C.POINTER(Dims), C.POINTER(Inputs), P, P, P, I64, I32, C.POINTER(Outputs), P]
    lib.lsr_forward_speculative.restype = C.c_int
    lib.lsr_forward_speculative.argtypes = [C.POINTER(Dims), C.POINTER(Inputs), P, P, P, I64, I32, C.POINTER(Outputs),
                                            C.POINTER(I64), C.POINTER(I32), C.POINTER(I32), P]
    lib.lsr_forward_front.restype = C.c_int
    lib.lsr_forward_front.argtypes = [C.POINTER(Dims), C.POINTER(Inputs), P, P, I64, P]
    lib.lsr_forward_abandon.restype = C.c_int
    lib.lsr_forward_abandon.argtypes = [P]
    lib.lsr_forward_status.restype = C.c_int
    lib.lsr_forward_status.argtypes = [C.POINTER(Dims), P, C.POINTER(I64), C.POINTER(I32), C.POINTER(I32), P]
    lib.lsr_backward.restype = C.c_int
    lib.lsr_backward.argtypes = [C.POINTER(Dims), C.POINTER(Inputs), P, P, P, I64, P, C.POINTER(Outputs),
                                 C.POINTER(OutGrads), P, C.POINTER(InGrads), P]
    lib.lsr_profile_enable.argtypes = [C.c_int]
    lib.lsr_profile_stage_name.restype = C.c_char_p
    lib.lsr_profile_stage_name.argtypes = [C.c_int]
    lib.lsr_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(I64)]
    if hasattr(lib, "lsr_debug_set_knob"):    # (absent from a previous round's build selected with LSR_LIB for an A/B)
        lib.lsr_debug_set_knob.restype = C.c_int
        lib.lsr_debug_set_knob.argtypes = [C.c_char_p, C.c_int]
        lib.lsr_set_projection_contraction.restype = C.c_int
        lib.lsr_set_projection_contraction.argtypes = [C.c_int]
        lib.lsr_get_projection_contraction.restype = C.c_int
    lib.lsr_adapter_forward.restype = C.c_int
    lib.lsr_adapter_forward.argtypes = [C.POINTER(AdapterDims), C.POINTER(AdapterInputs),
                                        C.POINTER(AdapterOutputs), P]
    lib.lsr_adapter_backward.restype = C.c_int
    lib.lsr_adapter_backward.argtypes = [C.POINTER(AdapterDims), C.POINTER(AdapterInputs),
                                         C.POINTER(AdapterOutGrads), C.POINTER(AdapterInGrads), P]
    lib.lsr_latent_forward.restype = C.c_int
    lib.lsr_latent_forward.argtypes = [C.POINTER(LatentDims), C.POINTER(LatentInputs),
                                       C.POINTER(LatentOutputs), P]
    lib.lsr_latent_backward.restype = C.c_int
    lib.lsr_latent_backward.argtypes = [C.POINTER(LatentDims), C.POINTER(LatentInputs),
                                        C.POINTER(LatentOutGrads), P, P]
    lib.lsr_ply_pack.restype = C.c_int
    lib.lsr_ply_pack.argtypes = [I64, I32, C.POINTER(PlyInputs), P, P]
    lib.lsr_ply_write_host.restype = C.c_int
    lib.lsr_ply_write_host.argtypes = [C.c_char_p, P, I64]
    _lib = lib
    return lib


def set_knob(name: str, value: int) -> None:
    """Override one LSR_* development knob of the library for the rest of the process (A/B experiments only)."""
    check(load().lsr_debug_set_knob(name.encode(), int(value)), "lsr_debug_set_knob")


def profile_enable(on, only: tuple = ()) -> None:
    """on: all stages; only=("render_forward", ...): just those stages (two event packets each)."""
    lib = load()
    if only:
        names = [lib.lsr_profile_stage_name(i).decode() for i in range(lib.lsr_profile_num_stages())]
        mask = 0
        for n in only:
            mask |= 1 << (names.index(n) + 1)
        lib.lsr_profile_enable(mask)
    else:
        lib.lsr_profile_enable(1 if on else 0)


def profile_read() -> dict:
    """{stage: (total_ms, launches)} since the previous read (blocks until the events finish)."""
    lib = load()
    n = lib.lsr_profile_num_stages()
    ms, cnt = (C.c_double * n)(), (C.c_int64 * n)()
    check(lib.lsr_profile_read(ms, cnt), "lsr_profile_read")
    return {lib.lsr_profile_stage_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}


def check(rc: int, what: str):
    if rc != 0:
        lib = load()
        msg = lib.lsr_error_string(rc).decode()
        raise LsrError(f"{what} failed: {msg} (code {rc}, hipError {lib.lsr_last_hip_error()})")
