"""Drop-in module name for the reference's external dependency.

The reference imports ``GaussianRasterizationSettings`` and ``GaussianRasterizer`` from a module
literally called ``diff_gaussian_rasterization`` (/root/reference/src/model/decoder/
cuda_splatting.py:6-9).  Putting this repository's root on ``sys.path`` (or installing it) makes
that import resolve to the MI355X-native implementation without touching the reference.
"""
from latentsplat_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    make_view_table,
    rasterize_views,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_views", "make_view_table"]
