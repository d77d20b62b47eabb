"""latentsplat_amd — MI355X-native Gaussian-splat rasterizer path of latentSplat.

Only what the hot path needs: ``csrc/`` (HIP kernels + C ABI), ``rasterizer`` (drop-in
``GaussianRasterizer`` API + batched multi-view op), ``decoder`` (mirror of the reference's
``src/model/decoder`` surface) and ``synthetic`` (seeded scenes for tests / bench).
"""
__version__ = "0.1.0"
