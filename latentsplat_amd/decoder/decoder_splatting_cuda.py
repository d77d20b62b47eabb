"""``DecoderSplattingCUDA`` on the MI355X rasterizer.

Same constructor, methods and outputs as the reference class
(/root/reference/src/model/decoder/decoder_splatting_cuda.py:15-119); the config name stays
``"splatting_cuda"`` so existing experiment YAMLs select it unchanged.  Unlike the reference,
``forward`` does not replicate the Gaussians once per target view (:71-87): it hands scene-major
tensors to ``render_scenes`` and the kernels read each scene once for all of its views.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional

import torch
from torch import Tensor

from .cuda_splatting import DepthRenderingMode, RenderOutput, render_depth_scenes, render_scenes
from .decoder import Decoder, DecoderOutput
from .types import DiagonalGaussianDistribution, Gaussians


@dataclass
class DecoderSplattingCUDACfg:
    name: Literal["splatting_cuda"]


class DecoderSplattingCUDA(Decoder[DecoderSplattingCUDACfg]):
    background_color: Tensor

    def __init__(self, cfg: DecoderSplattingCUDACfg, background_color: list[float] = [0.0, 0.0, 0.0],
                 variational: bool = False) -> None:
        super().__init__(cfg)
        self.register_buffer("background_color", torch.tensor(background_color, dtype=torch.float32),
                             persistent=False)
        self.variational = variational

    def render_to_decoder_output(self, render_output: RenderOutput, b: int, v: int) -> DecoderOutput:
        def split(t: Optional[Tensor]) -> Optional[Tensor]:
            return None if t is None else t.unflatten(0, (b, v))

        mask = split(render_output.mask)
        posterior = None
        if render_output.feature is not None:
            feats = split(render_output.feature)
            if self.variational:
                mean, logvar = feats.chunk(2, dim=2)
            else:
                # background feature = 0 = mean = logvar: empty pixels get unit variance,
                # opaque pixels a vanishing one
                mean = feats
                logvar = (1 - mask.detach()[:, :, None]).log().expand_as(feats)
            posterior = DiagonalGaussianDistribution(mean, logvar)
        return DecoderOutput(color=split(render_output.color), feature_posterior=posterior, mask=mask,
                             depth=split(render_output.depth))

    def forward(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor,
                far: Tensor, image_shape: tuple[int, int],
                depth_mode: Optional[DepthRenderingMode] = None, return_colors: bool = True,
                return_features: bool = True) -> DecoderOutput:
        b, v = extrinsics.shape[:2]
        rendered = render_scenes(
            extrinsics, intrinsics, near, far, image_shape, self.background_color,
            gaussians.means, gaussians.covariances, gaussians.opacities,
            gaussians.color_harmonics if return_colors else None,
            gaussians.feature_harmonics if return_features else None)
        out = self.render_to_decoder_output(rendered, b, v)
        if depth_mode is not None and depth_mode != "depth":
            out.depth = self.render_depth(gaussians, extrinsics, intrinsics, near, far, image_shape, depth_mode)
        return out

    def render_depth(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor,
                     far: Tensor, image_shape: tuple[int, int],
                     mode: DepthRenderingMode = "depth") -> Tensor:
        # scene-major: the Gaussians are not replicated per view (the reference repeats them v-fold, :105-110)
        return render_depth_scenes(extrinsics, intrinsics, near, far, image_shape, gaussians.means,
                                   gaussians.covariances, gaussians.opacities, mode=mode)

    def last_layer_weights(self) -> None:
        return None
