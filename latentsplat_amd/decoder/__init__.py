"""Decoder surface (reference: /root/reference/src/model/decoder/__init__.py:1-17)."""
from .cuda_splatting import (DepthRenderingMode, RenderOutput, get_projection_matrix, render_cuda,
                             render_cuda_orthographic, render_depth_cuda, render_scenes)
from .decoder import Decoder, DecoderOutput
from .decoder_splatting_cuda import DecoderSplattingCUDA, DecoderSplattingCUDACfg
from .types import DiagonalGaussianDistribution, Gaussians, VariationalGaussians

DECODERS = {"splatting_cuda": DecoderSplattingCUDA}

DecoderCfg = DecoderSplattingCUDACfg


def get_decoder(decoder_cfg: DecoderCfg, background_color: list[float], variational: bool = False) -> Decoder:
    return DECODERS[decoder_cfg.name](decoder_cfg, background_color, variational)
