"""Decoder contract (reference: /root/reference/src/model/decoder/decoder.py:11-53)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Generic, Literal, Optional, TypeVar

from torch import Tensor, nn

from .types import DiagonalGaussianDistribution, Gaussians

DepthRenderingMode = Literal["depth", "log", "disparity", "relative_disparity"]


@dataclass
class DecoderOutput:
    color: Optional[Tensor]                                    # (batch, view, 3, h, w)
    feature_posterior: Optional[DiagonalGaussianDistribution]  # over (batch, view, c, h, w)
    mask: Tensor                                               # (batch, view, h, w)
    depth: Tensor                                              # (batch, view, h, w)


T = TypeVar("T")


class Decoder(nn.Module, ABC, Generic[T]):
    cfg: T

    def __init__(self, cfg: T) -> None:
        super().__init__()
        self.cfg = cfg

    @abstractmethod
    def forward(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor,
                far: Tensor, image_shape: tuple[int, int],
                depth_mode: Optional[DepthRenderingMode] = None, return_colors: bool = True,
                return_features: bool = True) -> DecoderOutput:
        ...

    @property
    @abstractmethod
    def last_layer_weights(self) -> Optional[Tensor]:
        ...
