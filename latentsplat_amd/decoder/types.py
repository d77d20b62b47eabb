"""Input / output containers of the decoder surface.

Counterparts of the reference types (paths relative to /root/reference):
  * ``Gaussians``                      — src/model/types.py:9-15
  * ``VariationalGaussians``           — src/model/types.py:18-32 (what the encoder returns and
    ``model_wrapper.py:362`` turns into ``Gaussians`` with ``sample()`` / ``flatten()``)
  * ``DiagonalGaussianDistribution``   — src/model/diagonal_gaussian_distribution.py:8-95
    (only what the decoder path touches: construction from mean/logvar or params, the logvar
    clamp to (-30, 20), ``sample`` / ``mode`` / ``kl`` / ``nll``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import Tensor


@dataclass
class Gaussians:
    means: Tensor                      # (b, g, 3)
    covariances: Tensor                # (b, g, 3, 3)
    opacities: Tensor                  # (b, g)
    color_harmonics: Optional[Tensor] = None    # (b, g, 3, d_color_sh)
    feature_harmonics: Optional[Tensor] = None  # (b, g, c, d_feature_sh)


class DiagonalGaussianDistribution:
    """Diagonal normal over tensors of any shape; zero variance when no logvar is given."""

    def __init__(self, mean: Optional[Tensor] = None, logvar: Optional[Tensor] = None,
                 params: Optional[Tensor] = None, dim: int = 0,
                 logvar_interval: Tuple[float, float] = (-30.0, 20.0)):
        if params is not None:
            assert mean is None and logvar is None, "If params are given, mean and logvar are not expected"
            mean, logvar = params.chunk(2, dim=dim)
        assert mean is not None, "Either mean or params must be given"
        self.dim = dim
        self.logvar_interval = logvar_interval
        self.mean = mean
        self._params = params
        self._set_logvar(logvar)

    def _set_logvar(self, logvar: Optional[Tensor]) -> None:
        if logvar is None:
            self.logvar, self.std, self.var = None, 0.0, 0.0
            return
        assert logvar.shape == self.mean.shape, "Shapes of mean and logvar must be identical"
        self.logvar = logvar.clamp(*self.logvar_interval)
        self.std = (0.5 * self.logvar).exp()
        self.var = self.logvar.exp()

    @property
    def params(self) -> Tensor:
        if self._params is not None:
            return self._params
        assert self.logvar is not None, "Trying accessing params without params or logvar"
        return torch.cat((self.mean, self.logvar), dim=self.dim)

    @property
    def device(self) -> torch.device:
        return self.mean.device

    def mode(self) -> Tensor:
        return self.mean

    def sample(self) -> Tensor:
        if self.logvar is None:
            return self.mean
        return self.mean + self.std * torch.randn_like(self.mean)

    def kl(self, other: Optional["DiagonalGaussianDistribution"] = None) -> Tensor:
        if self.logvar is None:
            return torch.zeros_like(self.mean)
        if other is None:
            return 0.5 * (self.mean ** 2 + self.var - 1.0 - self.logvar)
        return 0.5 * ((self.mean - other.mean) ** 2 / other.var + self.var / other.var
                      - 1.0 - self.logvar + other.logvar)

    def nll(self, sample: Tensor) -> Tensor:
        if self.logvar is None:
            return torch.zeros_like(self.mean)
        return 0.5 * (math.log(2.0 * math.pi) + self.logvar + (sample - self.mean) ** 2 / self.var)


@dataclass
class VariationalGaussians(Gaussians):
    """Gaussians whose latent-feature harmonics are a distribution; the decoder takes one of the
    three concrete views below."""
    feature_harmonics: Optional[DiagonalGaussianDistribution] = None

    def _to_gaussians(self, feature_harmonics: Tensor) -> Gaussians:
        return Gaussians(self.means, self.covariances, self.opacities, self.color_harmonics, feature_harmonics)

    def flatten(self) -> Gaussians:
        return self._to_gaussians(self.feature_harmonics.params)

    def mode(self) -> Gaussians:
        return self._to_gaussians(self.feature_harmonics.mode())

    def sample(self) -> Gaussians:
        return self._to_gaussians(self.feature_harmonics.sample())
