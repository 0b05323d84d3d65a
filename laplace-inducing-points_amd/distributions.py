"""Minimal stand-in for ``tfp.distributions.MultivariateNormalFullCovariance`` — the object
``posterior_lla_dense`` / ``predict_lla_dense`` return in the reference
(``src/lla.py:42-45,79-82``).  Only the members its callers use exist:
``.mean() .covariance() .stddev() .sample(seed=, sample_shape=)``
(``tests/test_lla.py:21,24``, ``tests/test_sample.py:478-479``).
Supports a batch of distributions: loc (..., k), covariance (..., k, k).
"""
from __future__ import annotations

import torch


class MultivariateNormalFullCovariance:
    def __init__(self, loc: torch.Tensor, covariance_matrix: torch.Tensor):
        self.loc = loc
        self.covariance_matrix = covariance_matrix

    def mean(self) -> torch.Tensor:
        return self.loc

    def covariance(self) -> torch.Tensor:
        return self.covariance_matrix

    def variance(self) -> torch.Tensor:
        return torch.diagonal(self.covariance_matrix, dim1=-2, dim2=-1)

    def stddev(self) -> torch.Tensor:
        return torch.sqrt(self.variance())

    def sample(self, sample_shape=(), seed=None) -> torch.Tensor:
        if isinstance(sample_shape, int):
            sample_shape = (sample_shape,)
        g = None
        if seed is not None:
            g = torch.Generator(device=self.loc.device).manual_seed(int(seed))
        cov = self.covariance_matrix
        L = torch.linalg.cholesky(0.5 * (cov + cov.transpose(-1, -2)))
        eps = torch.randn(tuple(sample_shape) + tuple(self.loc.shape), dtype=self.loc.dtype,
                          device=self.loc.device, generator=g)
        return self.loc + (L @ eps.unsqueeze(-1)).squeeze(-1)
