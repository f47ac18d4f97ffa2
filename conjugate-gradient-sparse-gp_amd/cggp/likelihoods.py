"""Gaussian likelihood (GPflow `gpflow.likelihoods.Gaussian`), used by `LpSVGP.elbo`
(`cggp/models.py:132`) and the metrics callback (`cggp/optimize.py:306`).

Elementwise glue on [B,1] tensors -- off the kernels of the hot path; plain torch.
"""

import math

import torch

LOG2PI = math.log(2.0 * math.pi)


class Gaussian:
    def __init__(self, variance=1.0):
        self.variance = float(variance if not isinstance(variance, torch.Tensor) else variance.item())
        if self.variance <= 0:
            raise ValueError("likelihood variance must be positive")

    def variational_expectations(self, X, Fmu, Fvar, Y):
        v = self.variance
        ve = -0.5 * LOG2PI - 0.5 * math.log(v) - 0.5 * ((Y - Fmu) ** 2 + Fvar) / v
        return ve.sum(dim=-1)

    def predict_log_density(self, X, Fmu, Fvar, Y):
        v = Fvar + self.variance
        ld = -0.5 * (LOG2PI + torch.log(v) + (Y - Fmu) ** 2 / v)
        return ld.sum(dim=-1)
