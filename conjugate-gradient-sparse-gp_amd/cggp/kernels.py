"""Stationary kernels, inducing points and covariance helpers (rows K1-K3 of SURVEY §8a).

Host-side mirror of the GPflow objects the reference's hot path touches
(`gpflow.kernels.SquaredExponential/Matern12/Matern32/Matern52`, `InducingPoints`,
`gpflow.covariances.Kuu/Kuf`; call sites `cggp/models.py:300,333-335`,
`cggp/cli_utils.py:363-368,455-473`).  The objects only hold hyper-parameters on the host; the
arithmetic is libmgp's (`ops.k_dense` and the fused sweeps).
"""

import numpy as np
import torch

from . import ops


def _as_float_list(x):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return [float(v) for v in np.atleast_1d(np.asarray(x, dtype=np.float64))]


class Stationary:
    """Isotropic-stationary kernel with ARD lengthscales; `name` selects the profile."""

    name = None

    def __init__(self, variance=1.0, lengthscales=1.0):
        self.variance = float(variance if not isinstance(variance, torch.Tensor) else variance.item())
        self.lengthscales = _as_float_list(lengthscales)
        if self.variance <= 0 or any(l <= 0 for l in self.lengthscales):
            raise ValueError("variance and lengthscales must be positive")

    def spec(self, D):
        return ops.KernelSpec(self.name, self.variance, self.lengthscales, D)

    def K(self, X, X2=None):
        X2 = X if X2 is None else X2
        return ops.k_dense(self.spec(X.shape[-1]), X, X2)

    def K_diag(self, X):
        return torch.full(tuple(X.shape[:-1]), self.variance, dtype=X.dtype, device=X.device)

    def __call__(self, X, X2=None, *, full_cov=True):
        if not full_cov:
            if X2 is not None:
                raise ValueError("full_cov=False takes a single input")
            return self.K_diag(X)
        return self.K(X, X2)


class SquaredExponential(Stationary):
    name = "se"


class Matern12(Stationary):
    name = "matern12"


class Matern32(Stationary):
    name = "matern32"


class Matern52(Stationary):
    name = "matern52"


RBF = SquaredExponential

_BY_NAME = {"se": SquaredExponential, "matern12": Matern12, "matern32": Matern32, "matern52": Matern52}


def name_to_kernel(name, dim=1):
    """`cggp/cli_utils.py:465-473` (lengthscales 0.1 per dimension)."""
    if name not in _BY_NAME:
        raise NotImplementedError(f"Unknown kernel name {name}")
    return _BY_NAME[name](lengthscales=[0.1] * dim)


def kernel_to_name(kernel):
    """`cggp/cli_utils.py:455-462`."""
    if not isinstance(kernel, Stationary) or kernel.name is None:
        raise NotImplementedError(f"Unknown kernel {kernel}")
    return kernel.name


class InducingPoints:
    """`gpflow.inducing_variables.InducingPoints`: holds Z [M, D] on the device."""

    def __init__(self, Z):
        if isinstance(Z, InducingPoints):
            Z = Z.Z
        if not isinstance(Z, torch.Tensor):
            raise TypeError("Z must be a torch.Tensor on the GPU")
        self.Z = Z

    @property
    def num_inducing(self):
        return self.Z.shape[0]


def inducingpoint_wrapper(iv):
    return iv if isinstance(iv, InducingPoints) else InducingPoints(iv)


def Kuu(inducing_variable, kernel, *, jitter=0.0, diag_add=None):
    """k(Z,Z) + jitter I (+ diag(diag_add): `add_diagonal` fused, `cggp/utils.py:11-17`) -> [M,M]."""
    Z = inducingpoint_wrapper(inducing_variable).Z
    return ops.k_dense(kernel.spec(Z.shape[1]), Z, Z, jitter=float(jitter), diag_add=diag_add)


def Kuf(inducing_variable, kernel, Xnew):
    """k(Z, Xnew) -> [M, N] (note the orientation, as gpflow.covariances.Kuf)."""
    Z = inducingpoint_wrapper(inducing_variable).Z
    return ops.k_dense(kernel.spec(Z.shape[1]), Z, Xnew)
