"""Oracle: GPflow stationary-kernel arithmetic.

GPflow is a third-party dependency of the reference (requirements.txt:1,
`gpflow>=2.5.2`, not vendored); the formulas below restate its published
algorithm.  Call sites in the reference that this pins:
`cggp/models.py:112,141-143,236,255-257,300,333-335`, `cggp/optimize.py:50`.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""

import numpy as np

KERNEL_NAMES = ("se", "matern12", "matern32", "matern52")


def square_distance(X, X2=None):
    """gpflow.utilities.ops.square_distance: ||a||^2 + ||b||^2 - 2 a.b (K1).

    No clamp at zero (GPflow does not clamp here).  Called explicitly at
    `cggp/optimize.py:50` and implicitly by every stationary kernel call.
    """
    X = np.asarray(X)
    if X2 is None:
        Xs = np.sum(np.square(X), axis=-1, keepdims=True)
        dist = -2.0 * (X @ X.T)
        dist = dist + Xs + Xs.T
        return dist
    X2 = np.asarray(X2)
    Xs = np.sum(np.square(X), axis=-1)
    X2s = np.sum(np.square(X2), axis=-1)
    dist = -2.0 * (X @ X2.T)
    dist = dist + Xs[:, None] + X2s[None, :]
    return dist


class Kernel:
    """Minimal stand-in for gpflow.kernels.IsotropicStationary (K2).

    `name` in KERNEL_NAMES; `variance` scalar; `lengthscales` scalar or [D] (ARD).
    """

    def __init__(self, name="se", variance=1.0, lengthscales=1.0, dtype=np.float64):
        assert name in KERNEL_NAMES, name
        self.name = name
        self.dtype = np.dtype(dtype)
        self.variance = self.dtype.type(variance)
        self.lengthscales = np.asarray(lengthscales, dtype=self.dtype)

    def scale(self, X):
        return np.asarray(X, dtype=self.dtype) / self.lengthscales

    def scaled_squared_euclid_dist(self, X, X2=None):
        return square_distance(self.scale(X), None if X2 is None else self.scale(X2))

    def K_r2(self, r2):
        t = self.dtype.type
        if self.name == "se":
            return self.variance * np.exp(t(-0.5) * r2)
        # GPflow: r = sqrt(maximum(r2, 1e-36)) for kernels defined through K_r
        r = np.sqrt(np.maximum(r2, t(1e-36)))
        if self.name == "matern12":
            return self.variance * np.exp(-r)
        if self.name == "matern32":
            s3 = t(np.sqrt(3.0))
            return self.variance * (t(1.0) + s3 * r) * np.exp(-s3 * r)
        s5 = t(np.sqrt(5.0))
        return self.variance * (t(1.0) + s5 * r + t(5.0 / 3.0) * np.square(r)) * np.exp(-s5 * r)

    def K(self, X, X2=None):
        return self.K_r2(self.scaled_squared_euclid_dist(X, X2))

    def K_diag(self, X):
        return np.full(np.asarray(X).shape[:-1], self.variance, dtype=self.dtype)

    def __call__(self, X, X2=None, full_cov=True):
        if not full_cov:
            assert X2 is None
            return self.K_diag(X)
        return self.K(X, X2)


def Kuu(Z, kernel, jitter=0.0):
    """gpflow.covariances.Kuu for InducingPoints: k(Z,Z) + jitter*I  -> [M,M] (K3)."""
    Kzz = kernel.K(Z)
    Kzz = Kzz + kernel.dtype.type(jitter) * np.eye(Kzz.shape[0], dtype=kernel.dtype)
    return Kzz


def Kuf(Z, kernel, Xnew):
    """gpflow.covariances.Kuf for InducingPoints: k(Z, Xnew) -> [M,N] (K3; note orientation)."""
    return kernel.K(Z, Xnew)


def k_direct(kernel, X, X2):
    """Independent second restatement: direct differences (no expansion).

    Used only to cross-check `Kernel.K` (two restatements must agree).
    """
    a = kernel.scale(X)[:, None, :]
    b = kernel.scale(X2)[None, :, :]
    r2 = np.sum(np.square(a - b), axis=-1)
    return kernel.K_r2(r2)
