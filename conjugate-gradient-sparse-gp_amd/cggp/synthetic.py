"""Synthetic inputs of SURVEY.md §8(d): identical arrays feed the CPU and GPU paths.

numpy only (host side); shapes follow BASELINE.json `configs`.  The reference
feeds z-scored data (`cggp/data.py:101-110`), picks Z as rows of X
(`cggp/cli_utils.py:156-158`) and uses sigma^2 = 0.1 (`cli_utils.py:153`).
"""

from typing import NamedTuple
import numpy as np

CONFIGS = {
    # name: (N, D, M, dtype, kernel)
    "C1": (2048, 1, 128, "float64", "se"),
    "C2": (100_000, 8, 2048, "float64", "se"),
    "C3": (1 << 20, 8, 4096, "float64", "se"),
    "C3r": (1_000_000, 8, 4096, "float64", "se"),  # ragged N
    "C4": (10_000_000, 2, 8192, "float32", "se"),
    "C5": (1 << 20, 32, 4096, "float64", "matern32"),
}


class Synthetic(NamedTuple):
    X: np.ndarray
    y: np.ndarray
    Z: np.ndarray
    noise_variance: float
    variance: float
    lengthscales: np.ndarray


def make_inputs(N, D, M, dtype="float64", need_y=True):
    X = np.random.default_rng(0).standard_normal((N, D))
    Z = X[np.random.default_rng(1).choice(N, M, replace=False)]
    if need_y:
        eps = np.random.default_rng(2).standard_normal((N, 1))
        y = np.sum(np.sin(X), axis=1, keepdims=True) / np.sqrt(D) + np.sqrt(0.1) * eps
    else:
        y = np.zeros((0, 1))
    dt = np.dtype(dtype)
    return Synthetic(X.astype(dt), y.astype(dt), Z.astype(dt), 0.1, 1.0, np.ones(D, dt))


def make_vectors(M, R, dtype="float64"):
    return np.random.default_rng(3).standard_normal((M, R)).astype(dtype)


def make_probes(M, P=64, dtype="float64"):
    """Rademacher probes from a documented stream (numpy PCG64, seed 4)."""
    return (2 * np.random.default_rng(4).integers(0, 2, size=(M, P)) - 1).astype(dtype)
