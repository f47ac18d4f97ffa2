"""Oracle: nearest-centre assignment + cluster statistics.

Restates `cggp/optimize.py:41-98` and `cggp/selection.py:14-32`: assign every
input to its nearest centre, then pseudo_u = per-cluster mean of y and
counts = per-cluster size (-> Lambda = s2 / counts, `cggp/models.py:226-228`).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""

import numpy as np

from .kernels import square_distance


def nearest_centre_sqdist(Z, X, chunk=8192):
    """argmin_m square_distance(Z, X)[m, i] (`optimize.py:50-51`), first index on ties."""
    X = np.asarray(X)
    idx = np.empty(X.shape[0], dtype=np.int64)
    for s in range(0, X.shape[0], chunk):
        d = square_distance(Z, X[s:s + chunk])
        idx[s:s + chunk] = np.argmin(d, axis=0)
    return idx


def nearest_centre(Z, X, distance_fn, chunk=2048):
    """`selection.py:14-32` (`kmeans_indices_and_distances`) for an arbitrary distance fn."""
    X = np.asarray(X)
    idx = np.empty(X.shape[0], dtype=np.int64)
    dist = np.empty(X.shape[0], dtype=X.dtype)
    for s in range(0, X.shape[0], chunk):
        xs = X[s:s + chunk]
        d = distance_fn((np.asarray(Z)[None, :, :], xs[:, None, :]))  # [chunk, M]
        j = np.argmin(d, axis=-1)
        idx[s:s + chunk] = j
        dist[s:s + chunk] = distance_fn((np.asarray(Z)[j], xs))
    return idx, dist


def cluster_stats(idx, y, M, empty="one"):
    """Per-cluster mean of y and count.

    empty="one": `oips_update_inducing_parameters` (`optimize.py:53-72`): count 0 -> 1,
    mean of an empty cluster is NaN in the reference (reduce_mean of nothing); the
    synthetic inputs of SURVEY §8d make Z rows of X, so no cluster is empty.
    empty="nan": `kmeans_update_inducing_parameters` (`optimize.py:88-96`):
    u = scatter_add(y) / counts (0/0 -> NaN), counts stay 0.
    """
    y = np.asarray(y).reshape(-1)
    counts = np.bincount(idx, minlength=M).astype(y.dtype)
    sums = np.bincount(idx, weights=y, minlength=M).astype(y.dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        means = sums / counts
    if empty == "one":
        counts = np.where(counts == 0, np.ones_like(counts), counts)
    return means.reshape(M, 1), counts.reshape(M, 1)
