"""Oracle: pairwise distance functions.

Restates `cggp/distance.py:9-34`.  Functions take a tuple `(x, y)` with
broadcasting over leading axes, exactly as the reference's do when they are
vmapped in `cggp/selection.py:24-31`.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""

import numpy as np


def euclid_distance(args):
    """`cggp/distance.py:9-11`: ||x - y||_2 over the last axis."""
    x, y = args
    return np.linalg.norm(np.asarray(x) - np.asarray(y), axis=-1)


def _k_pair(kernel, x, y):
    """k(x_i, y_i) elementwise over broadcast leading axes (stationary kernel)."""
    a = kernel.scale(x)
    b = kernel.scale(y)
    # GPflow's kernel(x, y) on matching rows goes through the expansion form
    r2 = np.sum(a * a, -1) + np.sum(b * b, -1) - 2.0 * np.sum(a * b, -1)
    return kernel.K_r2(r2)


def create_distance_fn(kernel, distance_type):
    """`cggp/distance.py:14-34`."""

    def cov(args):  # :15-22
        x, y = args
        x_dist = kernel.variance  # kernel(x, full_cov=False) == K_diag == variance
        y_dist = kernel.variance
        return x_dist + y_dist - 2 * _k_pair(kernel, x, y)

    def cor(args):  # :24-30
        x, y = args
        x_dist = kernel.variance
        y_dist = kernel.variance
        return 1.0 - _k_pair(kernel, x, y) / np.sqrt(x_dist * y_dist)

    functions = {"covariance": cov, "correlation": cor, "euclidean": euclid_distance}
    return functions[distance_type]
