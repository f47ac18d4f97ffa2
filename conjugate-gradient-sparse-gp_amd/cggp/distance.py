"""Pairwise distance functions -- host mirror of `cggp/distance.py` (rows D1-D3).

`create_distance_fn(kernel, distance_type)` returns `fn((x, y))` exactly as the reference's
does (`cggp/distance.py:14-34`); x and y broadcast over leading axes.  These are the eager,
elementwise forms used on a handful of points; the N x M nearest-centre search built on the
same distances runs fused in libmgp (`ops.nearest_center`, row F1).
"""

import math

import torch

DistanceTypes = ("euclidean", "covariance", "correlation")


def euclid_distance(args):
    """`cggp/distance.py:9-11`."""
    x, y = args
    return torch.linalg.norm(x - y, dim=-1)


# what the fused N x M searches of libmgp need to know about a distance function
euclid_distance.distance_type = "euclidean"
euclid_distance.kernel = None


def _rho(kernel, x, y):
    """k(x,y)/variance for matching rows, through GPflow's expansion of the scaled distance."""
    ls = torch.as_tensor(kernel.lengthscales, dtype=x.dtype, device=x.device)
    a, b = x / ls, y / ls
    r2 = (a * a).sum(-1) + (b * b).sum(-1) - 2.0 * (a * b).sum(-1)
    if kernel.name == "se":
        return torch.exp(-0.5 * r2)
    r = torch.sqrt(torch.clamp(r2, min=1e-36))
    if kernel.name == "matern12":
        return torch.exp(-r)
    if kernel.name == "matern32":
        s3 = math.sqrt(3.0)
        return (1.0 + s3 * r) * torch.exp(-s3 * r)
    s5 = math.sqrt(5.0)
    return (1.0 + s5 * r + 5.0 / 3.0 * r * r) * torch.exp(-s5 * r)


def create_distance_fn(kernel, distance_type):
    def cov(args):  # :15-22
        x, y = args
        kxy = kernel.variance * _rho(kernel, x, y)
        return kernel.variance + kernel.variance - 2 * kxy

    def cor(args):  # :24-30
        x, y = args
        kxy = kernel.variance * _rho(kernel, x, y)
        return 1.0 - kxy / math.sqrt(kernel.variance * kernel.variance)

    # tagged so that `selection.kmeans_*` / `ops.nearest_center` can run the same distance fused
    cov.distance_type, cov.kernel = "covariance", kernel
    cor.distance_type, cor.kernel = "correlation", kernel
    functions = {"covariance": cov, "correlation": cor, "euclidean": euclid_distance}
    return functions[distance_type]


def resolve_distance(distance_fn, kernel=None):
    """(distance_type, kernel) of a `distance_fn` argument as the reference's selection code takes it
    (`cggp/selection.py:14-18,35-41`): None (euclidean), `euclid_distance`, a function made by
    `create_distance_fn`, or -- build-side shorthand -- the distance type as a string.  The N x M search
    runs fused in libmgp, which evaluates the distance itself, so a foreign callable cannot be honoured
    and is refused loudly rather than silently replaced."""
    if distance_fn is None:
        return "euclidean", kernel
    if isinstance(distance_fn, str):
        if distance_fn not in DistanceTypes:
            raise ValueError(f"unknown distance type {distance_fn!r}")
        return distance_fn, kernel
    dt = getattr(distance_fn, "distance_type", None)
    if callable(distance_fn) and dt in DistanceTypes:
        return dt, (getattr(distance_fn, "kernel", None) or kernel)
    raise TypeError("distance_fn must be None, cggp.distance.euclid_distance or a function returned by "
                    "cggp.distance.create_distance_fn: the nearest-centre search is fused in libmgp and "
                    "cannot call an arbitrary Python distance")
