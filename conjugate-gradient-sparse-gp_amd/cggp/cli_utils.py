"""Model / update-function factories -- host mirror of the callable part of `cggp/cli_utils.py`.

The reference's experiment scripts reach the hot path through these factories
(`create_model_and_update_fn`, `create_update_fn`, `create_predict_fn`,
`batch_posterior_computation`, `cggp/cli_utils.py:143-436`); the click option types, dataset
loaders and path helpers of that module are CLI plumbing and are not rebuilt.  Same names,
argument order and return shapes; tensors are torch device tensors; `use_jit` is accepted and
ignored (there is no tracing compiler here -- the device work is libmgp).  Random draws
(`create_model`'s initial inducing rows) take a `seed`.
"""

import numpy as np
import torch

from . import kernels as _kernels
from . import selection
from .likelihoods import Gaussian
from .models import CGGP, SGPR, cdgp_class, sgpr_class  # noqa: F401  (re-exported as in the reference)
from .optimize import (assign_inducing_parameters, covertree_update_inducing_parameters,
                       kmeans_update_inducing_parameters, oips_update_inducing_parameters)

CLUSTERING_TYPES = ("kmeans", "kmeans2", "covertree", "oips", "uniform", "greedy")


def kernel_fn(dim):
    """`cli_utils.py:363-368`: Matern-3/2, unit variance and lengthscales."""
    return _kernels.Matern32(variance=1.0, lengthscales=[1.0] * dim)


def kernel_to_name(kernel):
    """`cli_utils.py:455-462`."""
    for cls, name in ((_kernels.SquaredExponential, "se"), (_kernels.Matern12, "matern12"),
                      (_kernels.Matern32, "matern32")):
        if isinstance(kernel, cls):
            return name
    raise NotImplementedError(f"Unknown kernel {kernel}")


def name_to_kernel(name, dim=1):
    """`cli_utils.py:465-473`: lengthscales 0.1 per dimension."""
    table = {"se": _kernels.SquaredExponential, "matern12": _kernels.Matern12, "matern32": _kernels.Matern32}
    if name not in table:
        raise NotImplementedError(f"Unknown kernel name {name}")
    return table[name](variance=1.0, lengthscales=[0.1] * dim)


def create_model(model_fn, kernel_fn, data, num_inducing_points=None, seed=0, **model_kwargs):
    """`cli_utils.py:143-168`: `num_inducing_points` (default 10 % of the rows) rows of x drawn
    without replacement as initial inducing inputs, Gaussian likelihood with variance 0.1,
    `model_fn(kernel, likelihood, iv, num_data=n, **model_kwargs)`."""
    x = data[0]
    n, dim = x.shape[0], x.shape[-1]
    m = int(num_inducing_points) if num_inducing_points is not None else int(n * 0.1)
    rand_indices = np.random.default_rng(seed).choice(n, size=m, replace=False)
    iv = x[torch.as_tensor(rand_indices, device=x.device)].clone()
    likelihood = Gaussian(variance=0.1)
    kernel = kernel_fn(dim)
    return model_fn(kernel, likelihood, iv, num_data=n, **model_kwargs)


def _kmeans_like(model, data, distance_type, clustering_fn):
    def update_fn():
        return kmeans_update_inducing_parameters(model, data, distance_type, clustering_fn())
    return update_fn


def create_kmeans_update_fn(model, data, use_jit=True, max_points=1, distance_type="euclidean"):
    """`cli_utils.py:187-207`: Lloyd's iteration started from the model's current Z."""
    x, _ = data

    def clustering_fn():
        iv, _ = selection.kmeans_lloyd(x, max_points, initial_centroids=model.inducing_variable.Z,
                                       distance_type=distance_type, kernel=model.kernel)
        return iv

    return _kmeans_like(model, data, distance_type, clustering_fn)


def create_kmeans2_update_fn(model, data, use_jit=True, max_points=1, distance_type="euclidean", seed=0):
    """`cli_utils.py:210-229`: `scipy.cluster.vq.kmeans2(x, max_points, minit="++")` on the host
    (as in the reference), then the device assignment / statistics."""
    x, _ = data

    def clustering_fn():
        from scipy.cluster.vq import kmeans2
        iv, _ = kmeans2(x.detach().cpu().numpy(), max_points, minit="++", seed=seed)
        return torch.from_numpy(np.ascontiguousarray(iv)).to(device=x.device, dtype=x.dtype)

    return _kmeans_like(model, data, distance_type, clustering_fn)


def create_greedy_update_fn(model, data, use_jit=True, max_points=1, distance_type="euclidean"):
    """`cli_utils.py:232-250`."""
    def update_fn():
        iv, _ = selection.greedy_selection(model.kernel, data[0], max_points)
        return oips_update_inducing_parameters(model, data, iv)
    return update_fn


def create_covertree_update_fn(model, data, use_jit=True, spatial_resolution=1.0, distance_type="euclidean"):
    """`cli_utils.py:253-266`."""
    def update_fn():
        return covertree_update_inducing_parameters(model, data, None, spatial_resolution)
    return update_fn


def create_oips_update_fn(model, data, rho=0.5, use_jit=True, max_points=None, distance_type="euclidean"):
    """`cli_utils.py:269-296`: `max_points` defaults to the dataset size."""
    if max_points is None or max_points <= 0:
        max_points = data[0].shape[0]

    def update_fn():
        iv, _ = selection.oips(model.kernel, data[0], rho, max_points)
        return oips_update_inducing_parameters(model, data, iv)
    return update_fn


def create_uniform_update_fn(model, data, max_points, use_jit=True, distance_type="euclidean", seed=0):
    """`cli_utils.py:299-325`."""
    if max_points > data[0].shape[0]:
        raise ValueError("Max points cannot be larger the dataset size")
    calls = [0]

    def update_fn():
        iv, _ = selection.uniform(data[0], max_points, seed=seed + calls[0])
        calls[0] += 1
        return oips_update_inducing_parameters(model, data, iv)
    return update_fn


def create_update_fn(clustering_type, model, data, use_jit=True, distance_type="euclidean", **clustering_kwargs):
    """`cli_utils.py:328-360`."""
    table = {"kmeans": create_kmeans_update_fn, "kmeans2": create_kmeans2_update_fn,
             "covertree": create_covertree_update_fn, "oips": create_oips_update_fn,
             "uniform": create_uniform_update_fn, "greedy": create_greedy_update_fn}
    if clustering_type not in table:
        raise ValueError(f"Unknown value for {clustering_type}")
    return table[clustering_type](model, data, use_jit=use_jit, distance_type=distance_type, **clustering_kwargs)


def create_model_and_update_fn(model_class, train_data, clustering_type, use_jit=True, distance_type="euclidean",
                               trainable_inducing_points=False, model_kwargs=None, clustering_kwargs=None):
    """`cli_utils.py:371-414`: the model and an `update_fn()` that re-clusters and assigns
    Z / pseudo_u / cluster_counts (Z only for models without cluster statistics)."""
    model_kwargs = {} if model_kwargs is None else model_kwargs
    clustering_kwargs = {} if clustering_kwargs is None else clustering_kwargs
    model = create_model(model_class, kernel_fn, train_data, **model_kwargs)
    internal_update_fn = create_update_fn(clustering_type, model, train_data, use_jit=use_jit,
                                          distance_type=distance_type, **clustering_kwargs)

    def update_fn():
        iv, means, counts = internal_update_fn()
        return assign_inducing_parameters(model, iv, means, counts)

    return model, update_fn


def create_predict_fn(model, use_jit=True):
    """`cli_utils.py:417-423`."""
    def predict_fn(inputs):
        return model.predict_f(inputs)
    return predict_fn


def batch_posterior_computation(predict_fn, data, batch_size):
    """`cli_utils.py:426-436`: predict in row batches, concatenate, return host arrays."""
    x = data[0]
    means, variances = [], []
    for s in range(0, x.shape[0], batch_size):
        mean, variance = predict_fn(x[s:s + batch_size])
        means.append(mean.detach().cpu().numpy())
        variances.append(variance.detach().cpu().numpy())
    return np.concatenate(means, axis=0), np.concatenate(variances, axis=0)
