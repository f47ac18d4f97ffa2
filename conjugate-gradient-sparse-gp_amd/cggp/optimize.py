"""Inducing-parameter updates (next row F1) -- host mirror of `cggp/optimize.py:41-98`.

Given centres Z: assign every input to its nearest centre, pseudo_u = per-cluster mean of y,
counts = per-cluster size (-> Lambda = sigma^2 / counts, `cggp/models.py:226-228`).  The N x M
search and the per-cluster sums run fused in libmgp; with row-sharded data the [M] sums and
counts are summed over ranks by one all-reduce.
"""

import numpy as np
import torch

from . import ops


def nearest_centre_statistics(kernel, Z, data, distance_type="sqeuclidean", allreduce=None):
    """(idx [N], sums [M], counts [M]) for this rank's rows."""
    x, y = data
    spec = kernel.spec(Z.shape[1])
    idx = ops.nearest_center(spec, x, Z, distance_type=distance_type, return_distance=False)
    sums, counts = ops.cluster_stats(idx, y, Z.shape[0])
    if allreduce is not None:
        both = torch.stack([sums, counts])
        allreduce(both.view(-1))
        sums, counts = both[0], both[1]
    return idx, sums, counts


def covertree_update_inducing_parameters(model, data, distance_fn, spatial_resolution):
    """`optimize.py:19-38`: centroids / cluster means / counts of the finest cover-tree level with
    empty clusters dropped, returned on the device and in the dtype of `data`."""
    from .covertree import CoverTree
    x, y = data
    tree = CoverTree(distance_fn, data, spatial_resolution=spatial_resolution)
    means, counts = tree.cluster_mean_and_counts
    keep = counts.reshape(-1) != 0.0
    to = (lambda a, like: torch.from_numpy(np.ascontiguousarray(a)).to(device=like.device, dtype=like.dtype)) \
        if isinstance(x, torch.Tensor) else (lambda a, like: a)
    return to(tree.centroids[keep], x), to(means[keep], y), to(counts[keep], y)


def oips_update_inducing_parameters(model, data, Z, allreduce=None):
    """`optimize.py:41-78` with the centres given: square_distance argmin, empty clusters get
    count 1 (:70) and keep the reference's NaN mean (reduce_mean of nothing)."""
    _, sums, counts = nearest_centre_statistics(model.kernel, Z, data, "sqeuclidean", allreduce)
    means = sums / counts  # 0/0 -> NaN, as tf.reduce_mean of an empty selection
    new_counts = torch.where(counts != 0, counts, torch.ones_like(counts))
    return Z, means[:, None], new_counts[:, None]


def kmeans_update_inducing_parameters(model, data, distance_fn, Z, allreduce=None):
    """`optimize.py:81-98`: u = scatter_add(y) / counts, counts kept as they are.  `distance_fn` as the
    reference takes it (None, `euclid_distance`, a `create_distance_fn` result) or the type as a string."""
    from .distance import resolve_distance
    dt, _ = resolve_distance(distance_fn, model.kernel)
    _, sums, counts = nearest_centre_statistics(model.kernel, Z, data, dt, allreduce)
    return Z, (sums / counts)[:, None], counts[:, None]


def assign_inducing_parameters(model, iv, means, counts):
    """`create_model_and_update_fn.update_fn` (`cggp/cli_utils.py:394-411`)."""
    model.inducing_variable.Z = iv.to(model.inducing_variable.Z.dtype)
    if hasattr(model, "pseudo_u"):
        model.pseudo_u = means.to(model.pseudo_u.dtype)
        model.cluster_counts = counts.to(model.cluster_counts.dtype)
    return iv, means, counts


def make_param_callback(model):
    """`optimize.py:267-282`: kernel / likelihood parameters keyed as the reference logs them."""
    def _callback(*args, **kwargs):
        k, lik = model.kernel, model.likelihood
        return {"kernel/variance": np.asarray(k.variance, dtype=np.float64),
                "kernel/lengthscales": np.asarray(k.lengthscales, dtype=np.float64),
                "likelihood/variance": np.asarray(lik.variance, dtype=np.float64)}
    return _callback


def make_metrics_callback(model, train_data, test_data, batch_size, use_jit=True, print_on=True,
                          check_numerics=True):
    """`optimize.py:285-364`: a `step_callback(step)` returning `{"train/elbo", "test/rmse",
    "test/nlpd"}` -- test error and log predictive density accumulated over batches of
    `batch_size`, the training ELBO summed over training batches (models with internal data: one
    `elbo()` call)."""
    import json
    import math

    def step_callback(step, *args, **kwargs):
        x, y = test_data
        sq, lpd, n = 0.0, 0.0, 0
        for s in range(0, x.shape[0], batch_size):
            xb, yb = x[s:s + batch_size], y[s:s + batch_size]
            mu, var = model.predict_f(xb)
            lpd += float(model.likelihood.predict_log_density(xb, mu, var, yb).sum())
            sq += float(((yb - mu) ** 2).sum())
            n += xb.shape[0]
        import inspect
        if len(inspect.signature(model.elbo).parameters) == 0:  # internal-data models (SGPR)
            elbo = float(model.elbo())
        elif hasattr(model, "elbo_over_batches"):  # same sum, batch-independent solves done once
            elbo = float(model.elbo_over_batches(train_data, batch_size))
        else:
            xt, yt = train_data
            elbo = 0.0
            for s in range(0, xt.shape[0], batch_size):
                elbo += float(model.elbo((xt[s:s + batch_size], yt[s:s + batch_size])))
        metrics = {"train/elbo": float(elbo), "test/rmse": math.sqrt(sq / n), "test/nlpd": -lpd / n}
        if print_on:
            fmt = {k: np.format_float_scientific(v, precision=4) for k, v in metrics.items()}
            print(f"Step [{step}], metrics: {json.dumps(fmt)}")
        if check_numerics and not math.isfinite(elbo):
            raise FloatingPointError(f"The training ELBO has got an undefined value {elbo}")
        return metrics

    return step_callback
