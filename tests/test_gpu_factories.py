"""The model / update-function factories of `cggp/cli_utils.py:143-436` (callers of the hot path):
`create_model_and_update_fn` for every clustering type, `create_predict_fn` +
`batch_posterior_computation`, checked against the oracle's pipeline on the same inputs."""
import warnings

import numpy as np
import pytest
import torch

from oracle import cg as ocg, cluster as oc, covertree as oct_, kernels as ok, models as om, selection as osel

pytestmark = pytest.mark.gpu


def T(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device="cuda:0", dtype=dtype)


def data(N=3000, D=2, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-2, 2, (N, D))
    y = np.sin(2 * X[:, :1]) * np.cos(X[:, 1:2]) + 0.1 * rng.standard_normal((N, 1))
    return X, y


def oracle_predict(Z, u, counts, Xs, D):
    ko = ok.Kernel("matern32", 1.0, np.ones(D))  # kernel_fn: Matern-3/2, unit parameters
    ref = om.CGGP(ko, 0.1, Z, ocg.ConjugateGradient(1e-13, max_iterations=5000), num_probes=None, pseudo_u=u,
                  cluster_counts=counts)
    return ref.predict_f(Xs)


def relerr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("clustering,kwargs", [
    ("oips", dict(rho=0.8, max_points=400)),
    ("covertree", dict(spatial_resolution=0.35)),
    ("kmeans", dict(max_points=40)),
    ("greedy", dict(max_points=50)),
    ("uniform", dict(max_points=60)),
    ("kmeans2", dict(max_points=30)),
])
def test_create_model_and_update_fn(clustering, kwargs):
    from cggp import cli_utils
    X, y = data()
    Xt, yt = T(X), T(y)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, update_fn = cli_utils.create_model_and_update_fn(
            lambda *a, **k: cli_utils.cdgp_class(*a, error_threshold=1e-13, **k), (Xt, yt), clustering,
            model_kwargs=dict(num_inducing_points=40), clustering_kwargs=kwargs)
        Z0 = model.inducing_variable.Z.cpu().numpy().copy()
        assert Z0.shape == (40, 2) and cli_utils.kernel_to_name(model.kernel) == "matern32"
        iv, means, counts = update_fn()
    M = iv.shape[0]
    assert model.inducing_variable.Z.shape == (M, 2) and model.pseudo_u.shape == (M, 1)
    assert model.cluster_counts.shape == (M, 1)
    Z = iv.cpu().numpy()
    ko = ok.Kernel("matern32", 1.0, np.ones(2))
    # ---- the clustering itself against the oracle where it is deterministic
    if clustering == "oips":
        Zo, _ = osel.oips(ko, X, 0.8, 400)
        assert np.array_equal(Z, Zo)
    elif clustering == "covertree":
        Zo, mo, co = oct_.covertree_update_inducing_parameters((X, y), 0.35)
        assert np.allclose(Z, Zo, atol=1e-13) and np.array_equal(counts.cpu().numpy(), co)
        assert np.allclose(means.cpu().numpy(), mo, atol=1e-13)
    elif clustering == "kmeans":
        Zo, _ = osel.kmeans_lloyd(X, 40, initial_centroids=Z0)
        assert np.allclose(Z, Zo, atol=1e-9)
    # ---- statistics: every row counted once (empty clusters are reported as 1 by the oips family)
    c = counts.cpu().numpy().reshape(-1)
    if clustering in ("kmeans", "kmeans2", "covertree"):
        assert c.sum() == X.shape[0]
    else:
        idx = oc.nearest_centre_sqdist(Z, X)
        u0, c0 = oc.cluster_stats(idx, y, M)
        assert np.array_equal(c, c0.reshape(-1))
        assert np.allclose(means.cpu().numpy(), u0, atol=1e-12, equal_nan=True)
    # ---- prediction through the factories against the oracle's CGGP with the same parameters
    ok_rows = ~np.isnan(means.cpu().numpy()).reshape(-1) & (c > 0)
    if ok_rows.all():
        predict_fn = cli_utils.create_predict_fn(model)
        mu, var = cli_utils.batch_posterior_computation(predict_fn, (Xt[:500], yt[:500]), batch_size=128)
        mu0, var0 = oracle_predict(Z, means.cpu().numpy(), counts.cpu().numpy(), X[:500], 2)
        assert mu.shape == (500, 1) and var.shape == (500, 1)
        assert relerr(mu, mu0) < 1e-6 and np.max(np.abs(var - var0)) < 1e-6


def test_name_to_kernel_and_errors():
    from cggp import cli_utils, kernels
    k = cli_utils.name_to_kernel("se", 3)
    assert isinstance(k, kernels.SquaredExponential) and list(k.lengthscales) == [0.1] * 3
    assert cli_utils.kernel_to_name(cli_utils.name_to_kernel("matern12", 2)) == "matern12"
    with pytest.raises(NotImplementedError):
        cli_utils.name_to_kernel("rbf")
    with pytest.raises(NotImplementedError):
        cli_utils.kernel_to_name(kernels.Matern52(1.0, [1.0]))
    X, y = data(200)
    with pytest.raises(ValueError):
        cli_utils.create_update_fn("spectral", None, (T(X), T(y)))
    with pytest.raises(ValueError):
        cli_utils.create_uniform_update_fn(None, (T(X), T(y)), max_points=500)


def test_metrics_and_param_callbacks():
    """`make_metrics_callback` / `make_param_callback` (optimize.py:267-364) against the oracle's
    closed forms on the same model."""
    from cggp import cli_utils
    from cggp.optimize import make_metrics_callback, make_param_callback
    X, y = data(1200)
    Xt, yt = T(X[:1000]), T(y[:1000])
    Xs, ys = T(X[1000:]), T(y[1000:])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, update_fn = cli_utils.create_model_and_update_fn(
            lambda *a, **k: cli_utils.cdgp_class(*a, error_threshold=1e-13, num_probes=None, **k), (Xt, yt), "oips",
            model_kwargs=dict(num_inducing_points=30), clustering_kwargs=dict(rho=0.7, max_points=200))
        iv, means, counts = update_fn()
    cb = make_metrics_callback(model, (Xt, yt), (Xs, ys), batch_size=128, print_on=False)
    m = cb(3)
    ko = ok.Kernel("matern32", 1.0, np.ones(2))
    ref = om.CGGP(ko, 0.1, iv.cpu().numpy(), ocg.ConjugateGradient(1e-13, max_iterations=5000), num_probes=None,
                  pseudo_u=means.cpu().numpy(), cluster_counts=counts.cpu().numpy(), num_data=1000)
    mu0, var0 = ref.predict_f(X[1000:])
    r0, n0 = om.rmse_nlpd(mu0, var0, y[1000:], 0.1)
    assert abs(m["test/rmse"] - r0) < 1e-6 * r0 and abs(m["test/nlpd"] - n0) < 1e-6 * abs(n0)
    Xtr, ytr = X[:1000], y[:1000]
    e0 = sum(ref.elbo((Xtr[s:s + 128], ytr[s:s + 128])) for s in range(0, 1000, 128))
    assert abs(m["train/elbo"] - e0) < 1e-6 * abs(e0)
    direct = sum(float(model.elbo((Xt[s:s + 128], yt[s:s + 128]))) for s in range(0, 1000, 128))
    assert abs(model.elbo_over_batches((Xt, yt), 128, shared_inverse=False) - direct) < 1e-8 * abs(direct)
    assert abs(model.elbo_over_batches((Xt, yt), 128, shared_inverse=True) - direct) < 1e-7 * abs(direct)
    p = make_param_callback(model)()
    assert float(p["kernel/variance"]) == 1.0 and list(p["kernel/lengthscales"]) == [1.0, 1.0]
    assert abs(float(p["likelihood/variance"]) - 0.1) < 1e-15
