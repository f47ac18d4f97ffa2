"""Next row F2: gradients through the kernel blocks and the CG model, and the Adam loop.

The reference's own gradient tests (`cggp/cg_test.py:34-46,68-77`) compare CG-based gradients
with autodiff through direct solves; here the autograd gradient of the CG model's ELBO is compared
with finite differences of the oracle's Cholesky twin (whose value includes log|Kmm+Lambda|, which
the CG model carries only in its gradient, `cggp/models.py:30-46`)."""

import numpy as np
import pytest
import torch

from oracle import cluster as oc, kernels as ok, models as om

pytestmark = pytest.mark.gpu
KINDS = ["se", "matern12", "matern32", "matern52"]


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.mark.parametrize("name", KINDS)
@pytest.mark.parametrize("D", [1, 3, 8])
def test_k_dense_vjp_matches_finite_differences(name, D):
    from cggp import ops
    rng = np.random.default_rng(0)
    A, B = rng.standard_normal((150, D)), rng.standard_normal((60, D)) + 0.3
    G = rng.standard_normal((150, 60))
    var, ls = 1.3, rng.random(D) + 0.7
    spec = ops.KernelSpec(name, var, ls.tolist(), D)
    dvar, dls = ops.k_dense_vjp(spec, T(A), T(B), T(G))
    f = lambda v, l: float(np.sum(G * ok.Kernel(name, v, l).K(A, B)))
    h = 1e-6
    assert abs(dvar - (f(var + h, ls) - f(var - h, ls)) / (2 * h)) < 1e-6 * max(1.0, abs(dvar))
    for d in range(D):
        e = np.zeros(D)
        e[d] = h
        fd = (f(var, ls + e) - f(var, ls - e)) / (2 * h)
        assert abs(dls[d] - fd) < 2e-6 * max(1.0, abs(fd)), (d, dls[d], fd)


@pytest.mark.parametrize("name", KINDS)
@pytest.mark.parametrize("D", [33, 77, 90])
def test_k_dense_vjp_any_dimension(name, D):
    """Row F2 above the fused limit (D = 77 / 90 are the reference's `buzz` / `song`, cli_utils.py:72-86): the
    tile-wise reduction of csrc/generic.hip against finite differences of the oracle's kernel, every lengthscale."""
    from cggp import ops
    rng = np.random.default_rng(0)
    na, nb = 150, 70  # ragged against the 64-wide tiles
    A, B = rng.standard_normal((na, D)), rng.standard_normal((nb, D)) + 0.3
    B[:3] = A[:3]  # coincident pairs: the Matern floor (f' = 0 there)
    G = rng.standard_normal((na, nb))
    var, ls = 1.3, (rng.random(D) + 0.7) * np.sqrt(D)
    spec = ops.KernelSpec(name, var, ls.tolist(), D)
    dvar, dls = ops.k_dense_vjp(spec, T(A), T(B), T(G))
    dls = np.asarray(dls)

    def f(v, l):
        k = ok.Kernel(name, v, l)
        d = (A[:, None, :] - B[None, :, :]) / l  # direct differences: the derivative of the exact kernel
        return float(np.sum(G * k.K_r2(np.sum(d * d, -1))))

    h = 1e-6
    assert abs(dvar - (f(var + h, ls) - f(var - h, ls)) / (2 * h)) < 1e-6 * max(1.0, abs(dvar))
    scale = max(1.0, float(np.max(np.abs(dls))))
    for d in range(D):
        e = np.zeros(D)
        e[d] = h * ls[d]
        fd = (f(var, ls + e) - f(var, ls - e)) / (2 * h * ls[d])
        assert abs(dls[d] - fd) < 2e-6 * scale, (d, dls[d], fd)
    # strided G (a view into a wider matrix) and the fp32 instantiation
    Gw = np.zeros((na, nb + 9))
    Gw[:, :nb] = G
    dvar_s, dls_s = ops.k_dense_vjp(spec, T(A), T(B), T(Gw)[:, :nb])
    assert dvar_s == dvar and np.array_equal(np.asarray(dls_s), dls)
    dvar32, dls32 = ops.k_dense_vjp(spec, T(A).float(), T(B).float(), T(G).float())
    assert abs(dvar32 - dvar) < 2e-4 * max(1.0, abs(dvar)) and np.max(np.abs(np.asarray(dls32) - dls)) < 2e-4 * scale


@pytest.mark.parametrize("D", [33, 77])
def test_elbo_gradient_any_dimension(D):
    """The ELBO gradient through CG with D > 32 inputs against finite differences of the oracle's Cholesky twin."""
    from cggp import kernels
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.training import TrainableCGGP
    name = "matern32"
    X, y, Z, u, counts = _problem(name, D=D)
    rng = np.random.default_rng(4)
    var, ls, s2 = 1.2, (0.8 + 0.4 * rng.random(D)) * np.sqrt(D), 0.15
    m = TrainableCGGP(kernels.Matern32(var, ls), s2, T(Z), ConjugateGradient(1e-15, max_iterations=5000),
                      num_probes=None, pseudo_u=T(u), cluster_counts=T(counts), num_data=X.shape[0])
    xb, yb = X[:100], y[:100]
    e = m.elbo((T(xb), T(yb)))
    e.backward()
    sig = lambda p: torch.sigmoid(p.raw.detach())
    g_var = float(m.kernel.variance_p.raw.grad / sig(m.kernel.variance_p))
    g_ls = (m.kernel.lengthscales_p.raw.grad / sig(m.kernel.lengthscales_p)).numpy()

    def twin(v, l, s):
        t = om.ClusterGP(ok.Kernel(name, v, l), s, Z, pseudo_u=u, cluster_counts=counts, num_data=X.shape[0])
        return float(t.elbo((xb, yb)))

    h = 1e-5
    fd_var = (twin(var + h, ls, s2) - twin(var - h, ls, s2)) / (2 * h)
    assert abs(g_var - fd_var) < 1e-4 * max(1.0, abs(fd_var)), (g_var, fd_var)
    scale = max(1.0, float(np.max(np.abs(g_ls))))
    for d in (0, 1, D // 2, 32, D - 1):
        dl = np.zeros(D)
        dl[d] = h
        fd = (twin(var, ls + dl, s2) - twin(var, ls - dl, s2)) / (2 * h)
        assert abs(g_ls[d] - fd) < 1e-4 * scale, (d, g_ls[d], fd)


def _problem(name, N=240, D=2, M=14, seed=1):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    y = np.sin(X).sum(1, keepdims=True) + 0.3 * rng.standard_normal((N, 1))
    Z = X[rng.choice(N, M, replace=False)]
    idx = oc.nearest_centre_sqdist(Z, X)
    u, counts = oc.cluster_stats(idx, y, M)
    return X, y, Z, u, counts


@pytest.mark.parametrize("name", ["se", "matern32", "matern52"])
def test_elbo_gradient_matches_cholesky_twin(name):
    from cggp import kernels
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.training import TrainableCGGP
    X, y, Z, u, counts = _problem(name)
    var, ls, s2 = 1.2, np.array([0.9, 1.4]), 0.15
    cls = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32, "matern52": kernels.Matern52}[name]
    m = TrainableCGGP(cls(var, ls), s2, T(Z), ConjugateGradient(1e-15, max_iterations=5000), num_probes=None,
                      pseudo_u=T(u), cluster_counts=T(counts), num_data=X.shape[0])
    xb, yb = X[:100], y[:100]
    e = m.elbo((T(xb), T(yb)))
    e.backward()
    sig = lambda p: torch.sigmoid(p.raw.detach())
    g_var = float(m.kernel.variance_p.raw.grad / sig(m.kernel.variance_p))
    g_ls = (m.kernel.lengthscales_p.raw.grad / sig(m.kernel.lengthscales_p)).numpy()
    g_s2 = float(m.noise_p.raw.grad / sig(m.noise_p))

    def twin(v, l, s):
        t = om.ClusterGP(ok.Kernel(name, v, l), s, Z, pseudo_u=u, cluster_counts=counts, num_data=X.shape[0])
        return float(t.elbo((xb, yb)))

    # value: the CG model omits 0.5*log|Kmm+Lambda| (models.py:46); its gradient does not
    KL = om.add_diagonal(ok.Kuu(Z, ok.Kernel(name, var, ls)), (s2 / counts)[:, 0])
    assert abs(float(e) - (twin(var, ls, s2) + 0.5 * np.linalg.slogdet(KL)[1])) < 1e-6 * abs(float(e))
    h = 1e-5
    fd_var = (twin(var + h, ls, s2) - twin(var - h, ls, s2)) / (2 * h)
    fd_s2 = (twin(var, ls, s2 + h) - twin(var, ls, s2 - h)) / (2 * h)
    assert abs(g_var - fd_var) < 1e-4 * max(1.0, abs(fd_var)), (g_var, fd_var)
    assert abs(g_s2 - fd_s2) < 1e-4 * max(1.0, abs(fd_s2)), (g_s2, fd_s2)
    for d in range(2):
        dl = np.zeros(2)
        dl[d] = h
        fd = (twin(var, ls + dl, s2) - twin(var, ls - dl, s2)) / (2 * h)
        assert abs(g_ls[d] - fd) < 1e-4 * max(1.0, abs(fd)), (d, g_ls[d], fd)


def test_fused_solves_equal_solve_by_solve():
    """One CG over [pseudo_u | Kmn | probes] with the log-det gradient reusing K^-1 Zp gives the same
    ELBO and the same parameter gradients as the reference's solve-by-solve order (same probes),
    and with probes = all +-basis columns (P = M) the gradient of the exact branch."""
    from cggp import kernels
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.training import TrainableCGGP
    X, y, Z, u, counts = _problem("se")
    rng = np.random.default_rng(3)
    probes = T(rng.choice([-1.0, 1.0], size=(Z.shape[0], 6)))
    out = {}
    for fused in (True, False):
        m = TrainableCGGP(kernels.SquaredExponential(1.2, [0.9, 1.4]), 0.15, T(Z),
                          ConjugateGradient(1e-15, max_iterations=5000), num_probes=6, pseudo_u=T(u),
                          cluster_counts=T(counts), num_data=X.shape[0], fused_solves=fused)
        e = m.elbo((T(X[:80]), T(y[:80])), probes=probes)
        e.backward()
        out[fused] = (float(e), [p.grad.clone() for p in m.parameters()])
    # both stop at the reference's guard floor (||r|| ~ 1e-8): agreement to that level
    assert abs(out[True][0] - out[False][0]) < 1e-7 * abs(out[False][0])
    for ga, gb in zip(out[True][1], out[False][1]):
        assert float((ga - gb).abs().max()) < 1e-5 * max(1.0, float(gb.abs().max()))


def test_adam_training_reduces_the_loss_and_updates_inducing_parameters():
    from cggp import kernels
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.optimize import oips_update_inducing_parameters
    from cggp.training import TrainableCGGP, train_using_adam_and_update
    X, y, Z, u, counts = _problem("se", N=2000, D=2, M=32, seed=2)
    m = TrainableCGGP(kernels.SquaredExponential(0.3, [3.0, 3.0]), 1.0, T(Z), ConjugateGradient(1e-10),
                      num_probes=5, pseudo_u=T(u), cluster_counts=T(counts), num_data=2000)
    calls = []

    def update_fn():
        _, means, c = oips_update_inducing_parameters(m.frozen_model(), (T(X), T(y)), T(Z))
        m.pseudo_u, m.cluster_counts = means, c
        calls.append(1)

    losses = train_using_adam_and_update((T(X), T(y)), m, iterations=40, batch_size=500, learning_rate=0.05,
                                         update_fn=update_fn, update_during_training=True)
    assert len(losses) == 40 and len(calls) == 41
    assert np.mean(losses[-5:]) < np.mean(losses[:5])
    assert all(np.isfinite(losses))
    fm = m.frozen_model()
    mu, var = fm.predict_f(T(X[:50]))
    assert torch.isfinite(mu).all() and torch.isfinite(var).all()


def test_lbfgs_training_reduces_the_full_batch_loss():
    """`train_using_lbfgs_and_update` (optimize.py:152-195): the full-batch loss falls monotonically
    over accepted steps (fixed probes make it deterministic), the callbacks fire as upstream, and the
    optimum is a stationary point of the Cholesky twin's ELBO up to the Hutchinson noise."""
    from cggp import kernels
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.training import TrainableCGGP, train_using_lbfgs_and_update
    X, y, Z, u, counts = _problem("se", N=300, M=16)
    m = TrainableCGGP(kernels.SquaredExponential(0.4, [2.5, 2.5]), 0.8, T(Z), ConjugateGradient(1e-13, max_iterations=4000),
                      num_probes=None, pseudo_u=T(u), cluster_counts=T(counts), num_data=X.shape[0])
    seen = []
    l0 = float(m.training_loss((T(X), T(y))))
    res = train_using_lbfgs_and_update((T(X), T(y)), m, 25, monitor=lambda it: seen.append(it))
    l1 = float(m.training_loss((T(X), T(y))))
    assert l1 < l0 - 1.0 and abs(l1 - res.fun) < 1e-6 * abs(l1)
    assert seen[0] == 0 and seen[1:] == list(range(1, res.nit + 1))
    assert train_using_lbfgs_and_update((T(X), T(y)), m, 0, monitor=lambda it: seen.append(it)) is None
    assert seen[-2:] == [0, -1]
    # probes: deterministic objective (same probes every evaluation)
    mp = TrainableCGGP(kernels.SquaredExponential(0.4, [2.5, 2.5]), 0.8, T(Z), ConjugateGradient(1e-13, max_iterations=4000),
                       num_probes=8, pseudo_u=T(u), cluster_counts=T(counts), num_data=X.shape[0])
    r2 = train_using_lbfgs_and_update((T(X), T(y)), mp, 10)
    assert r2.nit >= 1 and np.isfinite(r2.fun)


def test_independent_logdet_probes_match_the_reference_estimator():
    """`independent_logdet_probes=True`: the log-det gradient comes from `eval_logdet`'s own probe draw and its
    own probe solve (`cggp/models.py:38-41`), not from the trace estimator's solution.  With the two probe sets
    injected by hand the parameter gradient equals (value part) + (1/P) K^-1 Zq (Zq)^T contracted with dK, i.e.
    what `eval_logdet_grad` returns for Zq; and with Zq == Zp it equals the default (reusing) path."""
    from cggp import kernels
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.models import rademacher
    from cggp.training import TrainableCGGP
    X, y, Z, u, counts = _problem("se")
    M = Z.shape[0]
    Zp = rademacher((M, 6), torch.float64, dev(), 0)

    def grads(independent, own_seed):
        m = TrainableCGGP(kernels.SquaredExponential(1.2, [0.9, 1.4]), 0.15, T(Z),
                          ConjugateGradient(1e-15, max_iterations=5000), num_probes=6, pseudo_u=T(u),
                          cluster_counts=T(counts), num_data=X.shape[0], independent_logdet_probes=independent)
        m.logdet_probe_seed = own_seed
        e = m.elbo((T(X[:80]), T(y[:80])), probes=Zp)
        e.backward()
        return float(e), torch.cat([p.grad.reshape(-1) for p in m.parameters()])

    e_reuse, g_reuse = grads(False, 0)
    e_same, g_same = grads(True, 0)      # own draw from seed 0 == Zp: the same estimator, one more solve
    e_other, g_other = grads(True, 12345)
    assert e_reuse == e_same == e_other  # the value never contains the log-det (models.py:46)
    assert float((g_same - g_reuse).abs().max()) < 1e-6 * float(g_reuse.abs().max())
    assert float((g_other - g_reuse).abs().max()) > 1e-4 * float(g_reuse.abs().max())  # a different draw


def test_vanilla_lbfgs_trainers():
    """`train_vanilla_using_lbfgs` / `..._and_standard_ip_update` (optimize.py:101-150)."""
    from cggp import kernels
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.training import (TrainableCGGP, train_vanilla_using_lbfgs,
                               train_vanilla_using_lbfgs_and_standard_ip_update)
    X, y, Z, u, counts = _problem("se", N=300, M=16)

    def make():
        return TrainableCGGP(kernels.SquaredExponential(0.4, [2.5, 2.5]), 0.8, T(Z),
                             ConjugateGradient(1e-13, max_iterations=4000), num_probes=None, pseudo_u=T(u),
                             cluster_counts=T(counts), num_data=X.shape[0])
    m = make()
    l0 = float(m.training_loss((T(X), T(y))))
    res = train_vanilla_using_lbfgs((T(X), T(y)), m, clustering_fn=None, max_num_iters=15)
    assert float(m.training_loss((T(X), T(y)))) < l0 - 1.0 and res.nit >= 1
    m2, calls = make(), []

    def clustering_fn():
        calls.append(1)
        return T(Z) + 0.0  # same centres: the loss must still fall, and Z is re-assigned every step

    res2 = train_vanilla_using_lbfgs_and_standard_ip_update((T(X), T(y)), m2, clustering_fn, 8)
    assert len(calls) == res2.nit + 1 and float(m2.training_loss((T(X), T(y)))) < l0
