"""The oracle against INDEPENDENT third-party implementations that are installed here (scikit-learn, SciPy).

TensorFlow / GPflow -- what the reference computes with -- cannot be imported in this container, and the reference
ships no golden vectors, so the oracle's GPflow arithmetic (Gaussian likelihood) stays
"parity unpinned" against GPflow's own outputs.  What CAN be checked is that the restated formulas are the ones the
literature and another maintained library use:

* `oracle.kernels.Kernel` (SquaredExponential, Matern-1/2, -3/2, -5/2 with ARD lengthscales, GPflow's
  parametrisation) against `sklearn.gaussian_process.kernels.RBF / Matern(nu=0.5, 1.5, 2.5)`;
* the CDGP and SGPR predictive equations in the limit where they ARE exact GP regression -- every point its own
  cluster / its own inducing point -- against `sklearn.gaussian_process.GaussianProcessRegressor`;
* the CG recurrence of `oracle/cg.py` (restating `cggp/conjugate_gradient.py:59-98`) against
  `scipy.sparse.linalg.cg` iterate by iterate, with and without a (Jacobi) preconditioner.

None of this replaces the reference; it rules out a mis-remembered constant or a transposed convention in the
restatement.  CPU only.
"""

import numpy as np
import pytest

from oracle import cg as ocg
from oracle import kernels as ok
from oracle import models as om

sk = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern  # noqa: E402


def _sk_kernel(name, variance, ls):
    base = RBF(length_scale=ls) if name == "se" else Matern(length_scale=ls, nu={"matern12": 0.5, "matern32": 1.5,
                                                                                 "matern52": 2.5}[name])
    return ConstantKernel(variance) * base


@pytest.mark.parametrize("name", ok.KERNEL_NAMES)
@pytest.mark.parametrize("D", [1, 3, 8])
def test_kernels_against_scikit_learn(name, D):
    rng = np.random.default_rng(D)
    X, Z = rng.standard_normal((60, D)), rng.standard_normal((25, D))
    ls = rng.uniform(0.5, 2.0, D)
    k = ok.Kernel(name, 1.7, ls)
    ref = _sk_kernel(name, 1.7, ls)(X, Z)
    # the expansion form of the squared distance (GPflow's `square_distance`) costs ~1e-15 |x|^2; Matern-1/2 takes
    # its square root, so at coincident points -- the diagonal of K(X) -- GPflow's form gives 1.69999995 where exact
    # differences give 1.7 (a property of the reference's formula, kept on purpose)
    tol = 1e-12 if name != "matern12" else 1e-7
    assert np.max(np.abs(k.K(X, Z) - ref)) < tol
    assert np.max(np.abs(k.K(X) - _sk_kernel(name, 1.7, ls)(X))) < (tol if name != "matern12" else 1e-6)
    assert np.allclose(k.K_diag(X), 1.7)
    # a point well away from the cancellation regime: full precision for every profile
    far = k.K(X[:1] + 3.0, Z) - _sk_kernel(name, 1.7, ls)(X[:1] + 3.0, Z)
    assert np.max(np.abs(far)) < 1e-14


@pytest.mark.parametrize("name", ["se", "matern32"])
def test_cdgp_and_sgpr_reduce_to_exact_gp_regression(name):
    """CDGP with every point its own cluster (Z = X, counts = 1, pseudo_u = y) and SGPR with Z = X are exact GP
    regression (`models.py:250-276,324-354`; Titsias' bound is tight at Z = X): mean and latent variance against
    scikit-learn's GaussianProcessRegressor with the same fixed hyper-parameters."""
    rng = np.random.default_rng(5)
    N, D, s2 = 80, 2, 0.1
    X = rng.standard_normal((N, D))
    y = np.sin(X).sum(1, keepdims=True) + np.sqrt(s2) * rng.standard_normal((N, 1))
    Xs = rng.standard_normal((30, D))
    ls = np.array([0.8, 1.3])
    gpr = sk.GaussianProcessRegressor(kernel=_sk_kernel(name, 1.4, ls), alpha=s2, optimizer=None).fit(X, y[:, 0])
    mu_ref, sd_ref = gpr.predict(Xs, return_std=True)
    k = ok.Kernel(name, 1.4, ls)
    twin = om.ClusterGP(k, s2, X, pseudo_u=y, cluster_counts=np.ones((N, 1)))
    mu, var = twin.predict_f(Xs)
    assert np.max(np.abs(mu[:, 0] - mu_ref)) < 1e-8 and np.max(np.abs(var[:, 0] - sd_ref ** 2)) < 1e-8
    cg = om.CGGP(k, s2, X, ocg.ConjugateGradient(1e-26, max_iterations=400), num_probes=None, pseudo_u=y,
                 cluster_counts=np.ones((N, 1)))
    mu_c, var_c = cg.predict_f(Xs)
    assert np.max(np.abs(mu_c[:, 0] - mu_ref)) < 1e-7 and np.max(np.abs(var_c[:, 0] - sd_ref ** 2)) < 1e-7
    s = om.SGPR((X, y), k, X, s2, jitter=1e-10)
    mu_s, var_s = s.predict_f(Xs)
    assert np.max(np.abs(mu_s[:, 0] - mu_ref)) < 1e-6 and np.max(np.abs(var_s[:, 0] - sd_ref ** 2)) < 1e-6
    # and SGPR's collapsed bound at Z = X is the exact log marginal likelihood
    assert abs(s.elbo() - gpr.log_marginal_likelihood_value_) < 1e-5 * abs(gpr.log_marginal_likelihood_value_)


@pytest.mark.parametrize("jacobi", [False, True])
def test_cg_recurrence_against_scipy(jacobi):
    """`oracle/cg.py` after k steps == SciPy's conjugate gradient after k iterations (the textbook recurrence;
    SciPy's stopping rule is bypassed by an unreachable tolerance and `maxiter`)."""
    from scipy.sparse.linalg import LinearOperator, cg
    rng = np.random.default_rng(11)
    n = 120
    Q = rng.standard_normal((n, n))
    A = Q @ Q.T / n + np.diag(rng.uniform(0.5, 4.0, n))
    b = rng.standard_normal(n)
    Minv = LinearOperator((n, n), matvec=lambda v: v / np.diag(A)) if jacobi else None
    pre = ocg.JacobiPreconditioner() if jacobi else None
    for k in (1, 2, 5, 12):
        x_ref, _ = cg(A, b, x0=np.zeros(n), rtol=1e-300, atol=0.0, maxiter=k, M=Minv)
        sol, (steps, err) = ocg.conjugate_gradient(A, b[None, :], np.zeros((1, n)), 0.0, pre, max_iterations=k,
                                                   max_steps_cycle=10 ** 6)
        assert steps == k
        assert np.max(np.abs(sol[0] - x_ref)) < 1e-10 * np.max(np.abs(x_ref))
