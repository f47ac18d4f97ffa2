"""Predictive mean / variance -- the quantity the north star's 1e-6 tolerance is stated on
(`cggp/models.py:324-354`) -- at the inducing-set size of every BASELINE.json config:

* C2 (N=100 000, D=8, M=2048), C3 (N=2^20, D=8, M=4096) and C5's shape (N=2^20, D=32, M=4096, Matern-3/2; the
  lengthscale raised to sqrt(D) so that the system is not numerically diagonal): `CGGP.predict_f` on a 256-row batch against
  `oracle/models.py:CGGP.predict_f` given the SAME `pseudo_u` / `cluster_counts`
    - near the reference recurrence's guard floor (thr 1e-15, cap 4M): 1e-6 on the mean, 1e-6 * k** on the
      variance, against the oracle's Cholesky twin (`ClusterGP.predict_f`, `models.py:250-276`) and, at C2, also
      against the oracle's own CG run to the same threshold;
    - at the reference's threshold (`cli_utils.py:439`: 1e-6): HIP CG vs oracle CG, each stopping by the same
      rule on its own rounding trajectory, and both against the Cholesky twin --
      what `0.5||r||^2 <= 1e-6` leaves in (mu, var) is measured, and the two implementations must sit inside it.
* `predict_f_batched(shared_inverse=True)` (build-side option) against the per-batch form and the oracle at
  the same sizes.
* C5 (Matern-3/2, D=32, M=4096): three fixed CG steps of the SGPR operator on a 65 536-row slice against
  `oracle/cg.py` driving a dense fp64 operator on the host (the C4 pattern of tests/test_gpu_configs.py).

Oracle cost: a [257, M] x [M, M] product per CG step on the host -- 0.05 s at M=4096, so the oracle's CG runs at
the reference's threshold (~360 steps) everywhere, and at the tight threshold only for C2.
"""

import numpy as np
import pytest
import torch

from oracle import cg as ocg
from oracle import kernels as ok
from oracle import models as om

pytestmark = pytest.mark.gpu

B = 256  # rows of the prediction batch (the reference's batches are 1000-5000, configs/geospatial.toml:33-34)


def dev():
    return torch.device("cuda:0")


def _cdgp(cfg):
    from cggp import kernels, synthetic
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.models import CGGP
    from cggp.optimize import assign_inducing_parameters, oips_update_inducing_parameters
    N, D, M, dt, kname = synthetic.CONFIGS[cfg]
    syn = synthetic.make_inputs(N, D, M, dt)
    X, y, Z = (torch.from_numpy(a).to(dev()) for a in (syn.X, syn.y, syn.Z))
    # C5's kernel is Matern-3/2 in D = 32; its lengthscale is set to sqrt(D) here so that neighbouring points
    # correlate (with l = 1 in 32 dimensions k(x, z) ~ 1e-5 off the diagonal and Kmm + Lambda is numerically diagonal:
    # nothing for CG or the variance to do)
    ls = 1.0 if kname == "se" else float(np.sqrt(D))
    kern = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32}[kname](1.0, [ls] * D)
    m = CGGP(kern, 0.1, Z, ConjugateGradient(1e-6), num_probes=None, num_data=N)
    assign_inducing_parameters(m, *oips_update_inducing_parameters(m, (X, y), Z))
    ko = ok.Kernel(kname, 1.0, ls * np.ones(D))
    u, counts = m.pseudo_u.cpu().numpy(), m.cluster_counts.cpu().numpy()
    assert counts.sum() == N and counts.min() >= 1
    # prediction rows: a mix of training rows (coincident with some inducing points) and perturbed ones
    rng = np.random.default_rng(21)
    rows = rng.integers(0, N, 4 * B)
    Xs = syn.X[rows].copy()
    Xs[B // 2:] += 0.05 * rng.standard_normal(Xs[B // 2:].shape)
    del X, y
    return syn, m, ko, u, counts, Xs


@pytest.fixture(scope="module", params=["C2", "C3", "C5"])
def cdgp(request):
    out = _cdgp(request.param)
    yield (request.param,) + out
    torch.cuda.empty_cache()


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def test_predict_f_config_size_near_guard_floor(cdgp):
    """thr 1e-15: both CGs run to (near) the recurrence's floor; (mu, var) meet 1e-6 / 1e-6 k**."""
    from cggp.conjugate_gradient import ConjugateGradient
    cfg, syn, m, ko, u, counts, Xs = cdgp
    M = syn.Z.shape[0]
    xb = Xs[:B]
    m.conjugate_gradient = ConjugateGradient(1e-15, max_iterations=4 * M)
    mu, var = m.predict_f(torch.from_numpy(xb).to(dev()))
    mu, var = mu.cpu().numpy(), var.cpu().numpy()
    assert mu.shape == (B, 1) and var.shape == (B, 1)
    twin = om.ClusterGP(ko, 0.1, syn.Z, pseudo_u=u, cluster_counts=counts, num_data=syn.X.shape[0])
    mu_c, var_c = twin.predict_f(xb)
    kss = 1.0  # k** = variance of the kernel
    assert _rel(mu, mu_c) < 1e-6, _rel(mu, mu_c)
    assert np.max(np.abs(var - var_c)) / kss < 1e-6, np.max(np.abs(var - var_c))
    assert var.min() > 0 and var.max() <= kss + 1e-9
    if cfg == "C2":  # the oracle's own CG to the same threshold (M=2048: ~15 ms per host step)
        ref = om.CGGP(ko, 0.1, syn.Z, ocg.ConjugateGradient(1e-15, max_iterations=4 * M), num_probes=None,
                      pseudo_u=u, cluster_counts=counts, num_data=syn.X.shape[0])
        mu_o, var_o = ref.predict_f(xb)
        assert _rel(mu, mu_o) < 1e-6 and np.max(np.abs(var - var_o)) / kss < 1e-6
        assert _rel(mu_o, mu_c) < 1e-6 and np.max(np.abs(var_o - var_c)) / kss < 1e-6


def test_predict_f_config_size_reference_threshold(cdgp):
    """thr 1e-6 (`cli_utils.py:439`), cap n (`conjugate_gradient.py:190-192`): the reference's own setting.
    HIP and oracle CG stop by the same rule; each is as far from the exact (Cholesky) value as that rule
    leaves it, and they must agree with each other at least as well as either agrees with the exact value."""
    from cggp.conjugate_gradient import ConjugateGradient
    cfg, syn, m, ko, u, counts, Xs = cdgp
    xb = Xs[:B]
    m.conjugate_gradient = ConjugateGradient(1e-6)
    mu, var = m.predict_f(torch.from_numpy(xb).to(dev()))
    steps_hip = int(m.conjugate_gradient.last_stats[0])
    mu, var = mu.cpu().numpy(), var.cpu().numpy()
    cgo = ocg.ConjugateGradient(1e-6)
    ref = om.CGGP(ko, 0.1, syn.Z, cgo, num_probes=None, pseudo_u=u, cluster_counts=counts,
                  num_data=syn.X.shape[0])
    mu_o, var_o = ref.predict_f(xb)
    twin = om.ClusterGP(ko, 0.1, syn.Z, pseudo_u=u, cluster_counts=counts, num_data=syn.X.shape[0])
    mu_c, var_c = twin.predict_f(xb)
    e_mu_o, e_var_o = _rel(mu_o, mu_c), float(np.max(np.abs(var_o - var_c)))
    e_mu, e_var = _rel(mu, mu_c), float(np.max(np.abs(var - var_c)))
    d_mu, d_var = _rel(mu, mu_o), float(np.max(np.abs(var - var_o)))
    print(f"\n{cfg} thr 1e-6: HIP steps {steps_hip}; mean HIP-exact {e_mu:.2e} oracle-exact {e_mu_o:.2e} "
          f"HIP-oracle {d_mu:.2e}; var HIP-exact {e_var:.2e} oracle-exact {e_var_o:.2e} HIP-oracle {d_var:.2e}")
    # what the stopping rule leaves: bounded for both by the same constant (measured ~1e-5 .. 1e-4)
    assert e_mu < 1e-3 and e_mu_o < 1e-3 and e_var < 1e-3 and e_var_o < 1e-3
    # the HIP path is no further from the exact value than the oracle's CG is, up to a factor for the
    # rounding-dependent step at which each trajectory crosses the threshold
    assert e_mu <= 4 * e_mu_o + 1e-9 and e_var <= 4 * e_var_o + 1e-9
    # and the two CG implementations agree inside that same envelope
    assert d_mu <= 4 * max(e_mu, e_mu_o) + 1e-9 and d_var <= 4 * max(e_var, e_var_o) + 1e-9


def test_predict_f_batched_shared_inverse_config_size(cdgp):
    """`predict_f_batched(shared_inverse=True)`: one M-column CG + one GEMM per batch, against the per-batch
    B-column CG (`cli_utils.py:426-436` over `models.py:340`) and against the oracle's closed form."""
    from cggp.conjugate_gradient import ConjugateGradient
    cfg, syn, m, ko, u, counts, Xs = cdgp
    M = syn.Z.shape[0]
    xt = torch.from_numpy(Xs).to(dev())  # 4 batches of 256 rows
    twin = om.ClusterGP(ko, 0.1, syn.Z, pseudo_u=u, cluster_counts=counts, num_data=syn.X.shape[0])
    mu_c, var_c = twin.predict_f(Xs)
    # tight: columns of the inverse to 1e-24 -> the guard floor; per-batch form at 1e-15
    m.conjugate_gradient = ConjugateGradient(1e-15, max_iterations=4 * M)
    mu_s, var_s = m.predict_f_batched(xt, B, shared_inverse=True)
    mu_b, var_b = m.predict_f_batched(xt, B, shared_inverse=False)
    mu_s, var_s, mu_b, var_b = (t.cpu().numpy() for t in (mu_s, var_s, mu_b, var_b))
    assert mu_s.shape == (4 * B, 1) and var_s.shape == (4 * B, 1)
    assert _rel(mu_s, mu_c) < 1e-6 and np.max(np.abs(var_s - var_c)) < 1e-6
    assert _rel(mu_b, mu_c) < 1e-6 and np.max(np.abs(var_b - var_c)) < 1e-6
    assert _rel(mu_s, mu_b) < 1e-6 and np.max(np.abs(var_s - var_b)) < 1e-6
    # the reference's threshold: the shared inverse is solved to thr^2, so it is the MORE accurate of the two
    m.conjugate_gradient = ConjugateGradient(1e-6)
    mu_s6, var_s6 = m.predict_f_batched(xt, B, shared_inverse=True)
    mu_b6, var_b6 = m.predict_f_batched(xt, B, shared_inverse=False)
    mu_s6, var_s6, mu_b6, var_b6 = (t.cpu().numpy() for t in (mu_s6, var_s6, mu_b6, var_b6))
    e_s, e_b = float(np.max(np.abs(var_s6 - var_c))), float(np.max(np.abs(var_b6 - var_c)))
    print(f"\n{cfg} batched thr 1e-6: var shared-exact {e_s:.2e} per-batch-exact {e_b:.2e}; "
          f"inverse CG steps {int(m.inverse_stats[0])}")
    assert e_s <= e_b + 1e-9 and e_b < 1e-3
    # both forms share `a` (one solve at the model's threshold): the means are the same computation
    assert _rel(mu_s6, mu_b6) < 1e-12


class _HostSgprOperator:
    """Dense fp64 S = s2 (Kmm + jI) + K_mn K_nm applied on the host: oracle/models.py:SgprNormalOperator with
    the K_nm chunks evaluated by oracle/cpu_baseline.py on every host core."""

    def __init__(self, X, Z, ko, name, s2, jitter):
        from oracle import cpu_baseline
        self._apply = cpu_baseline.sgpr_operator_apply
        self.X, self.Z = torch.from_numpy(X), torch.from_numpy(Z)
        self.Kmm = torch.from_numpy(ok.Kuu(Z, ko, jitter=jitter))
        self.s2, self.name = s2, name
        self.shape = (Z.shape[0], Z.shape[0])
        self.ls = torch.ones(Z.shape[1], dtype=torch.float64)

    def rmatmul(self, P):
        out = self._apply(self.X, self.Z, torch.from_numpy(np.ascontiguousarray(P.T)), self.Kmm, self.s2, 1.0,
                          self.ls, self.name, chunk=8192)
        return out.numpy().T


def test_c5_sgpr_cg_steps_against_oracle():
    """C5's "CG" leg: Matern-3/2, D=32, M=4096, fp64.  Three steps of S alpha = K_mn y on a 65 536-row slice
    (2.7e8 pairs per host operator application, four of them) against oracle/cg.py, step count, iterate and
    error statistic; then the same three steps at the full N = 2^20 against the true residual."""
    from cggp import kernels, ops, synthetic
    from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient
    N, D, M, dt, kname = synthetic.CONFIGS["C5"]
    assert (kname, D, M) == ("matern32", 32, 4096)
    syn = synthetic.make_inputs(N, D, M, dt)
    X, y, Z = (torch.from_numpy(a).to(dev()) for a in (syn.X, syn.y, syn.Z))
    kern = kernels.Matern32(1.0, [1.0] * D)
    ko = ok.Kernel(kname, 1.0, np.ones(D))
    ns = 65536
    Xs, ys = X[:ns].contiguous(), y[:ns].contiguous()
    op = SgprNormalOperator(kern, Xs, Z, 0.1, jitter=1e-6)
    rhs = ops.kmn_matvec(kern.spec(D), Xs, Z, ys).t().contiguous()
    oop = _HostSgprOperator(syn.X[:ns], syn.Z, ko, kname, 0.1, 1e-6)
    rhs_o = np.zeros((1, M))
    for s in range(0, ns, 16384):
        rhs_o += (ko.K(syn.Z, syn.X[s:s + 16384]) @ syn.y[s:s + 16384]).T
    assert _rel(rhs.cpu().numpy(), rhs_o) < 1e-11
    steps = 3
    sol, (k, err) = conjugate_gradient(op, rhs, None, 0.0, max_iterations=steps, max_steps_cycle=10 ** 6,
                                       check_every=steps)
    sol_o, (k_o, err_o) = ocg.conjugate_gradient(oop, rhs_o, np.zeros_like(rhs_o), 0.0, max_iterations=steps,
                                                 max_steps_cycle=10 ** 6)
    assert int(k) == steps == int(k_o)
    assert _rel(sol.cpu().numpy(), sol_o) < 1e-9, _rel(sol.cpu().numpy(), sol_o)
    assert abs(float(err) - float(err_o[0, 0])) / float(err_o[0, 0]) < 1e-8
    # full N: the recurrence residual equals the recomputed one, and it fell
    opf = SgprNormalOperator(kern, X, Z, 0.1, jitter=1e-6)
    rhsf = ops.kmn_matvec(kern.spec(D), X, Z, y).t().contiguous()
    solf, (kf, errf) = conjugate_gradient(opf, rhsf, None, 0.0, max_iterations=steps, max_steps_cycle=10 ** 6,
                                          check_every=steps)
    r = rhsf - opf.rmatmul(solf)
    true_half = 0.5 * float((r * r).sum())
    assert int(kf) == steps and abs(true_half - float(errf)) / true_half < 1e-8
    assert true_half < 0.5 * float((rhsf * rhsf).sum())


def test_c2_cdgp_end_to_end_elbo_and_full_covariance():
    """BASELINE.json configs[1] ("CDGP RBF N=100k, D=8, M=2048 ... CG") as one pipeline at its stated size:
    nearest-centre assignment and cluster statistics over all 100 000 rows against `oracle/cluster.py`, then at
    M = 2048 the exact-trace KL (`models.py:304-306`: an M-column CG), the ELBO of a 4096-row minibatch with the
    reference's scale N / batch (`models.py:125-134,163-169`), the full predictive covariance of a 128-row batch
    (`models.py:347-349`), and mean + variance of EVERY row through `predict_f_batched` -- against closed forms from
    the oracle's Cholesky twin (CGGP's KL omits 0.5 log|Kmm+Lambda|, `models.py:46,319`: twin KL minus that term)."""
    from cggp import kernels, synthetic
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.models import CGGP
    from cggp.optimize import assign_inducing_parameters, oips_update_inducing_parameters
    from oracle import cluster as oc
    N, D, M, dt, kname = synthetic.CONFIGS["C2"]
    syn = synthetic.make_inputs(N, D, M, dt)
    X, y, Z = (torch.from_numpy(a).to(dev()) for a in (syn.X, syn.y, syn.Z))
    kern = kernels.SquaredExponential(1.0, [1.0] * D)
    m = CGGP(kern, 0.1, Z, ConjugateGradient(1e-13, max_iterations=4 * M), num_probes=None, num_data=N)
    assign_inducing_parameters(m, *oips_update_inducing_parameters(m, (X, y), Z))
    idx = oc.nearest_centre_sqdist(syn.Z, syn.X)
    u, counts = oc.cluster_stats(idx, syn.y, M)
    assert np.array_equal(m.cluster_counts.cpu().numpy(), counts)
    assert np.max(np.abs(m.pseudo_u.cpu().numpy() - u)) < 1e-12
    ko = ok.Kernel(kname, 1.0, np.ones(D))
    twin = om.ClusterGP(ko, 0.1, syn.Z, pseudo_u=u, cluster_counts=counts, num_data=N)
    Kmm, KL = twin._KmmLambda()
    logdet = np.linalg.slogdet(KL)[1]
    kl_exact = twin.prior_kl() - 0.5 * logdet
    kl = m.prior_kl()
    assert abs(kl - kl_exact) / abs(kl_exact) < 1e-8, (kl, kl_exact)
    # minibatch ELBO, scale = N / 4096
    nb = 4096
    xb, yb = syn.X[:nb], syn.y[:nb]
    mu_c, var_c = twin.predict_f(xb)
    ve = om.gaussian_variational_expectations(mu_c, var_c, yb, twin.noise_variance)
    elbo_exact = np.sum(ve) * (N / nb) - kl_exact
    elbo = m.elbo((X[:nb], y[:nb]))
    assert abs(elbo - elbo_exact) / abs(elbo_exact) < 1e-7, (elbo, elbo_exact)
    # full covariance of a batch
    _, cov = m.predict_f(X[:128], full_cov=True)
    _, cov_c = twin.predict_f(syn.X[:128], full_cov=True)
    assert cov.shape == (1, 128, 128)
    assert float(np.max(np.abs(cov.cpu().numpy() - cov_c))) < 1e-7
    # every row, both forms of the batched prediction against each other and spot rows against the twin
    m.conjugate_gradient = ConjugateGradient(1e-6)
    mu_s, var_s = m.predict_f_batched(X, 8192, shared_inverse=True)
    assert mu_s.shape == (N, 1) and var_s.shape == (N, 1) and float(var_s.min()) > 0
    rows = np.r_[0, N - 1, np.random.default_rng(3).integers(0, N, 510)]
    mu_r, var_r = twin.predict_f(syn.X[rows])
    assert _rel(mu_s.cpu().numpy()[rows], mu_r) < 1e-3  # `a` is solved to the reference's 1e-6 threshold
    assert float(np.max(np.abs(var_s.cpu().numpy()[rows] - var_r))) < 1e-6
