"""Generates tests/golden/*.npz from the CPU oracle (run: python tests/golden/make_golden.py).

The reference ships no golden vectors and cannot be run here (TensorFlow/GPflow absent), so these
fixtures are outputs of the oracle restatement -- data only: seeded inputs and expected outputs.
They travel to the GPU box (which has no /root/reference) and pin both the oracle against
regressions (tests/test_golden.py, CPU) and libmgp against the oracle (GPU).
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import cg as ocg, cluster as oc, kernels as ok, models as om  # noqa: E402


def case(name, D, N, M, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    y = np.sin(X).sum(1, keepdims=True) / np.sqrt(D) + np.sqrt(0.1) * rng.standard_normal((N, 1))
    Z = X[rng.choice(N, M, replace=False)]
    ls = rng.random(D) + 0.6
    var = 1.3
    kern = ok.Kernel(name, var, ls)
    V = rng.standard_normal((M, 3))
    W = rng.standard_normal((N, 2))
    idx = oc.nearest_centre_sqdist(Z, X)
    u, counts = oc.cluster_stats(idx, y, M)
    Knm = kern.K(X, Z)
    cg = ocg.ConjugateGradient(1e-15, max_iterations=4000)
    model = om.CGGP(kern, 0.1, Z, cg, num_probes=None, pseudo_u=u, cluster_counts=counts)
    Xs = X[:64] + 0.05
    mu, fvar = model.predict_f(Xs)
    probes = om.rademacher(M, 5, seed=4)
    model5 = om.CGGP(kern, 0.1, Z, cg, num_probes=5, pseudo_u=u, cluster_counts=counts)
    KL = om.add_diagonal(ok.Kuu(Z, kern), (0.1 / counts)[:, 0])
    rhs = rng.standard_normal((M, 2))
    sol8, (_, err8) = ocg.ConjugateGradient(0.0, max_iterations=8).solve_with_stats(KL, rhs)
    sgpr = om.SGPR((X, y), kern, Z, 0.1, jitter=1e-6)
    smu, svar = sgpr.predict_f(Xs)
    return dict(
        X=X, y=y, Z=Z, lengthscales=ls, variance=np.array(var), noise=np.array(0.1), V=V, W=W,
        knm_v=Knm @ V, kmn_w=Knm.T @ W, kuu_lambda=KL, kmn_knm=Knm.T @ Knm,
        idx=idx, pseudo_u=u, counts=counts, Xs=Xs, cggp_mu=mu, cggp_var=fvar,
        prior_kl_exact=np.array(model.prior_kl()), probes=probes,
        prior_kl_probes=np.array(model5.prior_kl(probes=probes)),
        cg_rhs=rhs, cg_sol_8steps=sol8, cg_err_8steps=err8,
        logdet_grad_probes=om.eval_logdet_grad(KL, cg, 1.0, probes=probes),
        sgpr_mu=smu, sgpr_var=svar, sgpr_elbo=np.array(sgpr.elbo()),
    )


if __name__ == "__main__":
    for name, D, N, M, seed in [("se", 1, 128, 16, 0), ("se", 8, 256, 32, 1), ("matern32", 2, 200, 24, 2),
                                ("matern52", 3, 160, 20, 3), ("matern12", 2, 150, 16, 4)]:
        np.savez_compressed(os.path.join(HERE, f"{name}_D{D}_N{N}_M{M}.npz"), **case(name, D, N, M, seed))
        print("wrote", name, D, N, M)
