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

from oracle import cg as ocg, cluster as oc, covertree as oct_, kernels as ok, models as om  # noqa: E402


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


def extras(seed=7):
    """Build-side additions and the F3 cover tree: a preconditioned CG trajectory and a cover tree."""
    rng = np.random.default_rng(seed)
    n = 48
    Xa = rng.standard_normal((n, 2))
    A = om.add_diagonal(ok.Kernel("se", 1.1, np.array([0.8, 1.2])).K(Xa), 0.05 * np.ones(n))
    E = rng.standard_normal((n, 24))  # rank 24 of 48: far from converged after a few steps
    Pinv = np.linalg.inv(A + 0.1 * E @ E.T)
    Pinv = 0.5 * (Pinv + Pinv.T)
    rhs = rng.standard_normal((n, 3))
    sol4, (_, err4) = ocg.ConjugateGradient(0.0, preconditioner=ocg.DensePreconditioner(Pinv),
                                            max_iterations=4).solve_with_stats(A, rhs)
    x = rng.uniform(-3, 3, (600, 2))
    y = np.sin(x.sum(1, keepdims=True)) + 0.05 * rng.standard_normal((600, 1))
    tree = oct_.CoverTree((x, y), spatial_resolution=0.6)
    means, counts = tree.cluster_mean_and_counts
    leaf_of_row = np.empty(600, dtype=np.int64)
    for k, rows in enumerate(tree.cluster_rows):
        leaf_of_row[rows] = k
    return dict(pcg_A=A, pcg_Pinv=Pinv, pcg_rhs=rhs, pcg_sol_4steps=sol4, pcg_err_4steps=err4,
                ct_x=x, ct_y=y, ct_resolution=np.array(0.6), ct_level_sizes=np.array([len(lv) for lv in tree.levels]),
                ct_centroids=tree.centroids, ct_means=means, ct_counts=counts, ct_leaf_of_row=leaf_of_row)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extras":
        np.savez_compressed(os.path.join(HERE, "extras_seed7.npz"), **extras())
        print("wrote extras")
        sys.exit(0)
    for name, D, N, M, seed in [("se", 1, 128, 16, 0), ("se", 8, 256, 32, 1), ("matern32", 2, 200, 24, 2),
                                ("matern52", 3, 160, 20, 3), ("matern12", 2, 150, 16, 4)]:
        np.savez_compressed(os.path.join(HERE, f"{name}_D{D}_N{N}_M{M}.npz"), **case(name, D, N, M, seed))
        print("wrote", name, D, N, M)
