"""Next row F3: inducing-point selection -- oracle on the CPU, libmgp-backed version on the GPU."""

import numpy as np
import pytest
import torch

from oracle import distance as od, kernels as ok, selection as osel


def _data(n=600, d=2, seed=0):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.standard_normal((n // 2, d)) - 2.0, rng.standard_normal((n - n // 2, d)) + 2.0])


def test_oracle_kmeans_lloyd_converges_and_oips_threshold():
    X = _data()
    c0 = X[[0, 300, 5, 400]]
    C, md = osel.kmeans_lloyd(X, 4, 1e-6, c0)
    assert C.shape == (4, 2) and md < 1.5
    _, d0 = __import__("oracle").cluster.nearest_centre(c0, X, od.euclid_distance)
    assert md <= np.mean(d0)  # Lloyd never worsens the objective it monitors
    kern = ok.Kernel("se", 1.0, [1.0, 1.0])
    Z, idx = osel.oips(kern, X, 0.5, 50)
    K = kern.K(Z)
    assert idx[0] == 0 and np.all(np.diff(idx) > 0)
    assert np.max(K - np.eye(len(idx))) < 0.5  # kept points are mutually below the threshold
    Zg, ig = osel.greedy_selection(kern, X, 20, np.arange(len(X)))
    assert len(set(ig.tolist())) == 20


@pytest.mark.gpu
def test_selection_matches_oracle_on_gpu():
    from cggp import kernels, selection
    dev = torch.device("cuda:0")
    X = _data(800, 3, seed=1)
    Xt = torch.from_numpy(X).to(dev)
    # k-means Lloyd from the same initial centroids
    c0 = X[[0, 500, 7, 650, 30]]
    C0, md0 = osel.kmeans_lloyd(X, 5, 1e-7, c0)
    C, md = selection.kmeans_lloyd(Xt, 5, 1e-7, torch.from_numpy(c0).to(dev))
    assert np.max(np.abs(C.cpu().numpy() - C0)) < 1e-10 and abs(md - md0) < 1e-10
    idx, dist = selection.kmeans_indices_and_distances(C, Xt)
    assert idx.shape == (800,) and abs(float(dist.mean()) - md) < 1e-3  # md is measured one update earlier
    # OIPS: identical index sequence (integer work: exact)
    for name, rho, cap in [("se", 0.6, 60), ("matern32", 0.4, 25), ("se", 0.9, 1000)]:
        ko = ok.Kernel(name, 1.3, [0.8, 1.1, 0.9])
        kg = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32}[name](1.3, [0.8, 1.1, 0.9])
        Z0, i0 = osel.oips(ko, X, rho, cap)
        Z, i = selection.oips(kg, Xt, rho, cap, chunk=128)
        assert np.array_equal(i.cpu().numpy(), i0) and np.array_equal(Z.cpu().numpy(), Z0)
    # greedy conditional-variance selection with the same shuffle
    perm = np.random.default_rng(3).permutation(800)
    Zg0, ig0 = osel.greedy_selection(ko, X, 30, perm)
    Zg, ig = selection.greedy_selection(kg, Xt, 30, perm=torch.from_numpy(perm))
    assert np.array_equal(ig.cpu().numpy(), ig0)
    # uniform: with replacement, injected or seeded
    S, ind = selection.uniform(Xt, 17, indices=[5, 5, 9])
    assert S.shape == (3, 3) and torch.equal(S[0], S[1])
    S2, ind2 = selection.uniform(Xt, 17, seed=1)
    assert S2.shape == (17, 3) and int(ind2.max()) < 800
