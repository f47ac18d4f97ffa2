"""Row F3: cover-tree clustering (`cggp/covertree.py:26-179`, `cggp/optimize.py:19-38`).

Host code (no GPU): libmgp's `mgp_covertree_build` against the numpy oracle on seeded inputs, the
hand-checkable cases that pin the oracle, and structural invariants at a size the oracle does not
reach.  The reference class cannot be imported here (TensorFlow), see oracle/covertree.py.
"""
import warnings

import numpy as np
import pytest

from oracle import covertree as oct_


def build(data, **kw):
    from cggp.covertree import CoverTree
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return CoverTree(None, data, **kw)


def test_hand_case_two_clusters():
    x = np.array([[0.0], [1.0], [10.0], [11.0]])
    y = np.array([[1.0], [3.0], [5.0], [9.0]])
    # root: mean 5.5, radius 5.5; level 1 radius 2.75: seed 0 takes {0,1}, seed 10 takes {10,11}
    for tree in (oct_.CoverTree((x, y), num_levels=2, lloyds=False, voronoi=False),
                 build((x, y), num_levels=2, lloyds=False, voronoi=False)):
        assert np.array_equal(tree.centroids, [[0.0], [10.0]])
        m, c = tree.cluster_mean_and_counts
        assert np.array_equal(m, [[2.0], [7.0]]) and np.array_equal(c, [[2.0], [2.0]])
        assert tree.levels[0][0].radius == 5.5 and tree.levels[1][0].radius == 2.75
    # Lloyd re-centring: the centre moves to the mean of the seed's ball
    for tree in (oct_.CoverTree((x, y), num_levels=2), build((x, y), num_levels=2)):
        assert np.array_equal(tree.centroids, [[0.5], [10.5]])
    # resolution 3: levels = ceil(log2(5.5/3)) + 1 = 2, root radius 3 * 2 = 6
    for tree in (oct_.CoverTree((x, y), spatial_resolution=3.0), build((x, y), spatial_resolution=3.0)):
        assert len(tree.levels) == 2 and tree.levels[0][0].radius == 6.0
        assert np.array_equal(tree.centroids, [[0.5], [10.5]])


def test_hand_case_lloyd_fallback_and_voronoi():
    x = np.array([[0.0], [2.0], [3.0], [4.0], [5.0], [10.0]])
    y = np.arange(6.0)[:, None]
    # root mean 4, radius 6 -> level 1 radius 3
    for mk in (lambda **k: oct_.CoverTree((x, y), **k), lambda **k: build((x, y), **k)):
        t = mk(num_levels=2, lloyds=True, voronoi=False)
        # seed 0: ball {0,2,3} (<= 3) -> mean 5/3 -> takes rows within 3: {0,2,3,4}; seed 5: ball {5} ->
        # centre 5 is 10/3 >= 3 from 5/3 -> kept, takes {5}; seed 10 alone
        assert np.allclose(t.centroids[:, 0], [5.0 / 3.0, 5.0, 10.0])
        assert [list(r) for r in (nd.rows for nd in t.levels[-1])] == [[0, 1, 2, 3], [4], [5]]
        t = mk(num_levels=2, lloyds=True, voronoi=True)
        # nearest centre: 4 is 2.33 from 5/3 and 1 from 5 -> moves to the second cluster
        assert [list(r) for r in (nd.rows for nd in t.levels[-1])] == [[0, 1, 2], [3, 4], [5]]
        m, c = t.cluster_mean_and_counts
        assert np.allclose(m[:, 0], [1.0, 3.5, 5.0]) and np.array_equal(c[:, 0], [3.0, 2.0, 1.0])


@pytest.mark.parametrize("seed,n,d,levels,lloyds,voronoi", [
    (0, 300, 2, 3, True, True), (1, 300, 2, 4, False, True), (2, 500, 3, 3, True, False),
    (3, 400, 1, 5, True, True), (4, 200, 8, 3, True, True), (5, 600, 2, 5, False, False),
])
def test_matches_oracle(seed, n, d, levels, lloyds, voronoi):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d))
    y = rng.standard_normal((n, 1))
    ref = oct_.CoverTree((x, y), num_levels=levels, lloyds=lloyds, voronoi=voronoi)
    got = build((x, y), num_levels=levels, lloyds=lloyds, voronoi=voronoi)
    assert [len(lv) for lv in got.levels] == [len(lv) for lv in ref.levels]
    for lg, lr in zip(got.levels, ref.levels):
        for a, b in zip(lg, lr):
            assert np.allclose(a.point, b.point, rtol=0, atol=1e-13)
            assert np.array_equal(a.rows, b.rows)
            assert a.radius == b.radius
    mg, cg = got.cluster_mean_and_counts
    mr, cr = ref.cluster_mean_and_counts
    assert np.array_equal(cg, cr)
    assert np.allclose(mg, mr, rtol=0, atol=1e-13, equal_nan=True)


def test_spatial_resolution_matches_oracle_and_update_fn():
    from cggp.optimize import covertree_update_inducing_parameters
    rng = np.random.default_rng(7)
    x = rng.uniform(-3, 3, (800, 2))
    y = np.sin(x.sum(1, keepdims=True))
    iv0, m0, c0 = oct_.covertree_update_inducing_parameters((x, y), 0.7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        iv, m, c = covertree_update_inducing_parameters(None, (x, y), None, 0.7)
    assert iv.shape == iv0.shape and np.allclose(iv, iv0, atol=1e-13)
    assert np.array_equal(c, c0) and np.allclose(m, m0, atol=1e-13)
    assert c.min() >= 1 and c.sum() == 800


@pytest.mark.parametrize("voronoi", [False, True])
def test_invariants_at_size(voronoi):
    """20k rows (the oracle's per-node numpy passes would take minutes): the finest level
    partitions the rows (coarser levels have handed theirs down); every row lies within the level
    radius of its centre; with Voronoi on, no level-1 centre is closer than the assigned one."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((20000, 3))
    y = rng.standard_normal((20000, 1))
    t = build((x, y), spatial_resolution=0.4, voronoi=voronoi)
    assert len(t.levels) >= 4 and all(len(a) <= len(b) for a, b in zip(t.levels, t.levels[1:]))
    leaves = t.levels[-1]
    assert np.array_equal(np.sort(np.concatenate([nd.rows for nd in leaves])), np.arange(20000))
    assert all(nd.rows.size == 0 for lv in t.levels[:-1] for nd in lv)
    for nd in leaves[:: max(1, len(leaves) // 200)]:
        if nd.rows.size:
            assert np.max(np.linalg.norm(x[nd.rows] - nd.point, axis=1)) <= nd.radius * (1 + 1e-12)
        assert nd.parent is not None and nd in nd.parent.children
        assert np.linalg.norm(nd.point - nd.parent.point) <= nd.parent.radius * (1 + 1e-12)
    m, c = t.cluster_mean_and_counts
    assert c.sum() == 20000 and np.isnan(m[c == 0]).all() and np.isfinite(m[c > 0]).all()
    if voronoi:
        t2 = build((x, y), num_levels=2, voronoi=True)
        pts = np.stack([nd.point for nd in t2.levels[1]])
        for k, nd in enumerate(t2.levels[1]):
            d = np.linalg.norm(x[nd.rows][:, None, :] - pts[None], axis=-1)
            assert np.all(np.argmin(d, axis=1) == k)


def test_errors_and_dtypes():
    x = np.zeros((5, 2))
    with pytest.raises(RuntimeError):
        build((x, np.zeros((5, 1))), spatial_resolution=1.0)  # all rows coincide
    with pytest.raises(ValueError):
        build((x, np.zeros((4, 1))), num_levels=2)
    rng = np.random.default_rng(0)
    x32 = rng.standard_normal((100, 2)).astype(np.float32)
    y32 = rng.standard_normal((100, 1)).astype(np.float32)
    t = build((x32, y32), num_levels=3)
    m, c = t.cluster_mean_and_counts
    assert t.centroids.dtype == np.float32 and m.dtype == np.float32 and c.dtype == np.float32
    import torch
    tt = build((torch.from_numpy(x32), torch.from_numpy(y32)), num_levels=3)
    assert np.array_equal(tt.centroids, t.centroids)
