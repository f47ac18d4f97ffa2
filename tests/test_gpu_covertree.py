"""Row F3 on the GPU: `mgp_covertree_build_device` (csrc/covertree_dev.hip) -- the sequential acceptance of centres on
the host, the all-pairs-shaped passes as device filters -- must give the tree `mgp_covertree_build` gives, node for
node and bit for bit (the host construction is pinned to the oracle in tests/test_covertree.py), and through it the
oracle's."""
import time
import warnings

import numpy as np
import pytest
import torch

from oracle import covertree as oct_

pytestmark = pytest.mark.gpu


def build(data, device, **kw):
    from cggp.covertree import CoverTree
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return CoverTree(None, data, device=device, **kw)


def same_tree(a, b):
    assert [len(lv) for lv in a.levels] == [len(lv) for lv in b.levels]
    for la, lb in zip(a.levels, b.levels):
        for na, nb in zip(la, lb):
            assert np.array_equal(na.point, nb.point) and na.radius == nb.radius
            assert np.array_equal(na.rows, nb.rows)
            assert (na.parent is None) == (nb.parent is None)
            if na.parent is not None:
                assert np.array_equal(na.parent.point, nb.parent.point)


@pytest.mark.parametrize("seed,n,d,kw", [
    (0, 300, 2, dict(num_levels=3)), (1, 300, 2, dict(num_levels=4, lloyds=False)),
    (2, 500, 3, dict(num_levels=3, voronoi=False)), (3, 400, 1, dict(num_levels=5)),
    (4, 2000, 8, dict(num_levels=3)), (5, 600, 2, dict(num_levels=5, lloyds=False, voronoi=False)),
    (6, 5000, 2, dict(spatial_resolution=0.15)), (7, 3000, 5, dict(spatial_resolution=0.9)),
    (8, 1500, 77, dict(num_levels=3)), (9, 1, 3, dict(num_levels=2)),
])
def test_device_construction_equals_host_construction(seed, n, d, kw):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d))
    if n > 10:
        x[7] = x[3]  # coincident rows
    y = rng.standard_normal((n, 1))
    host = build((x, y), False, **kw)
    dev = build((x, y), None, **kw)
    assert host.built_on == "host" and dev.built_on == "device"
    same_tree(dev, host)
    mh, ch = host.cluster_mean_and_counts
    md, cd = dev.cluster_mean_and_counts
    assert np.array_equal(ch, cd) and np.array_equal(mh, md, equal_nan=True)
    # ... from a device tensor, and through the update function
    dev2 = build((torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()), None, **kw)
    same_tree(dev2, host)


def test_device_construction_matches_the_oracle():
    rng = np.random.default_rng(21)
    x = rng.standard_normal((400, 3))
    y = rng.standard_normal((400, 1))
    ref = oct_.CoverTree((x, y), num_levels=4)
    got = build((x, y), None, num_levels=4)
    assert [len(lv) for lv in got.levels] == [len(lv) for lv in ref.levels]
    for lg, lr in zip(got.levels, ref.levels):
        for a, b in zip(lg, lr):
            assert np.allclose(a.point, b.point, rtol=0, atol=1e-13) and np.array_equal(a.rows, b.rows)


def test_hand_cases_on_the_device():
    x = np.array([[0.0], [2.0], [3.0], [4.0], [5.0], [10.0]])
    y = np.arange(6.0)[:, None]
    t = build((x, y), None, num_levels=2, lloyds=True, voronoi=True)
    assert [list(nd.rows) for nd in t.levels[-1]] == [[0, 1, 2], [3, 4], [5]]
    with pytest.raises(RuntimeError):
        build((np.zeros((5, 2)), np.zeros((5, 1))), None, spatial_resolution=1.0)  # all rows coincide


def test_realistic_dimension_is_fast_and_partitions_the_rows():
    """D = 8, where the host construction degenerates to all-pairs (75 s for 2e5 rows on one core):
    60 000 rows here, a few seconds on the device path; structure checked, equality with the host on a 6 000-row cut."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((60000, 8))
    y = np.sin(x[:, :1])
    t0 = time.perf_counter()
    t = build((x, y), None, spatial_resolution=1.0)
    dt = time.perf_counter() - t0
    leaves = t.levels[-1]
    assert np.array_equal(np.sort(np.concatenate([nd.rows for nd in leaves])), np.arange(60000))
    for nd in leaves[:: max(1, len(leaves) // 100)]:
        if nd.rows.size:
            assert np.max(np.linalg.norm(x[nd.rows] - nd.point, axis=1)) <= nd.radius * 2.0 + 1e-12
    print(f"cover tree, 60000 x 8, resolution 1.0: {len(leaves)} leaves, {len(t.levels)} levels, {dt:.2f} s")
    assert dt < 60.0
    same_tree(build((x[:6000], y[:6000]), None, spatial_resolution=1.0), build((x[:6000], y[:6000]), False,
                                                                              spatial_resolution=1.0))
