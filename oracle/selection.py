"""Oracle: inducing-point selection -- restates `cggp/selection.py:35-153`.

Random draws are injected (initial centroids, permutation, indices) so that results can be
compared across implementations.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""

import numpy as np

from .cluster import nearest_centre
from .distance import euclid_distance


def kmeans_lloyd(points, k_centroids, threshold=1e-5, initial_centroids=None, distance_fn=None, max_loops=10000):
    """`selection.py:35-73`: Lloyd iterations until the mean point-to-centre distance stops
    improving by more than `threshold`; empty clusters keep a zero centroid (count clipped to 1)."""
    if distance_fn is None:
        distance_fn = euclid_distance
    points = np.asarray(points)

    def body(centroids):
        idx, dist = nearest_centre(centroids, points, distance_fn)  # :48-50
        counts = np.maximum(np.bincount(idx, minlength=k_centroids), 1).astype(points.dtype)[:, None]  # :52-56
        sums = np.zeros((k_centroids, points.shape[1]), points.dtype)
        np.add.at(sums, idx, points)  # :58-63
        return sums / counts, float(np.mean(dist))

    centroids, mean_d = body(np.asarray(initial_centroids))  # :70
    prev = np.inf
    loops = 1
    while prev - mean_d > threshold and loops < max_loops:  # :44-45, :71
        new_centroids, new_mean = body(centroids)
        centroids, prev, mean_d = new_centroids, mean_d, new_mean
        loops += 1
    return centroids, mean_d


def oips(kernel, inputs, rho, max_points):
    """`selection.py:76-103`: scan the inputs in order, keep a point when its largest covariance
    with the kept set is below rho * k(x,x).  (Line :82 of the reference is a no-op concat.)"""
    inputs = np.asarray(inputs)
    n = inputs.shape[0]
    kxx = kernel.K_diag(inputs)
    i0 = int(np.argmax(kxx))  # :79
    sel = [i0]
    i, j = 1, 1  # :96-97
    while i < n and j < max_points:  # :85-86
        kix = kernel.K(inputs[i:i + 1], inputs[sel])  # :90
        if np.max(kix) < rho * kxx[i]:  # :91-93
            sel.append(i)
            j += 1
        i += 1
    sel = np.asarray(sel)
    return inputs[sel], sel


def greedy_selection(kernel, inputs, max_points, perm):
    """`selection.py:113-153` (conditional-variance / pivoted-Cholesky selection); `perm` is the
    shuffle the reference draws with tf.random.shuffle."""
    inputs = np.asarray(inputs)
    n = inputs.shape[0]
    m = min(n, max_points)
    X = inputs[perm]
    di = kernel.K_diag(X).copy()
    inds = [int(np.argmax(di))]
    ci = np.zeros((1, n), inputs.dtype)
    cur = 1
    while cur < m:
        j = inds[-1]
        dj = np.sqrt(di[j])
        cj = ci[:cur, j:j + 1]
        K = kernel.K(X, X[j:j + 1])
        ei = (K - ci.T @ cj) / dj
        ci = np.concatenate([ci, ei.T], axis=0)
        di = di - np.square(ei)[:, 0]
        inds.append(int(np.argmax(di)))
        cur += 1
    perm_inds = np.asarray(perm)[np.asarray(inds)]
    return inputs[perm_inds], perm_inds
