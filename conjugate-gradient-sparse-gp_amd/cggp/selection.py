"""Inducing-point selection (next row F3) -- host mirror of `cggp/selection.py`.

`kmeans_indices_and_distances`, `kmeans_lloyd`, `oips`, `uniform`, `greedy_selection` with the
reference's signatures (`cggp/selection.py:14-153`).  The N x M searches, the per-cluster sums and
the kernel columns run in libmgp (`nearest_center`, `cluster_stats`, `k_dense`); the sequential
selection logic stays on the host, as in the reference.  Random draws take a `seed`, or can be
injected (`initial_centroids`, `perm`, `indices`) -- TensorFlow's streams are not reproducible here.
The cover tree lives in `cggp.covertree` (host C++ in libmgp).
"""

import numpy as np
import torch

from . import ops

from .distance import resolve_distance


def kmeans_indices_and_distances(centroids, points, distance_fn=None, kernel=None):
    """`selection.py:14-32`: index of the nearest centroid and the distance to it, per point.
    `distance_fn` as in the reference: None, `euclid_distance` or a `create_distance_fn(...)` result
    (also accepted: the distance type as a string, with `kernel`)."""
    D = points.shape[1]
    dtype_name, kernel = resolve_distance(distance_fn, kernel)
    if kernel is None:
        from .kernels import SquaredExponential
        kernel = SquaredExponential(1.0, [1.0] * D)  # unused by the euclidean distances
    idx, dist = ops.nearest_center(kernel.spec(D), points, centroids, distance_type=dtype_name)
    return idx, dist


def kmeans_lloyd(points, k_centroids, threshold=1e-5, initial_centroids=None, distance_fn=None, kernel=None,
                 seed=0, max_loops=10000, distance_type=None):
    """`selection.py:35-73` (same positional order: points, k_centroids, threshold, initial_centroids,
    distance_fn).  `distance_type=` is the build's older spelling of a string `distance_fn`."""
    if distance_fn is None and distance_type is not None:
        distance_fn = distance_type
    if initial_centroids is None:  # :65-67
        g = torch.Generator().manual_seed(seed)
        initial_centroids = points[torch.randperm(points.shape[0], generator=g)[:k_centroids].to(points.device)]

    def body(centroids):
        idx, dist = kmeans_indices_and_distances(centroids, points, distance_fn, kernel)  # :48-50
        # per-cluster coordinate sums in one pass, deterministic order (:58-63)
        sums, counts = ops.cluster_stats(idx, points, k_centroids)
        if sums.dim() == 1:  # one input dimension
            sums = sums[:, None]
        counts = torch.clamp(counts, min=1.0)  # :55
        return sums / counts[:, None], float(dist.mean())

    centroids, mean_d = body(initial_centroids.contiguous())  # :70
    prev, loops = float("inf"), 1
    while prev - mean_d > threshold and loops < max_loops:  # :44-45,:71
        new_centroids, new_mean = body(centroids)
        centroids, prev, mean_d = new_centroids, mean_d, new_mean
        loops += 1
    return centroids, mean_d


def uniform(inputs, max_points, seed=0, indices=None):
    """`selection.py:106-110`: `max_points` indices drawn uniformly WITH replacement."""
    if indices is None:
        g = torch.Generator().manual_seed(seed)
        indices = torch.randint(0, inputs.shape[0], (max_points,), generator=g)
    indices = torch.as_tensor(indices).to(inputs.device)
    return inputs[indices], indices


def oips(kernel, inputs, rho, max_points, chunk=4096):
    """`selection.py:76-103`.  Exactly the reference's sequential scan, evaluated in blocks: a
    candidate whose largest covariance with the CURRENT set already reaches rho*k(x,x) can never
    be accepted later (the set only grows), so each block needs one device panel k(block, set) and
    a small host pass over the survivors."""
    n, D = inputs.shape
    spec = kernel.spec(D)
    var = kernel.variance  # k(x,x) of a stationary kernel; argmax of a constant is index 0 (:79)
    sel = [0]
    Zsel = inputs[0:1].clone()
    i = 1
    while i < n and len(sel) < max_points:
        hi = min(n, i + chunk)
        blk = inputs[i:hi]
        kmax = ops.k_dense(spec, blk, Zsel).max(dim=1).values.cpu().numpy()
        surv = np.nonzero(kmax < rho * var)[0]
        if surv.size:
            S = blk[torch.as_tensor(surv, device=inputs.device)]
            Kss = ops.k_dense(spec, S, S).cpu().numpy()
            accepted = []
            for a, s_idx in enumerate(surv):
                if len(sel) >= max_points:
                    break
                if accepted and np.max(Kss[a, accepted]) >= rho * var:
                    continue
                accepted.append(a)
                sel.append(i + int(s_idx))
            if accepted:
                Zsel = torch.cat([Zsel, S[torch.as_tensor(accepted, device=inputs.device)]], dim=0)
        i = hi
    idx = torch.as_tensor(sel, device=inputs.device)
    return inputs[idx], idx


def greedy_selection(kernel, inputs, max_points, seed=0, perm=None):
    """`selection.py:113-153`: greedy conditional-variance selection (pivoted Cholesky)."""
    n, D = inputs.shape
    m = min(n, max_points)
    if perm is None:
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
    perm = torch.as_tensor(perm).to(inputs.device)
    X = inputs[perm].contiguous()
    spec = kernel.spec(D)
    di = kernel.K_diag(X).clone()
    inds = [int(torch.argmax(di))]
    ci = torch.zeros((m, n), dtype=inputs.dtype, device=inputs.device)  # rows filled as selected
    cur = 1
    while cur < m:
        j = inds[-1]
        dj = torch.sqrt(di[j])
        cj = ci[:cur, j:j + 1]  # [cur, 1]
        K = ops.k_dense(spec, X, X[j:j + 1].contiguous())  # [n, 1]
        ei = (K - ci[:cur].t() @ cj) / dj
        ci[cur] = ei[:, 0]
        di = di - ei[:, 0] ** 2
        inds.append(int(torch.argmax(di)))
        cur += 1
    perm_inds = perm[torch.as_tensor(inds, device=inputs.device)]
    return inputs[perm_inds], perm_inds
