"""CPU baseline for bench.py's `cpu_baseline` leg (kind "port").  TEST INFRASTRUCTURE ONLY.

The reference's TensorFlow/GPflow CPU path cannot be installed here, so the timed
baseline is this port of its *dense* algorithm: per row chunk build K_chunk = k(X_chunk, Z)
through GPflow's `square_distance` expansion + exp (gpflow/utilities/ops.py,
gpflow/kernels/stationaries.py), then GEMV/GEMM against it -- exactly what
`cggp/models.py:334,351` (`Kuf` then `matmul`) and the dense `p @ A` of
`cggp/conjugate_gradient.py:65` do -- in torch-CPU with every host core
(the cores this process may use: cgroup quota / affinity), as BASELINE.md §3 prescribes.
"""

import time

import numpy as np
import torch


def _k_chunk(Xc, Zs, Z2, variance, name):
    r2 = (Xc * Xc).sum(1, keepdim=True) + Z2[None, :] - 2.0 * (Xc @ Zs.T)
    if name == "se":
        return variance * torch.exp(-0.5 * r2)
    r = torch.sqrt(torch.clamp(r2, min=1e-36))
    if name == "matern12":
        return variance * torch.exp(-r)
    if name == "matern32":
        s3 = 3.0 ** 0.5
        return variance * (1.0 + s3 * r) * torch.exp(-s3 * r)
    s5 = 5.0 ** 0.5
    return variance * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * torch.exp(-s5 * r)


def sgpr_operator_apply(X, Z, v, Kmm, s2, variance, lengthscales, name="se", chunk=16384):
    """S v = s2 Kmm v + K_mn (K_nm v), K built per row chunk (dense path)."""
    Zs = Z / lengthscales
    Z2 = (Zs * Zs).sum(1)
    out = s2 * (Kmm @ v)
    for s in range(0, X.shape[0], chunk):
        Xc = X[s:s + chunk] / lengthscales
        K = _k_chunk(Xc, Zs, Z2, variance, name)
        out += K.T @ (K @ v)
    return out


def time_cg_iteration(X, Z, variance, lengthscales, s2, name="se", dtype=torch.float64, repeats=2):
    """Seconds for ONE CG iteration of the SGPR normal-equation operator on the given rows
    (operator application + the vector updates of conjugate_gradient.py:66-84), best of repeats."""
    from .hostinfo import available_cores
    threads = available_cores()  # cgroup/affinity share of the box, not the raw socket count
    torch.set_num_threads(threads)
    X = torch.from_numpy(np.ascontiguousarray(X)).to(dtype)
    Z = torch.from_numpy(np.ascontiguousarray(Z)).to(dtype)
    ls = torch.as_tensor(lengthscales, dtype=dtype)
    M = Z.shape[0]
    Zs = Z / ls
    Kmm = _k_chunk(Zs, Zs, (Zs * Zs).sum(1), variance, name)
    p = torch.randn(M, 1, dtype=dtype, generator=torch.Generator().manual_seed(3))
    r = p.clone()
    v = torch.zeros_like(p)
    rz = (r * r).sum()
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        Ap = sgpr_operator_apply(X, Z, p, Kmm, s2, variance, ls, name)
        denom = (p * Ap).sum()
        gamma = rz / denom
        v = v + gamma * p
        r = r - gamma * Ap
        new_rz = (r * r).sum()
        p2 = r + p * new_rz / rz
        best = min(best, time.perf_counter() - t0)
        del p2
    return best, threads
