"""Oracle in extended precision (numpy.longdouble: x87 80-bit, eps 1.1e-19).  TEST INFRASTRUCTURE ONLY.

The SGPR predictive variance of GPflow (`gpflow.models.SGPR.predict_f`, reached through
`sgpr_class`, `cggp/cli_utils.py:444-446`) subtracts two O(1) quantities that both pass through
`L = chol(Kmm + jitter I)` -- a matrix of condition ~ M / jitter.  In fp64 the result is therefore only
determined to about cond * eps relative to k**, i.e. 1e-5..1e-4 of a variance of 1e-3.  This module
restates the same two-Cholesky algorithm with every number in longdouble (LAPACK has no such type, so
Cholesky and the triangular solves are written out, vectorised over columns), so that tests can
measure how far the fp64 oracle itself is from the exactly-rounded answer and hold the HIP path to that
bound instead of to a tolerance neither fp64 implementation can meet.
"""

import numpy as np

LD = np.longdouble


def k_se(X, X2, variance, lengthscales):
    """GPflow SquaredExponential through the square_distance expansion, in longdouble."""
    a = np.asarray(X, LD) / np.asarray(lengthscales, LD)
    b = np.asarray(X2, LD) / np.asarray(lengthscales, LD)
    r2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - LD(2) * (a @ b.T)
    return LD(variance) * np.exp(LD(-0.5) * r2)


def cholesky(A):
    A = np.array(A, LD, copy=True)
    n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        d = A[j, j] - np.dot(L[j, :j], L[j, :j])
        if not d > 0:
            raise np.linalg.LinAlgError("matrix is not positive definite in longdouble")
        L[j, j] = np.sqrt(d)
        if j + 1 < n:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def solve_lower(L, B):
    """L^-1 B for lower-triangular L, all columns of B at once."""
    B = np.array(B, LD, copy=True)
    n = L.shape[0]
    for i in range(n):
        if i:
            B[i] -= L[i, :i] @ B[:i]
        B[i] /= L[i, i]
    return B


def sgpr_predict_se(X, Y, Z, Xnew, variance, lengthscales, noise_variance, jitter):
    """`oracle.models.SGPR.predict_f` (GPflow SGPR, SE kernel, zero mean) in longdouble -> (mean, var)."""
    kuf = k_se(Z, X, variance, lengthscales)
    kuu = k_se(Z, Z, variance, lengthscales) + LD(jitter) * np.eye(len(Z), dtype=LD)
    sigma = np.sqrt(LD(noise_variance))
    L = cholesky(kuu)
    A = solve_lower(L, kuf) / sigma
    B = A @ A.T + np.eye(len(Z), dtype=LD)
    LB = cholesky(B)
    c = solve_lower(LB, A @ np.asarray(Y, LD)) / sigma
    Kus = k_se(Z, Xnew, variance, lengthscales)
    tmp1 = solve_lower(L, Kus)
    tmp2 = solve_lower(LB, tmp1)
    mean = tmp2.T @ c
    var = (LD(variance) + (tmp2 * tmp2).sum(0) - (tmp1 * tmp1).sum(0))[:, None]
    return mean, var
