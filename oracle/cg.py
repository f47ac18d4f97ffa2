"""Oracle: batched preconditioned conjugate gradient.

Restates `cggp/conjugate_gradient.py:24-212` in numpy.  Layout follows the
reference: the function-level solver works on row vectors (`rhs [Bt,n]`,
`p @ A`), the callable facade takes column layout (`rhs [n,Bt]`).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""

import numpy as np


class EyePreconditioner:
    """`cggp/conjugate_gradient.py:131-134`: z = vec, rz = sum(vec^2, -1, keepdims)."""

    def __call__(self, vec, mat):
        return vec, np.sum(np.square(vec), axis=-1, keepdims=True)


class JacobiPreconditioner:
    """Build-side addition (not in the reference): z = vec / diag(A)."""

    def __call__(self, vec, mat):
        z = vec / np.diagonal(mat)[None, :]
        return z, np.sum(z * vec, axis=-1, keepdims=True)


class BlockPreconditioner:
    """Block-Jacobi with the reference's constructor (`conjugate_gradient.py:137-157`).

    The reference version gathers `vec` on the batch axis and never scatters the
    result back (shape-inconsistent with CG1, zero call sites, untested:
    SURVEY §8a row CG4), so its behaviour is parity-unpinned.  This is the
    *intended* operation: per index block solve A[idx,idx] z[idx] = vec[idx]
    by Cholesky; indices not covered by any block pass through unchanged.
    """

    def __init__(self, block_indices):
        self.block_indices = np.asarray(block_indices, dtype=np.int64)

    def __call__(self, vec, mat):
        z = np.array(vec, copy=True)
        for idx in self.block_indices:
            A = mat[np.ix_(idx, idx)]
            L = np.linalg.cholesky(A)
            b = vec[:, idx].T  # [bs, Bt]
            y = np.linalg.solve(L, b)
            z[:, idx] = np.linalg.solve(L.T, y).T
        return z, np.sum(z * vec, axis=-1, keepdims=True)


class DensePreconditioner:
    """Build-side addition (not in the reference): z = vec @ Pinv for a symmetric positive
    definite Pinv [n, n] (e.g. the inverse of a cheap approximation of A)."""

    def __init__(self, inverse):
        self.inverse = np.asarray(inverse)

    def __call__(self, vec, mat):
        z = vec @ self.inverse
        return z, np.sum(z * vec, axis=-1, keepdims=True)


def _matmul_right(p, A):
    """`state.p @ A` (`conjugate_gradient.py:65`); A may be an operator with .rmatmul."""
    if hasattr(A, "rmatmul"):
        return A.rmatmul(p)
    return p @ A


def conjugate_gradient(
    matrix,
    rhs,
    initial_solution,
    error_threshold,
    preconditioner=None,
    max_iterations=None,
    max_steps_cycle=100,
    min_float=1e-16,
):
    """`cggp/conjugate_gradient.py:24-122` forward pass.

    Returns (solution [Bt,n], (steps:int, error [Bt,1] = 0.5*rz_final)).
    `min_float` is the reference's hard-coded breakdown guard (:50); it is a parameter
    here only so tests can separate the model algebra from the guard's stagnation floor
    (with the reference value CG cannot push 0.5||r||^2 much below ~1e-17).
    """
    if preconditioner is None:  # :44-45
        preconditioner = EyePreconditioner()
    A = matrix
    if max_iterations is None:  # :47-48
        max_iterations = A.shape[0]
    dtype = np.asarray(initial_solution).dtype
    min_float = dtype.type(min_float)  # :50 (1e-16 rounded to the solve dtype)
    zero = dtype.type(0.0)
    half = dtype.type(0.5)
    thr = dtype.type(error_threshold)

    b = np.asarray(rhs, dtype=dtype)
    v = np.array(initial_solution, dtype=dtype, copy=True)

    r = b - _matmul_right(v, A)  # :87-88
    z, rz = preconditioner(r, A)  # :89
    p = z  # :90
    i = 0  # :91

    def stopping_condition(r, i):  # :59-62
        norm_r_sq = np.sum(np.square(r), axis=-1, keepdims=True)
        over_threshold = np.any(half * norm_r_sq > thr)
        return bool(over_threshold) and (i < max_iterations)

    while stopping_condition(r, i):  # :93-95
        pA = _matmul_right(p, A)  # :65
        denom = np.sum(p * pA, axis=-1, keepdims=True)  # :66
        with np.errstate(divide="ignore", invalid="ignore"):
            gamma = rz / denom  # :67
        gamma = np.where(denom <= min_float, zero, gamma)  # :68
        v = v + gamma * p  # :69
        reset = (i % max_steps_cycle) == (max_steps_cycle - 1)  # :71
        if reset:  # :72-76
            r = b - _matmul_right(v, A)
        else:
            r = r - gamma * pA
        z, new_rz = preconditioner(r, A)  # :77
        with np.errstate(divide="ignore", invalid="ignore"):
            z_update = p * new_rz / rz  # :78
        z_update = np.where(rz <= min_float, zero, z_update)  # :79
        p = z if reset else z + z_update  # :80-84
        rz = new_rz
        i = i + 1  # :70

    return v, (i, half * rz)  # :96-98,120


def conjugate_gradient_vjp(matrix, solution, dx, error_threshold, preconditioner=None,
                           max_iterations=None, max_steps_cycle=100, min_float=1e-16):
    """Backward of CG1 (`conjugate_gradient.py:100-118`).

    db = CG(A, dx) from a zero start with the same stopping rule;
    dA = -solution^T @ db; no gradient to the initial solution.
    """
    db, _ = conjugate_gradient(
        matrix, dx, np.zeros_like(dx), error_threshold, preconditioner,
        max_iterations, max_steps_cycle, min_float,
    )
    dA = -solution.T @ db
    return dA, db


class ConjugateGradient:
    """Callable facade `cggp/conjugate_gradient.py:160-212` (column layout, drops stats)."""

    def __init__(self, error_threshold, preconditioner=None, max_iterations=None,
                 max_steps_cycle=None, min_float=1e-16):
        self.min_float = min_float
        self.error_threshold = error_threshold
        if preconditioner is None:
            preconditioner = EyePreconditioner()
        self.preconditioner = preconditioner
        self.max_iterations = max_iterations
        self.max_steps_cycle = max_steps_cycle

    def solve_with_stats(self, matrix, rhs, initial_solution=None):
        """As `paper_condition_wasserstein.py:262-294` (`stats_conjugate_gradient`)."""
        rhs = np.asarray(rhs).T  # :183
        if initial_solution is None:
            initial_solution = np.zeros_like(rhs)  # :185-186
        else:
            initial_solution = np.asarray(initial_solution).T  # :188
        max_iterations = self.max_iterations
        if max_iterations is None:
            max_iterations = matrix.shape[-1]  # :190-192
        max_steps_cycle = self.max_steps_cycle
        if max_steps_cycle is None:
            max_steps_cycle = max_iterations + 1  # :194-196
        solution, stats = conjugate_gradient(
            matrix, rhs, initial_solution, self.error_threshold,
            preconditioner=self.preconditioner, max_iterations=max_iterations,
            max_steps_cycle=max_steps_cycle, min_float=self.min_float,
        )
        return solution.T, stats  # :211

    def __call__(self, matrix, rhs, initial_solution=None):
        return self.solve_with_stats(matrix, rhs, initial_solution)[0]
