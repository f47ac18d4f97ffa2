"""`cggp/utils.py:11-17`."""

import torch


def add_diagonal(matrix, diagonal):
    """Returns `matrix + diag(diagonal)` for a [n,n] matrix and a length-n vector (not in place)."""
    if matrix.dim() != 2 or matrix.shape[0] != matrix.shape[1]:
        raise ValueError("matrix must be [n, n]")
    diagonal = diagonal.reshape(-1)
    if diagonal.shape[0] != matrix.shape[0]:
        raise ValueError("diagonal must have n entries")
    out = matrix.clone()
    out.diagonal().add_(diagonal.to(out.dtype))
    return out
