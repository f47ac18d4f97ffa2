"""Tensor-level wrappers over the C ABI (include/mgp.h).  No arithmetic happens here.

Every function takes CUDA(HIP) torch tensors, validates shapes on the host (the kernels
assume them) and enqueues libmgp work on torch's current stream.
"""

import ctypes

import torch

from . import _hip
from ._hip import COLS, ROWS  # noqa: F401


class KernelSpec:
    """Host-side kernel hyper-parameters in the form libmgp takes them (row K2)."""

    def __init__(self, kind, variance, lengthscales, D):
        self.kind = kind
        self.variance = float(variance)
        ls = [float(x) for x in (lengthscales if hasattr(lengthscales, "__len__") else [lengthscales])]
        self.lengthscales = ls * D if len(ls) == 1 else ls
        self.D = int(D)
        if len(self.lengthscales) != self.D:
            raise ValueError(f"lengthscales has {len(self.lengthscales)} entries, inputs have D={self.D}")

    def struct(self, dtype_c):
        return _hip.make_kernel_struct(self.kind, dtype_c, self.D, self.variance, self.lengthscales)


def _points(t, name, D=None, dtype=None):
    t = _hip.check_tensor(t, name, dtype=dtype)
    if t.dim() != 2:
        raise ValueError(f"{name} must be [n, D], got {tuple(t.shape)}")
    if D is not None and t.shape[1] != D:
        raise ValueError(f"{name} has D={t.shape[1]}, expected {D}")
    return t


def knm_matvec(spec, X, Z, V, v_layout=COLS, out_layout=None):
    """out = k(X, Z) @ V.  V [M,R] (COLS) or [R,M] (ROWS); out [N,R] or [R,N]."""
    X = _points(X, "X", spec.D)
    Z = _points(Z, "Z", spec.D, X.dtype)
    V = _hip.check_tensor(V, "V", dtype=X.dtype)
    if V.dim() != 2:
        raise ValueError("V must be 2-D")
    M = Z.shape[0]
    R = V.shape[1] if v_layout == COLS else V.shape[0]
    if (V.shape[0] if v_layout == COLS else V.shape[1]) != M:
        raise ValueError(f"V shape {tuple(V.shape)} does not match M={M}")
    out_layout = v_layout if out_layout is None else out_layout
    N = X.shape[0]
    out = torch.empty((N, R) if out_layout == COLS else (R, N), dtype=X.dtype, device=X.device)
    if N == 0 or R == 0:
        return out
    # M == 0 (empty inducing set) goes through the C-ABI too: libmgp writes the zeros
    hd = _hip.get_handle(X.device)
    k = spec.struct(_hip.dtype_code(X))
    hd.check(hd.lib.mgp_knm_matvec(hd.h, ctypes.byref(k), _hip.ptr(X), N, _hip.ptr(Z), M, _hip.ptr(V), R,
                                   v_layout, _hip.ptr(out), out_layout))
    return out


def kmn_matvec(spec, X, Z, W, w_layout=COLS, out_layout=None):
    """out = k(Z, X) @ W = K_nm^T W.  W [N,R] (COLS) or [R,N] (ROWS); out [M,R] or [R,M]."""
    X = _points(X, "X", spec.D)
    Z = _points(Z, "Z", spec.D, X.dtype)
    W = _hip.check_tensor(W, "W", dtype=X.dtype)
    if W.dim() != 2:
        raise ValueError("W must be 2-D")
    N, M = X.shape[0], Z.shape[0]
    R = W.shape[1] if w_layout == COLS else W.shape[0]
    if (W.shape[0] if w_layout == COLS else W.shape[1]) != N:
        raise ValueError(f"W shape {tuple(W.shape)} does not match N={N}")
    out_layout = w_layout if out_layout is None else out_layout
    out = torch.empty((M, R) if out_layout == COLS else (R, M), dtype=X.dtype, device=X.device)
    if M == 0 or R == 0:
        return out
    # N == 0 (a rank without rows) goes through the C-ABI too: libmgp writes the zeros
    hd = _hip.get_handle(X.device)
    k = spec.struct(_hip.dtype_code(X))
    hd.check(hd.lib.mgp_kmn_matvec(hd.h, ctypes.byref(k), _hip.ptr(X), N, _hip.ptr(Z), M, _hip.ptr(W), R,
                                   w_layout, _hip.ptr(out), out_layout))
    return out


def k_dense(spec, A, B, jitter=0.0, diag_add=None):
    """out [na, nb] = k(A, B) (+ jitter I) (+ diag(diag_add)); the diagonal terms need na == nb."""
    A = _points(A, "A", spec.D)
    B = _points(B, "B", spec.D, A.dtype)
    na, nb = A.shape[0], B.shape[0]
    if (jitter != 0.0 or diag_add is not None) and na != nb:
        raise ValueError("diagonal terms need a square block")
    if diag_add is not None:
        diag_add = _hip.check_tensor(diag_add, "diag_add", dtype=A.dtype).reshape(-1)
        if diag_add.shape[0] != na:
            raise ValueError("diag_add length mismatch")
    out = torch.empty((na, nb), dtype=A.dtype, device=A.device)
    if na == 0 or nb == 0:
        return out
    hd = _hip.get_handle(A.device)
    k = spec.struct(_hip.dtype_code(A))
    hd.check(hd.lib.mgp_k_dense(hd.h, ctypes.byref(k), _hip.ptr(A), na, _hip.ptr(B), nb, _hip.ptr(out), nb,
                                float(jitter), _hip.ptr(diag_add)))
    return out


def k_dense_vjp(spec, A, B, G):
    """(dL/dvariance: float, dL/dlengthscales: list[D]) for K = k(A,B) given G = dL/dK (row F2)."""
    A = _points(A, "A", spec.D)
    B = _points(B, "B", spec.D, A.dtype)
    G = _hip.check_tensor(G, "G", dtype=A.dtype, shape=(A.shape[0], B.shape[0]))
    dvar = ctypes.c_double(0.0)
    dls = (ctypes.c_double * _hip.MGP_MAX_D)()
    if A.shape[0] and B.shape[0]:
        hd = _hip.get_handle(A.device)
        k = spec.struct(_hip.dtype_code(A))
        hd.check(hd.lib.mgp_k_dense_vjp(hd.h, ctypes.byref(k), _hip.ptr(A), A.shape[0], _hip.ptr(B), B.shape[0],
                                        _hip.ptr(G), B.shape[0], ctypes.byref(dvar), dls))
    return dvar.value, [dls[d] for d in range(spec.D)]


def kmm_lambda_matvec(spec, Z, lam, V):
    """out [R,M] = V [R,M] @ (k(Z,Z) + diag(lam)), matrix-free (row M2)."""
    Z = _points(Z, "Z", spec.D)
    M = Z.shape[0]
    lam = _hip.check_tensor(lam, "lam", dtype=Z.dtype, shape=(M,))
    V = _hip.check_tensor(V, "V", dtype=Z.dtype)
    if V.dim() != 2 or V.shape[1] != M:
        raise ValueError(f"V must be [R, M={M}], got {tuple(V.shape)}")
    out = torch.empty_like(V)
    if V.shape[0] == 0 or M == 0:
        return out
    hd = _hip.get_handle(Z.device)
    k = spec.struct(_hip.dtype_code(Z))
    hd.check(hd.lib.mgp_kmm_lambda_matvec(hd.h, ctypes.byref(k), _hip.ptr(Z), M, _hip.ptr(lam), _hip.ptr(V),
                                          V.shape[0], _hip.ptr(out)))
    return out


def kmn_knm(spec, X, Z):
    """out [M,M] = k(Z,X) k(X,Z) on the matrix cores (row S1)."""
    X = _points(X, "X", spec.D)
    Z = _points(Z, "Z", spec.D, X.dtype)
    M = Z.shape[0]
    out = torch.empty((M, M), dtype=X.dtype, device=X.device)
    if M == 0:
        return out
    hd = _hip.get_handle(X.device)
    k = spec.struct(_hip.dtype_code(X))
    hd.check(hd.lib.mgp_kmn_knm(hd.h, ctypes.byref(k), _hip.ptr(X), X.shape[0], _hip.ptr(Z), M, _hip.ptr(out)))
    return out


def kmn_sq_colsum(spec, X, Z):
    """out [M] = sum_i k(x_i, z_m)^2 = diag(K_mn K_nm)."""
    X = _points(X, "X", spec.D)
    Z = _points(Z, "Z", spec.D, X.dtype)
    M = Z.shape[0]
    out = torch.empty((M,), dtype=X.dtype, device=X.device)
    if M == 0:
        return out
    hd = _hip.get_handle(X.device)
    k = spec.struct(_hip.dtype_code(X))
    hd.check(hd.lib.mgp_kmn_sq_colsum(hd.h, ctypes.byref(k), _hip.ptr(X), X.shape[0], _hip.ptr(Z), M, _hip.ptr(out)))
    return out


def symm_matmul(A, P):
    """out [Bt,n] = P [Bt,n] @ A [n,n] for symmetric A (row M2)."""
    A = _hip.check_tensor(A, "A")
    P = _hip.check_tensor(P, "P", dtype=A.dtype)
    if A.dim() != 2 or A.shape[0] != A.shape[1]:
        raise ValueError("A must be square")
    if P.dim() != 2 or P.shape[1] != A.shape[0]:
        raise ValueError(f"P shape {tuple(P.shape)} does not match n={A.shape[0]}")
    out = torch.empty_like(P)
    if P.numel() == 0:
        return out
    hd = _hip.get_handle(A.device)
    hd.check(hd.lib.mgp_symm_matmul(hd.h, _hip.dtype_code(A), _hip.ptr(A), A.shape[0], _hip.ptr(P), P.shape[0],
                                    _hip.ptr(out)))
    return out


def colwise_dot(A, B):
    """out [cols] = sum over rows of A*B (tf.reduce_sum(A * B, axis=0), models.py:343)."""
    A = _hip.check_tensor(A, "A")
    B = _hip.check_tensor(B, "B", dtype=A.dtype, shape=tuple(A.shape))
    if A.dim() != 2:
        raise ValueError("A must be 2-D")
    out = torch.empty((A.shape[1],), dtype=A.dtype, device=A.device)
    if A.shape[1] == 0:
        return out
    hd = _hip.get_handle(A.device)
    hd.check(hd.lib.mgp_colwise_dot(hd.h, _hip.dtype_code(A), _hip.ptr(A), _hip.ptr(B), A.shape[0], A.shape[1],
                                    _hip.ptr(out)))
    return out


def dot_all(A, B):
    """float(sum(A*B)) accumulated in fp64 (Hutchinson trace, models.py:313)."""
    A = _hip.check_tensor(A, "A")
    B = _hip.check_tensor(B, "B", dtype=A.dtype)
    if A.numel() != B.numel():
        raise ValueError("size mismatch")
    hd = _hip.get_handle(A.device)
    out = ctypes.c_double(0.0)
    hd.check(hd.lib.mgp_dot_all(hd.h, _hip.dtype_code(A), _hip.ptr(A), _hip.ptr(B), A.numel(), ctypes.byref(out)))
    return out.value


DIST_TYPES = {"sqeuclidean": 0, "euclidean": 1, "covariance": 2, "correlation": 3}


def nearest_center(spec, X, Z, distance_type="sqeuclidean", return_distance=True):
    """idx [N] int64 = argmin_m d(Z_m, X_i); optional best distance [N] (row F1)."""
    X = _points(X, "X", spec.D)
    Z = _points(Z, "Z", spec.D, X.dtype)
    if Z.shape[0] == 0:
        raise ValueError("need at least one centre")
    N = X.shape[0]
    idx = torch.empty((N,), dtype=torch.int64, device=X.device)
    best = torch.empty((N,), dtype=X.dtype, device=X.device) if return_distance else None
    if N > 0:
        hd = _hip.get_handle(X.device)
        k = spec.struct(_hip.dtype_code(X))
        hd.check(hd.lib.mgp_nearest_center(hd.h, ctypes.byref(k), DIST_TYPES[distance_type], _hip.ptr(X), N,
                                           _hip.ptr(Z), Z.shape[0], _hip.ptr(idx), _hip.ptr(best)))
    return (idx, best) if return_distance else idx


def cluster_stats(idx, y, M, method="auto"):
    """(sums, counts [M]) of y per cluster, deterministic order.  y [N] -> sums [M]; y [N,C] ->
    sums [M,C].

    "sweep": the fused N x M transpose sweep (`mgp_cluster_stats`, one column per launch).
    "sorted": group the rows by cluster with one stable device sort, then `mgp_segment_sums` --
    N C work instead of N M C.  "auto" takes the sort once N M exceeds 2^29 pairs (C3: 0.40 ms vs
    0.93 ms; C2, 2e8 pairs: 0.21 vs 0.09 ms) or several columns are asked for (C3, 8 columns: 0.55 ms
    vs 7.4 ms)."""
    idx = _hip.check_tensor(idx, "idx", dtype=torch.int64).reshape(-1)
    y = _hip.check_tensor(y, "y")
    multi = y.dim() == 2 and y.shape[1] > 1
    Y = y if multi else y.reshape(-1, 1)
    N, C = Y.shape
    if idx.shape[0] != N:
        raise ValueError("idx / y length mismatch")
    if method == "auto":
        method = "sorted" if (C > 1 or N * M > (1 << 29)) else "sweep"
    hd = _hip.get_handle(y.device)
    if method == "sweep":
        cols, counts = [], None
        for c in range(C):
            sums = torch.empty((M,), dtype=y.dtype, device=y.device)
            counts = torch.empty((M,), dtype=y.dtype, device=y.device)
            col = Y[:, c].contiguous()
            hd.check(hd.lib.mgp_cluster_stats(hd.h, _hip.dtype_code(y), _hip.ptr(idx), _hip.ptr(col), N, M,
                                              _hip.ptr(sums), _hip.ptr(counts)))
            cols.append(sums)
        sums = torch.stack(cols, dim=1)
    elif method == "sorted":
        if N and (int(idx.min()) < 0 or int(idx.max()) >= M):
            raise ValueError("cluster index out of range")
        order = torch.argsort(idx, stable=True)
        cnt = torch.bincount(idx, minlength=M)
        offsets = torch.zeros((M + 1,), dtype=torch.int64, device=idx.device)
        torch.cumsum(cnt, dim=0, out=offsets[1:])
        sums = torch.empty((M, C), dtype=y.dtype, device=y.device)
        Yc = Y.contiguous()
        hd.check(hd.lib.mgp_segment_sums(hd.h, _hip.dtype_code(y), _hip.ptr(order), _hip.ptr(offsets), _hip.ptr(Yc),
                                         N, C, M, _hip.ptr(sums)))
        counts = cnt.to(y.dtype)
    else:
        raise ValueError(f"unknown method {method!r}")
    return (sums if multi else sums[:, 0].contiguous()), counts
