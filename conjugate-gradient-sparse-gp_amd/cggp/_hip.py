"""ctypes binding of libmgp.so (the C ABI of include/mgp.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
arithmetic step of the path runs in libmgp.so.  There is NO fallback: if the library is
missing, or a call is made without a GPU, this module raises.
"""

import ctypes
import os
import threading

import torch

MGP_VERSION = 210
MGP_MAX_D = 512
MGP_FUSED_MAX_D = 32
MGP_COMM_ID_BYTES = 128
F32, F64 = 0, 1
SE, MATERN12, MATERN32, MATERN52 = 0, 1, 2, 3
COLS, ROWS = 0, 1
PRE_EYE, PRE_JACOBI, PRE_BLOCK, PRE_DENSE, PRE_CALLBACK = 0, 1, 2, 3, 4
OP_DENSE, OP_SGPR, OP_KMM_LAMBDA = 0, 1, 2

KERNEL_KINDS = {"se": SE, "matern12": MATERN12, "matern32": MATERN32, "matern52": MATERN52}

_LIB_NAME = "libmgp.so"
_lib = None
_lib_lock = threading.Lock()


class MgpKernel(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("D", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("variance", ctypes.c_double),
        ("lengthscales", ctypes.c_double * MGP_MAX_D),
    ]


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                ctypes.c_int, ctypes.c_void_p)


class MgpOperator(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("n", ctypes.c_int64),
        ("A", ctypes.c_void_p),
        ("kernel", ctypes.POINTER(MgpKernel)),
        ("X", ctypes.c_void_p),
        ("N", ctypes.c_int64),
        ("Z", ctypes.c_void_p),
        ("M", ctypes.c_int64),
        ("Kmm", ctypes.c_void_p),
        ("s2", ctypes.c_double),
        ("lam", ctypes.c_void_p),
        ("allreduce", ALLREDUCE_FN),
        ("allreduce_ctx", ctypes.c_void_p),
        ("partial_buf", ctypes.c_void_p),
        ("kmm_row_begin", ctypes.c_int64),
        ("kmm_row_end", ctypes.c_int64),
        ("comm", ctypes.c_void_p),
        ("world_size", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


PRECOND_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                              ctypes.c_int64, ctypes.c_void_p)


class MgpPrecond(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("block_size", ctypes.c_int32),
        ("num_blocks", ctypes.c_int64),
        ("diag_inv", ctypes.c_void_p),
        ("block_index", ctypes.c_void_p),
        ("block_inv", ctypes.c_void_p),
        ("dense_inv", ctypes.c_void_p),
        ("apply", PRECOND_FN),
        ("apply_ctx", ctypes.c_void_p),
        ("cb_r", ctypes.c_void_p),
        ("cb_z", ctypes.c_void_p),
    ]


class MgpCgStats(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("converged", ctypes.c_int32),
                ("seconds", ctypes.c_double)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against mgp.h
_P, _I, _L, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
_KP = ctypes.POINTER(MgpKernel)
SIGNATURES = {
    "mgp_version": (_I, []),
    "mgp_create": (_I, [ctypes.POINTER(_P), _I]),
    "mgp_create_ex": (_I, [ctypes.POINTER(_P), _I, ctypes.c_size_t]),
    "mgp_workspace_bytes": (ctypes.c_size_t, [_P]),
    "mgp_destroy": (_I, [_P]),
    "mgp_set_stream": (_I, [_P, _P]),
    "mgp_last_error": (ctypes.c_char_p, [_P]),
    "mgp_build_arch": (ctypes.c_char_p, []),
    "mgp_knm_matvec": (_I, [_P, _KP, _P, _L, _P, _L, _P, ctypes.c_int32, _I, _P, _I]),
    "mgp_kmn_matvec": (_I, [_P, _KP, _P, _L, _P, _L, _P, ctypes.c_int32, _I, _P, _I]),
    "mgp_k_dense": (_I, [_P, _KP, _P, _L, _P, _L, _P, _L, _D, _P]),
    "mgp_kmn_knm": (_I, [_P, _KP, _P, _L, _P, _L, _P]),
    "mgp_kmn_sq_colsum": (_I, [_P, _KP, _P, _L, _P, _L, _P]),
    "mgp_symm_matmul": (_I, [_P, _I, _P, _L, _P, _L, _P]),
    "mgp_pcg_solve": (_I, [_P, ctypes.POINTER(MgpOperator), ctypes.POINTER(MgpPrecond), _P, _P, _L, _D,
                           _L, _L, _D, ctypes.c_int32, _P, _P, ctypes.POINTER(MgpCgStats)]),
    "mgp_operator_apply": (_I, [_P, ctypes.POINTER(MgpOperator), _P, _L, _P]),
    "mgp_colwise_dot": (_I, [_P, _I, _P, _P, _L, _L, _P]),
    "mgp_dot_all": (_I, [_P, _I, _P, _P, _L, ctypes.POINTER(_D)]),
    "mgp_nearest_center": (_I, [_P, _KP, _I, _P, _L, _P, _L, _P, _P]),
    "mgp_cluster_stats": (_I, [_P, _I, _P, _P, _L, _L, _P, _P]),
    "mgp_k_dense_vjp": (_I, [_P, _KP, _P, _L, _P, _L, _P, _L, ctypes.POINTER(_D), ctypes.POINTER(_D)]),
    "mgp_segment_sums": (_I, [_P, _I, _P, _P, _P, _L, _L, _L, _P]),
    "mgp_kmm_lambda_matvec": (_I, [_P, _KP, _P, _L, _P, _P, _L, _P]),
    "mgp_profile_enable": (_I, [_P, _I]),
    "mgp_profile_read": (_I, [_P, ctypes.POINTER(_L), ctypes.POINTER(_D)]),
    "mgp_profile_read_each": (_I, [_P, ctypes.POINTER(_D), _L, ctypes.POINTER(_L)]),
    "mgp_profile_read_clocks": (_I, [_P, ctypes.POINTER(_D), _L, ctypes.POINTER(_L)]),
    # collectives: RCCL communicator ranks
    "mgp_comm_unique_id": (_I, [_P]),
    "mgp_comm_init_rank": (_I, [ctypes.POINTER(_P), _I, _I, _I, _P]),
    "mgp_comm_init_all": (_I, [_I, ctypes.POINTER(_I), ctypes.POINTER(_P)]),
    "mgp_comm_destroy": (_I, [_P]),
    "mgp_comm_size": (_I, [_P]),
    "mgp_comm_rank": (_I, [_P]),
    "mgp_comm_group_begin": (_I, []),
    "mgp_comm_group_end": (_I, []),
    "mgp_allreduce_sum": (_I, [_P, ctypes.c_size_t, _I, _P, _P]),
    "mgp_comm_last_error": (ctypes.c_char_p, []),
    # host-only entry points (cover tree, row F3)
    "mgp_host_last_error": (ctypes.c_char_p, []),
    "mgp_covertree_build": (_I, [_P, _L, _I, _D, _I, _I, _I, ctypes.POINTER(_P)]),
    "mgp_covertree_build_device": (_I, [_P, _P, _P, _L, _I, _D, _I, _I, _I, ctypes.POINTER(_P)]),
    "mgp_covertree_destroy": (None, [_P]),
    "mgp_covertree_num_levels": (_I, [_P]),
    "mgp_covertree_level_size": (_L, [_P, _I]),
    "mgp_covertree_level_radius": (_D, [_P, _I]),
    "mgp_covertree_level_nodes": (_I, [_P, _I, _P, _P, _P]),
    "mgp_covertree_level_rows": (_I, [_P, _I, _P, _P]),
}


def lib_path():
    # MGP_LIBRARY lets A/B measurements load an experimental build; the default is the in-tree one
    return os.environ.get("MGP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load_library():
    """dlopen libmgp.so and type its entry points.  Raises if it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C conjugate-gradient-sparse-gp_amd/csrc`).  There is no CPU fallback.")
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


class MgpError(RuntimeError):
    pass


class Handle:
    """One libmgp handle per (device); the stream is refreshed from torch on every call."""

    def __init__(self, device_index):
        if not torch.cuda.is_available():
            raise RuntimeError("libmgp needs an MI355X: torch.cuda.is_available() is False "
                               "(there is no CPU fallback for the hot path)")
        self.lib = load_library()
        self.device_index = device_index
        h = ctypes.c_void_p()
        ws = int(os.environ.get("MGP_WORKSPACE_BYTES", "0") or 0)  # > 0: fixed workspace, no hipMalloc after create
        rc = self.lib.mgp_create_ex(ctypes.byref(h), device_index, ws)
        if rc != 0:
            raise MgpError(f"mgp_create_ex(device={device_index}, workspace_bytes={ws}) failed with {rc}")
        self.h = h

    def check(self, rc):
        if rc != 0:
            msg = self.lib.mgp_last_error(self.h)
            raise MgpError(f"libmgp error {rc}: {msg.decode() if msg else '?'}")

    def sync_stream(self):
        s = torch.cuda.current_stream(self.device_index).cuda_stream
        self.lib.mgp_set_stream(self.h, ctypes.c_void_p(s))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.mgp_destroy(self.h)
                self.h = None
        except Exception:
            pass


_handles = {}


def get_handle(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    hd = _handles.get(idx)
    if hd is None:
        hd = Handle(idx)
        _handles[idx] = hd
    hd.sync_stream()
    return hd


def dtype_code(t):
    if t.dtype == torch.float64:
        return F64
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"libmgp supports float32/float64 tensors, got {t.dtype}")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def check_tensor(t, name, dtype=None, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the hot path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} has dtype {t.dtype}, expected {dtype}")
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {shape}")
    return t.contiguous()


def make_kernel_struct(kind, dtype_c, D, variance, lengthscales):
    k = MgpKernel()
    k.kind = KERNEL_KINDS[kind] if isinstance(kind, str) else int(kind)
    k.dtype = dtype_c
    k.D = int(D)
    k.variance = float(variance)
    ls = [float(x) for x in lengthscales]
    if len(ls) == 1:
        ls = ls * D
    if len(ls) != D:
        raise ValueError(f"lengthscales has {len(ls)} entries for D={D}")
    if D > MGP_MAX_D:
        raise ValueError(f"D={D} > {MGP_MAX_D} is not supported")
    for i, v in enumerate(ls):
        k.lengthscales[i] = v
    return k
