"""The C-ABI library loads and exports every symbol include/mgp.h declares (no GPU needed);
the product path refuses to run without a GPU instead of falling back to anything."""

import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from cggp import _hip
    if not os.path.exists(_hip.lib_path()):
        import __graft_entry__ as g
        g.build()
    return _hip.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mgp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:int|int64_t|double|void|const char\*)\s+(mgp_\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_symbols_exported_and_typed(lib):
    from cggp import _hip
    names = declared_symbols()
    assert len(names) >= 18 and "mgp_pcg_solve" in names and "mgp_knm_matvec" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mgp.h but not exported by libmgp.so"
        assert n in _hip.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_hip.SIGNATURES) == names  # the binding declares nothing the header does not


def test_version_arch_and_struct_layout(lib):
    from cggp import _hip
    assert lib.mgp_version() == 100
    assert lib.mgp_build_arch() == b"gfx950"
    # struct sizes the C side compiles to (LP64): keeps the ctypes mirror honest
    assert ctypes.sizeof(_hip.MgpKernel) == 4 * 4 + 8 + 8 * _hip.MGP_MAX_D
    assert ctypes.sizeof(_hip.MgpOperator) == 8 + 8 + 8 * 14
    assert ctypes.sizeof(_hip.MgpPrecond) == 8 + 8 + 8 * 4
    assert ctypes.sizeof(_hip.MgpCgStats) == 16


def test_null_handle_is_an_error_not_a_crash(lib):
    k = __import__("cggp._hip", fromlist=["x"]).make_kernel_struct("se", 1, 2, 1.0, [1.0, 1.0])
    assert lib.mgp_knm_matvec(None, ctypes.byref(k), None, 0, None, 0, None, 0, 0, None, 0) == -1
    assert lib.mgp_set_stream(None, None) == -1
    assert lib.mgp_last_error(None) == b"invalid handle"


def test_host_entry_points_validate_arguments(lib):
    """The cover-tree entry points are host code: bad arguments come back as codes, never a crash."""
    tree = ctypes.c_void_p()
    x = (ctypes.c_double * 6)(0, 0, 1, 1, 2, 2)
    assert lib.mgp_covertree_build(None, 3, 2, 0.0, 2, 1, 1, ctypes.byref(tree)) == -2 and not tree.value
    assert b"N > 0" in lib.mgp_host_last_error()
    assert lib.mgp_covertree_build(x, 3, 2, 0.0, 2, 1, 1, None) == -1
    assert lib.mgp_covertree_build(x, 3, 2, 0.0, 0, 1, 1, ctypes.byref(tree)) == -1  # zero levels
    assert lib.mgp_covertree_build(x, 3, 2, 100.0, 1, 1, 1, ctypes.byref(tree)) == -1  # resolution > data radius
    assert lib.mgp_covertree_build(x, 3, 2, 0.0, 2, 1, 1, ctypes.byref(tree)) == 0 and tree.value
    assert lib.mgp_covertree_num_levels(tree) == 2
    assert lib.mgp_covertree_level_size(tree, 5) == -1 and lib.mgp_covertree_level_radius(tree, -1) == -1.0
    assert lib.mgp_covertree_level_nodes(tree, 7, None, None, None) == -1
    assert lib.mgp_covertree_level_rows(tree, 1, None, None) == -1
    n = lib.mgp_covertree_level_size(tree, 1)
    off = (ctypes.c_int64 * (n + 1))()
    assert lib.mgp_covertree_level_rows(tree, 1, off, None) == 0 and off[n] == 3  # offsets only
    lib.mgp_covertree_destroy(tree)
    lib.mgp_covertree_destroy(None)
    assert lib.mgp_covertree_num_levels(None) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_product_path_fails_loudly_without_gpu(lib):
    from cggp import kernels, ops
    from cggp.conjugate_gradient import ConjugateGradient
    k = kernels.SquaredExponential(1.0, [1.0, 1.0])
    X = torch.zeros((4, 2), dtype=torch.float64)
    with pytest.raises(RuntimeError, match="GPU|CPU fallback"):
        ops.knm_matvec(k.spec(2), X, X, torch.zeros((4, 1), dtype=torch.float64))
    with pytest.raises(RuntimeError):
        ConjugateGradient(1e-6)(torch.eye(4, dtype=torch.float64), torch.ones((4, 1), dtype=torch.float64))
    with pytest.raises(RuntimeError):
        k.K(X)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd", "cggp")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"
