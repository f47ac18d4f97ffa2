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
    names = re.findall(r"^\s*(?:int|int64_t|size_t|double|void|const char\*)\s+(mgp_\w+)\s*\(", text, flags=re.M)
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
    assert lib.mgp_version() == _hip.MGP_VERSION == 210
    assert lib.mgp_build_arch() == b"gfx950"
    # struct sizes the C side compiles to (LP64): keeps the ctypes mirror honest
    assert ctypes.sizeof(_hip.MgpKernel) == 4 * 4 + 8 + 8 * _hip.MGP_MAX_D
    assert ctypes.sizeof(_hip.MgpOperator) == 8 + 8 + 8 * 14 + 8 + 8
    assert ctypes.sizeof(_hip.MgpPrecond) == 8 + 8 + 8 * 4 + 8 * 4
    assert ctypes.sizeof(_hip.MgpCgStats) == 16


_ABI_PROBE = r"""
#include <stddef.h>
#include <stdio.h>
#include "mgp.h"
#define F(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main(void) {
  printf("version %d\n", mgp_version());           /* resolved from libmgp.so at link time */
  printf("macro_version %d\n", MGP_VERSION);
  printf("max_d %d\n", MGP_MAX_D);
  printf("comm_id_bytes %d\n", MGP_COMM_ID_BYTES);
  printf("sizeof.mgp_kernel %zu\n", sizeof(mgp_kernel));
  printf("sizeof.mgp_operator %zu\n", sizeof(mgp_operator));
  printf("sizeof.mgp_precond %zu\n", sizeof(mgp_precond));
  printf("sizeof.mgp_cg_stats %zu\n", sizeof(mgp_cg_stats));
  F(mgp_kernel, kind); F(mgp_kernel, dtype); F(mgp_kernel, D); F(mgp_kernel, reserved); F(mgp_kernel, variance);
  F(mgp_kernel, lengthscales);
  F(mgp_operator, kind); F(mgp_operator, dtype); F(mgp_operator, n); F(mgp_operator, A); F(mgp_operator, kernel);
  F(mgp_operator, X); F(mgp_operator, N); F(mgp_operator, Z); F(mgp_operator, M); F(mgp_operator, Kmm);
  F(mgp_operator, s2); F(mgp_operator, lambda); F(mgp_operator, allreduce); F(mgp_operator, allreduce_ctx);
  F(mgp_operator, partial_buf); F(mgp_operator, kmm_row_begin); F(mgp_operator, kmm_row_end); F(mgp_operator, comm);
  F(mgp_operator, world_size); F(mgp_operator, reserved);
  F(mgp_precond, kind); F(mgp_precond, block_size); F(mgp_precond, num_blocks); F(mgp_precond, diag_inv);
  F(mgp_precond, block_index); F(mgp_precond, block_inv); F(mgp_precond, dense_inv); F(mgp_precond, apply);
  F(mgp_precond, apply_ctx); F(mgp_precond, cb_r); F(mgp_precond, cb_z);
  F(mgp_cg_stats, iterations); F(mgp_cg_stats, converged); F(mgp_cg_stats, seconds);
  return 0;
}
"""


def test_header_compiles_as_c_and_matches_the_ctypes_mirror(lib, tmp_path):
    """A C compiler owns the ABI: include/mgp.h is compiled as plain C (gcc, -Wall -Werror -pedantic),
    linked against libmgp.so, and every struct size and field offset it reports is compared with the
    ctypes structures of cggp/_hip.py (the binding the tests and the product go through)."""
    import shutil
    import subprocess
    from cggp import _hip
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi_probe.c"
    src.write_text(_ABI_PROBE)
    exe = tmp_path / "abi_probe"
    libdir = os.path.dirname(_hip.lib_path())
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o",
           str(exe), "-L", libdir, "-l:libmgp.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = dict(line.rsplit(" ", 1) for line in out.stdout.strip().splitlines())
    got = {k: int(v) for k, v in got.items()}
    assert got["version"] == got["macro_version"] == _hip.MGP_VERSION
    assert got["max_d"] == _hip.MGP_MAX_D and got["comm_id_bytes"] == _hip.MGP_COMM_ID_BYTES
    mirror = {"mgp_kernel": _hip.MgpKernel, "mgp_operator": _hip.MgpOperator, "mgp_precond": _hip.MgpPrecond,
              "mgp_cg_stats": _hip.MgpCgStats}
    rename = {"lambda": "lam"}  # `lambda` is a Python keyword
    checked = 0
    for key, val in got.items():
        if key.startswith("sizeof."):
            assert ctypes.sizeof(mirror[key[7:]]) == val, key
            checked += 1
        elif "." in key:
            st, field = key.split(".")
            assert getattr(mirror[st], rename.get(field, field)).offset == val, key
            checked += 1
    assert checked == 4 + 6 + 20 + 11 + 3
    # and the other direction: the mirror declares no field the header does not have
    for st, cls in mirror.items():
        for fname, _ in cls._fields_:
            back = {v: k for k, v in rename.items()}.get(fname, fname)
            assert f"{st}.{back}" in got, (st, fname)


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


def test_integration_doc_carries_the_generated_ctypes_mirror():
    """INTEGRATION.md's struct stubs are generated from cggp/_hip.py (tools/gen_integration_stubs.py); the
    document may not drift from them (round 1's stale `lengthscales * 32` / missing fields)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_stubs", os.path.join(ROOT, "tools", "gen_integration_stubs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a = text.index("<!-- BEGIN GENERATED: tools/gen_integration_stubs.py -->")
    b = text.index("<!-- END GENERATED -->")
    block = text[a:b].split("```python\n", 1)[1].rsplit("```", 1)[0]
    assert block == mod.generate()
    ns = {}
    exec(block, ns)  # and it is valid Python that reproduces the layout
    from cggp import _hip
    for name in ("MgpKernel", "MgpOperator", "MgpPrecond", "MgpCgStats"):
        assert ctypes.sizeof(ns[name]) == ctypes.sizeof(getattr(_hip, name))
