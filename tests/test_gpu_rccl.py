"""The native multi-GPU boundary of include/mgp.h: RCCL communicator entry points, the per-step
all-reduce issued by libmgp itself, the agreement word that keeps every rank on the same iteration,
and the fixed workspace of mgp_create_ex.

One GPU box: RCCL refuses two ranks on one device, so the native path runs on a ONE-rank
communicator (bit-identical to the no-collective path) and the multi-rank logic of
the agreement word is driven through the callback hook with a scripted second rank.
"""

import ctypes
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd")


def dev():
    return torch.device("cuda:0")


def _problem(N=6000, D=3, M=64, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    Z = X[rng.choice(N, M, replace=False)]
    rhs = rng.standard_normal((2, M))
    return X, Z, rhs


def test_c_abi_communicator_single_rank():
    from cggp import _hip
    lib = _hip.load_library()
    comms = (ctypes.c_void_p * 1)()
    devs = (ctypes.c_int * 1)(0)
    rc = lib.mgp_comm_init_all(1, devs, comms)
    assert rc == 0, lib.mgp_comm_last_error()
    c = ctypes.c_void_p(comms[0])
    assert lib.mgp_comm_size(c) == 1 and lib.mgp_comm_rank(c) == 0
    for dt in (torch.float64, torch.float32):
        t = torch.arange(1, 4098, dtype=dt, device=dev())
        ref = t.clone()
        s = torch.cuda.current_stream().cuda_stream
        assert lib.mgp_comm_group_begin() == 0
        assert lib.mgp_allreduce_sum(ctypes.c_void_p(t.data_ptr()), t.numel(), _hip.dtype_code(t), c,
                                     ctypes.c_void_p(s)) == 0
        assert lib.mgp_comm_group_end() == 0
        torch.cuda.synchronize()
        assert torch.equal(t, ref)  # a sum over one rank
    assert lib.mgp_allreduce_sum(None, 4, _hip.F64, c, None) == -1 and b"buf" in lib.mgp_comm_last_error()
    assert lib.mgp_allreduce_sum(ctypes.c_void_p(1), 4, 7, c, None) == -3
    assert lib.mgp_allreduce_sum(ctypes.c_void_p(1), 4, _hip.F64, None, None) == -1
    assert lib.mgp_comm_destroy(c) == 0 and lib.mgp_comm_destroy(None) == 0
    assert lib.mgp_comm_size(None) == 0 and lib.mgp_comm_rank(None) == -1


_NATIVE_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path[:0] = [{root!r}, {pkg!r}]
    from cggp import kernels, ops, parallel
    from cggp.conjugate_gradient import ConjugateGradient, SgprNormalOperator, SubsampledNormalPreconditioner
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    rng = np.random.default_rng(0)
    N, D, M = 6000, 3, 64
    X = rng.standard_normal((N, D)); Z = X[rng.choice(N, M, replace=False)]; rhs = rng.standard_normal((M, 2))
    for dt in (torch.float64, torch.float32):
        Xt, Zt, bt = (torch.from_numpy(a).to(dev).to(dt) for a in (X, Z, rhs))
        kern = kernels.Matern32(1.2, [0.8, 1.0, 1.3])
        ar = parallel.make_allreduce(force=True)
        assert ar is not None and ar.comm is not None and ar.comm.world_size == 1  # native RCCL, no callback
        op_c = SgprNormalOperator(kern, Xt, Zt, 0.1, jitter=1e-6, allreduce=ar, kmm_rows=parallel.kmm_slab(M))
        st, keep = op_c._struct()
        assert st.comm and not st.allreduce and not st.partial_buf  # the library issues ncclAllReduce itself
        op_0 = SgprNormalOperator(kern, Xt, Zt, 0.1, jitter=1e-6)
        v = torch.from_numpy(rng.standard_normal((M, 3))).to(dev).to(dt)
        assert torch.equal(op_c.matmul(v), op_0.matmul(v))
        thr = 1e-12 if dt == torch.float64 else 1e-3
        cg = ConjugateGradient(thr, max_iterations=500, check_every=7)
        sc, (kc, ec) = cg.solve_with_stats(op_c, bt)
        s0, (k0, e0) = cg.solve_with_stats(op_0, bt)
        assert int(kc) == int(k0) and 1 < int(kc) < 500, (int(kc), int(k0))
        assert torch.equal(sc, s0) and torch.equal(ec, e0)  # SURVEY 4(iii): bit-identical to no collective
        # the other reductions of the path go through the same communicator
        pre_c = SubsampledNormalPreconditioner(op_c, rows_per_inducing=32, seed=3)
        pre_0 = SubsampledNormalPreconditioner(op_0, rows_per_inducing=32, seed=3)
        assert torch.equal(pre_c.inverse, pre_0.inverse)
        t = torch.arange(10, dtype=dt, device=dev); ar(t)
        assert torch.equal(t, torch.arange(10, dtype=dt, device=dev))
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("NATIVE_OK")
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_native_rccl_path_is_bit_identical_on_one_rank():
    """mgp_operator.comm: ncclAllReduce issued from inside mgp_pcg_solve on the solve's stream."""
    code = _NATIVE_WORKER.format(root=ROOT, pkg=PKG, port=_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "NATIVE_OK" in out.stdout, out.stderr[-3000:]


_FALLBACK_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path[:0] = [{root!r}, {pkg!r}]
    from cggp import kernels, parallel
    from cggp.conjugate_gradient import ConjugateGradient, SgprNormalOperator
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)

    class Broken:
        def __init__(self, group=None, device=None, timeout_s=None):
            raise RuntimeError("communicator bootstrap failed (scripted)")
    parallel.Communicator = Broken
    ar = parallel.make_allreduce(force=True)
    assert ar is not None and ar.comm is None and "scripted" in ar.native_error
    rng = np.random.default_rng(0)
    N, D, M = 6000, 3, 64
    X = rng.standard_normal((N, D)); Z = X[rng.choice(N, M, replace=False)]; rhs = rng.standard_normal((M, 2))
    Xt, Zt, bt = (torch.from_numpy(a).to(dev) for a in (X, Z, rhs))
    kern = kernels.SquaredExponential(1.2, [0.8, 1.0, 1.3])
    op_c = SgprNormalOperator(kern, Xt, Zt, 0.1, jitter=1e-6, allreduce=ar, kmm_rows=parallel.kmm_slab(M))
    st, keep = op_c._struct()
    assert not st.comm and st.allreduce  # the callback hook, reaching torch.distributed's RCCL on device tensors
    op_0 = SgprNormalOperator(kern, Xt, Zt, 0.1, jitter=1e-6)
    cg = ConjugateGradient(1e-12, max_iterations=500, check_every=7)
    sc, (kc, ec) = cg.solve_with_stats(op_c, bt)
    s0, (k0, e0) = cg.solve_with_stats(op_0, bt)
    assert int(kc) == int(k0) and torch.equal(sc, s0) and torch.equal(ec, e0)
    t = torch.arange(10, dtype=torch.float64, device=dev); ar(t)
    assert torch.equal(t, torch.arange(10, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("FALLBACK_OK")
""")


def test_nccl_group_falls_back_to_torch_all_reduce_when_the_native_communicator_fails():
    """parallel.AllReduce on backend "nccl": when libmgp's communicator cannot be created on some rank every rank
    agrees (one all-reduce) to use torch.distributed's all_reduce on the device tensor through the callback hook."""
    code = _FALLBACK_WORKER.format(root=ROOT, pkg=PKG, port=_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "FALLBACK_OK" in out.stdout, out.stderr[-3000:]


class _ScriptedSecondRank:
    """Stands in for rank 1 of a 2-rank job through the callback hook: it holds no rows (zero partial)
    and reports `active` for the first `agree` operator applications only."""

    comm = None
    world_size = 2

    def __init__(self, agree):
        self.agree, self.calls = agree, 0

    def __call__(self, t):
        if t.numel() == 1 or t.dtype not in (torch.float64, torch.float32):
            return
        t[-1] += 1.0 if self.calls < self.agree else 0.0
        self.calls += 1


def test_agreement_word_stops_every_rank_on_the_same_iteration():
    """If one rank's stopping test says "converged" an iteration earlier than another's (a one-ulp
    disagreement on 0.5||r||^2 > thr), nobody may run ahead: the reduced agreement word closes the
    local gate, the update of that iteration is skipped, and the step count and the solution are those
    of the iteration every rank completed."""
    from cggp import kernels
    from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient
    X, Z, rhs = _problem()
    Xt, Zt, bt = (torch.from_numpy(a).to(dev()) for a in (X, Z, rhs))
    kern = kernels.SquaredExponential(1.0, [1.0, 1.0, 1.0])
    M = Z.shape[0]
    for agree in (1, 5, 13):
        fake = _ScriptedSecondRank(agree)
        op = SgprNormalOperator(kern, Xt, Zt, 0.1, jitter=1e-6, allreduce=fake, kmm_rows=(0, M))
        sol, (k, err) = conjugate_gradient(op, bt, None, 1e-30, max_iterations=40, max_steps_cycle=10 ** 6,
                                           check_every=8)
        assert int(k) == agree, (int(k), agree)
        assert fake.calls >= agree + 1  # the collectives of the batch were all issued (no rank waits alone)
        ref_op = SgprNormalOperator(kern, Xt, Zt, 0.1, jitter=1e-6)
        ref, (kr, err_r) = conjugate_gradient(ref_op, bt, None, 1e-30, max_iterations=agree, max_steps_cycle=10 ** 6,
                                              check_every=8)
        assert int(kr) == agree and torch.equal(sol, ref) and torch.equal(err, err_r)


_WORKSPACE_WORKER = textwrap.dedent("""
    import sys
    import numpy as np, torch
    sys.path[:0] = [{root!r}, {pkg!r}]
    from cggp import _hip, kernels, ops
    from cggp.conjugate_gradient import ConjugateGradient, SgprNormalOperator
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    N, D, M = 20000, 3, 128
    X = torch.from_numpy(rng.standard_normal((N, D))).to(dev); Z = X[:M].clone()
    rhs = torch.from_numpy(rng.standard_normal((M, 2))).to(dev)
    op = SgprNormalOperator(kernels.SquaredExponential(1.0, [1.0] * D), X, Z, 0.1, jitter=1e-6)
    try:
        sol = ConjugateGradient(1e-10, max_iterations=300)(op, rhs)
        hd = _hip.get_handle(dev)
        print("RESULT ok", float(sol.double().abs().sum()), int(hd.lib.mgp_workspace_bytes(hd.h)))
    except _hip.MgpError as e:
        print("RESULT error", str(e))
""")


def test_fixed_workspace_never_allocates_and_fails_loudly_when_too_small():
    """mgp_create_ex(workspace_bytes): size it from a growing handle's mgp_workspace_bytes, get the same
    answer with no allocation after create; a workspace that is too small is an MGP_E_NOMEM with the
    shortfall named, not a hidden hipMalloc."""
    code = _WORKSPACE_WORKER.format(root=ROOT, pkg=PKG)

    def run(ws):
        env = dict(os.environ, MGP_WORKSPACE_BYTES=str(ws))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        return [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][0].split(" ", 2)

    tag, val, used = run(0)[1], *run(0)[2].split(" ")
    assert tag == "ok"
    need = int(used)
    assert need > 0
    fixed = run(2 * need)  # head-room for regions abandoned when an arena grows in the fixed pool
    assert fixed[1] == "ok" and fixed[2].split(" ")[0] == val  # same bits in the result
    small = run(4096)
    assert small[1] == "error" and "fixed workspace exhausted" in small[2]
