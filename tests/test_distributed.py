"""The N > 1 path: row sharding + one all-reduce of the [Bt, M] partial per operator application.

CPU (gloo, world sizes 2 and 3): the sharding helpers and the collective glue, with the oracle computing
each rank's partial -- rows split unevenly, and an inducing set (M = 40) that 3 does not divide, so the
row slabs of the replicated s2*Kmm.p term are ragged too.  GPU (gloo rehearsal on one card, 2 and 3
ranks): the real libmgp CG loop with the all-reduce callback crossing the C ABI, against the single-rank
solve and the oracle.
"""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _problem(N=1003, D=3, M=24):
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N, D))
    Z = X[rng.choice(N, M, replace=False)]
    y = np.sin(X).sum(1, keepdims=True)
    V = rng.standard_normal((M, 2))
    return X, Z, y, V


def test_shard_bounds_cover_and_ragged():
    sys.path.insert(0, PKG)
    from cggp.parallel import shard_bounds
    for N in (0, 1, 7, 8, 9, 1000, 1048576, 1000000):
        for G in (1, 2, 3, 4, 8):
            b = [shard_bounds(N, G, r) for r in range(G)]
            assert b[0][0] == 0 and b[-1][1] == N
            assert all(b[i][1] == b[i + 1][0] for i in range(G - 1))
            assert all(lo <= hi for lo, hi in b)
            per = -(-N // G) if N else 0
            assert all(hi - lo <= per for lo, hi in b)


def _cpu_worker(rank, world, port, out):
    _init(rank, world, port)
    from cggp import parallel
    from oracle import kernels as ok, models as om
    M = 40  # not a multiple of 3: ragged Kmm row slabs at world size 3 (14 + 14 + 12)
    X, Z, y, V = _problem(M=M)
    kern = ok.Kernel("matern32", 1.2, [0.8, 1.0, 1.3])
    lo, hi = parallel.shard_bounds(X.shape[0], world, rank)
    Xl = parallel.shard_rows(torch.from_numpy(X)).numpy()
    assert Xl.shape[0] == hi - lo
    allreduce = parallel.make_allreduce()
    assert allreduce is not None and allreduce.describe()[0] == world
    # what libmgp's SGPR operator hands the collective (csrc/cg.hip, apply_operator): the local partial
    # K_mn_g (K_n_g m V), PLUS this rank's row slab of the replicated s2*Kmm.V term, PLUS the agreement word --
    # then ONE all-reduce of the [M,R] + 1 buffer completes S.V on every rank
    Knm = kern.K(Xl, Z)
    part = Knm.T @ (Knm @ V)
    rb, re = parallel.kmm_slab(M)
    part[rb:re] += 0.1 * (ok.Kuu(Z, kern)[rb:re] @ V)
    buf = torch.from_numpy(np.concatenate([part.ravel(), [1.0]]))
    allreduce(buf)
    full = om.SgprNormalOperator(X, Z, kern, 0.1).matmul(V)
    got = buf.numpy()[:-1].reshape(M, -1)
    err = np.max(np.abs(got - full)) / np.max(np.abs(full))
    # every rank holds the same bits after the exchange (the replicated recurrences depend on it)
    same = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(same, buf)
    identical = all(torch.equal(same[0], t) for t in same)
    if rank == 0:
        out.put((err, float(buf[-1]), (rb, re), identical))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partial_sum_gloo_cpu(world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    err, word, slab0, identical = q.get()
    assert err < 1e-12  # fp64 re-association only
    assert word == float(world)  # the agreement word: a sum of `world` ones, exact
    assert slab0 == (0, -(-40 // world)) and identical


def test_kmm_slabs_tile_any_m():
    sys.path.insert(0, PKG)
    from cggp.parallel import kmm_slab
    for M in (1, 5, 40, 4096, 8192, 4001):
        for G in (1, 2, 3, 4, 5, 6, 7, 8):
            b = [kmm_slab(M, G, r) for r in range(G)]
            assert b[0][0] == 0 and b[-1][1] == M and all(b[i][1] == b[i + 1][0] for i in range(G - 1))
            # an EMPTY slab is never (0, 0): libmgp reads that pair as "unset = every row" (include/mgp.h)
            assert all(not (lo == 0 and hi == 0) for lo, hi in b)


def test_single_rank_has_no_collective():
    sys.path.insert(0, PKG)
    from cggp import parallel
    assert parallel.make_allreduce() is None  # world size 1: identical to the no-collective path


def _uneven_cuts(world):
    """Row cuts of the 2000-row SGPR problem: rank 0 holds 1300 rows, the last rank of a 3-rank job only 10."""
    return {2: [0, 1300, 2000], 3: [0, 1300, 1990, 2000]}[world]


def _gpu_worker(rank, world, port, out):
    _init(rank, world, port)
    from cggp import kernels, parallel
    from cggp.conjugate_gradient import ConjugateGradient, SgprNormalOperator
    dev = torch.device("cuda:0")
    X, Z, y, V = _problem(N=5001, D=3, M=40)
    kern = kernels.Matern32(1.2, [0.8, 1.0, 1.3])
    Xl = parallel.shard_rows(torch.from_numpy(X)).to(dev)
    Zt = torch.from_numpy(Z).to(dev)
    # rank 0 shards the replicated s2*Kmm.p term too; a second operator leaves it on rank 0
    op = SgprNormalOperator(kern, Xl, Zt, 0.1, jitter=1e-6, allreduce=parallel.make_allreduce(),
                            kmm_rows=parallel.kmm_slab(40))
    op_r0 = SgprNormalOperator(kern, Xl, Zt, 0.1, jitter=1e-6, allreduce=parallel.make_allreduce())
    v1 = torch.from_numpy(V[:, :1].copy()).to(dev)
    same = float((op.matmul(v1) - op_r0.matmul(v1)).abs().max()) / float(op.matmul(v1).abs().max())
    assert same < 1e-13, same
    Sv = op.matmul(torch.from_numpy(V).to(dev))
    rhs = torch.from_numpy(np.random.default_rng(1).standard_normal((40, 3))).to(dev)
    sol, (steps, err) = ConjugateGradient(1e-12, max_iterations=3000).solve_with_stats(op, rhs)
    # sharded build of the subsampled preconditioner (per-rank sample, all-reduced Gram) + PCG
    from cggp.conjugate_gradient import SubsampledNormalPreconditioner
    pre = SubsampledNormalPreconditioner(op, rows_per_inducing=32, seed=3)
    psol, (psteps, _) = ConjugateGradient(1e-12, preconditioner=pre, max_iterations=3000).solve_with_stats(op, rhs)
    # the SGPR model on an UNEVEN split that straddles the "auto" preconditioner rule (32 rows per inducing
    # point): 2000 rows in all (>= 32 * 40 = 1280), 1300 on rank 0 and 700 on rank 1.  N of the bound and the
    # preconditioner decision must come from the global row count on every rank, or the ranks take different
    # branches around collectives
    from cggp.models import SGPR
    Xu, Zu, yu, _ = _problem(N=2000, D=3, M=40)
    cuts = _uneven_cuts(world)
    cut = (cuts[rank], cuts[rank + 1])
    Xr, yr = (torch.from_numpy(a[cut[0]:cut[1]].copy()).to(dev) for a in (Xu, yu))
    m = SGPR((Xr, yr), kern, torch.from_numpy(Zu).to(dev), 0.1, ConjugateGradient(1e-12, max_iterations=3000),
             jitter=1e-6, allreduce=parallel.make_allreduce())
    assert m.num_data == 2000
    uses_pre = m.solver().preconditioner.__class__.__name__
    elbo = m.elbo()
    mu, var = m.predict_f(torch.from_numpy(Xu[:50].copy()).to(dev))
    torch.cuda.synchronize()
    if rank == 0:
        out.put((Sv.cpu().numpy(), sol.cpu().numpy(), int(steps), psol.cpu().numpy(), int(psteps),
                 pre.sample_rows, uses_pre, elbo, mu.cpu().numpy(), var.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sgpr_cg_on_one_gpu(world):
    """2 ranks, and 3 -- a world size that divides neither N = 5001 nor M = 40 (slabs 14 + 14 + 12)."""
    from oracle import cg as ocg, kernels as ok, models as om
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    Sv, sol, steps, psol, psteps, sample_rows, uses_pre, elbo, mu, var = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    X, Z, y, V = _problem(N=5001, D=3, M=40)
    oop = om.SgprNormalOperator(X, Z, ok.Kernel("matern32", 1.2, [0.8, 1.0, 1.3]), 0.1, jitter=1e-6)
    ref = oop.matmul(V)
    assert np.max(np.abs(Sv - ref)) / np.max(np.abs(ref)) < 1e-11
    rhs = np.random.default_rng(1).standard_normal((40, 3))
    o_sol, (o_steps, _) = ocg.ConjugateGradient(1e-12, max_iterations=3000).solve_with_stats(oop, rhs)
    assert steps < 3000 and abs(steps - o_steps) <= 6
    exact = np.linalg.solve(oop.dense(), rhs)
    scale = np.max(np.abs(exact))
    assert np.max(np.abs(sol - exact)) / scale < 1e-6 and np.max(np.abs(o_sol - exact)) / scale < 1e-6
    assert sample_rows == world * (-(-32 * 40 // world)) and psteps < steps
    assert np.max(np.abs(psol - exact)) / scale < 1e-6
    # uneven shards: same decision on both ranks, bound and predictions of the whole data set
    assert uses_pre == "SubsampledNormalPreconditioner"
    Xu, Zu, yu, _ = _problem(N=2000, D=3, M=40)
    ref = om.SGPR((Xu, yu), ok.Kernel("matern32", 1.2, [0.8, 1.0, 1.3]), Zu, 0.1, jitter=1e-6)
    assert abs(elbo - ref.elbo()) / abs(ref.elbo()) < 1e-9
    rmu, rvar = ref.predict_f(Xu[:50])
    assert np.max(np.abs(mu - rmu)) / np.max(np.abs(rmu)) < 1e-6
    assert np.max(np.abs(var - rvar)) / np.max(np.abs(rvar)) < 1e-4


# ---------------------------------------------------------------- bounded waits of the N > 1 launch
def test_bounded_call_returns_raises_and_times_out():
    import time
    sys.path.insert(0, PKG)
    from cggp import _hip, parallel
    assert parallel.bounded_call(lambda: 7, 5.0, "quick") == 7
    with pytest.raises(ZeroDivisionError):
        parallel.bounded_call(lambda: 1 // 0, 5.0, "raises")
    t0 = time.monotonic()
    with pytest.raises(_hip.MgpError, match="timed out"):
        parallel.bounded_call(lambda: time.sleep(30), 0.3, "a rank that never arrives")
    assert time.monotonic() - t0 < 5.0


def test_bench_watchdog_exits_124_with_the_stage_named():
    """bench.py's N > 1 launch never waits longer than its limits: a rank stuck in a stage leaves with status 124."""
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r); import bench; d = bench.Watchdog(3); "
            "d.arm('rendezvous', 0.5); time.sleep(60)" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 124, (out.returncode, out.stderr[-500:])
    assert "rank 3: timed out in stage 'rendezvous'" in out.stderr


@pytest.mark.gpu
def test_rccl_bootstrap_that_never_completes_is_bounded():
    """ncclCommInitRank blocks until every rank has joined: rank 0 of a 2-rank communicator whose peer never
    comes must give up after the limit (in a child process, which then leaves without waiting for the thread)."""
    import subprocess
    code = (
        "import sys, os, ctypes; sys.path[:0] = [%r, %r]\n"
        "import torch; from cggp import _hip, parallel\n"
        "torch.cuda.set_device(0); lib = _hip.load_library()\n"
        "buf = (ctypes.c_char * _hip.MGP_COMM_ID_BYTES)(); assert lib.mgp_comm_unique_id(buf) == 0\n"
        "c = ctypes.c_void_p()\n"
        "try:\n"
        "    parallel.bounded_call(lambda: lib.mgp_comm_init_rank(ctypes.byref(c), 0, 2, 0, buf), 5.0, 'bootstrap')\n"
        "except _hip.MgpError as e:\n"
        "    print(e, flush=True); os._exit(42)\n"
        "os._exit(0)\n" % (ROOT, PKG))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 42, (out.returncode, out.stdout[-500:], out.stderr[-1500:])
    assert "timed out" in out.stdout
