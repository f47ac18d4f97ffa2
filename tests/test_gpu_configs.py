"""BASELINE.json's configs at their stated sizes, every leg the config names:

* C4 (N=1e7, D=2, M=8192, fp32; an 8-GPU config) with ALL rows on one GPU: K_nm.v, K_mn.w against
  the fp64 oracle on spot rows/columns, adjointness, the 8-shard sum, fixed CG steps of the SGPR
  operator against `oracle/cg.py`, and the refreshed preconditioned solve down to its fp32 floor.
* C3's 64 Hutchinson probes at M=4096: `prior_kl(probes=)` and `logdet_gradient(probes=)` against
  `oracle/models.py` with the same injected probes (reference `cggp/models.py:37-44,308-314`).
* C5's explicit K_mn K_nm at N=2^20, M=4096, D=32 (Matern-3/2): entries against the oracle, symmetry.

What the oracle cannot do in seconds is said where it happens: a dense fp64 operator application on
1.25e6 x 8192 pairs takes ~100 s on 16 host cores, so the step-for-step CG comparison runs on a
65536-row slice of the shard and the full shard is held to the fp64 HIP solve (itself pinned to
the oracle in tests/test_gpu_parity.py) and to the true residual.
"""

import numpy as np
import pytest
import torch

from oracle import cg as ocg
from oracle import kernels as ok
from oracle import models as om

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


# --------------------------------------------------------------------------------------- C4
@pytest.fixture(scope="module")
def c4():
    from cggp import kernels, synthetic
    N, D, M, dt, kname = synthetic.CONFIGS["C4"]
    syn = synthetic.make_inputs(N, D, M, dt, need_y=True)
    X, Z, y = (torch.from_numpy(a).to(dev()) for a in (syn.X, syn.Z, syn.y))
    kern = kernels.SquaredExponential(1.0, [1.0] * D)
    ko = ok.Kernel(kname, 1.0, np.ones(D))
    yield syn, X, Z, y, kern, ko
    del X, Z, y
    torch.cuda.empty_cache()


def test_c4_full_n_kernel_products_fp32(c4):
    from cggp import ops, synthetic
    syn, X, Z, y, kern, ko = c4
    N, D = syn.X.shape
    M = syn.Z.shape[0]
    assert (N, D, M) == (10_000_000, 2, 8192) and X.dtype == torch.float32
    spec = kern.spec(D)
    v = torch.from_numpy(synthetic.make_vectors(M, 1, "float32")).to(dev())
    w = torch.from_numpy(np.random.default_rng(5).standard_normal((N, 1)).astype(np.float32)).to(dev())
    u = ops.knm_matvec(spec, X, Z, v)   # K_nm v   [N,1]
    t = ops.kmn_matvec(spec, X, Z, w)   # K_mn w   [M,1]
    assert torch.isfinite(u).all() and torch.isfinite(t).all()
    X64, Z64 = syn.X.astype(np.float64), syn.Z.astype(np.float64)
    # spot rows of K_nm v (first, last, ragged tail of the last workgroup, random) vs the fp64 oracle
    rows = np.r_[0, 1, N - 1, N - 2, np.random.default_rng(6).integers(0, N, 60)]
    ref = ko.K(X64[rows], Z64) @ v.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(u.cpu().numpy()[rows] - ref)) / np.max(np.abs(ref)) < 2e-4
    # spot columns of K_mn w: 1e7-term fp32 sums, ordered two-stage reduction
    cols = np.r_[0, M - 1, np.random.default_rng(7).integers(0, M, 4)]
    wn = w.cpu().numpy().astype(np.float64)
    ref_t = np.zeros((len(cols), 1))
    for s in range(0, N, 1 << 20):
        ref_t += ko.K(Z64[cols], X64[s:s + (1 << 20)]) @ wn[s:s + (1 << 20)]
    scale = np.sqrt(N)  # size of a 1e7-term sum of O(1) terms with random signs
    assert np.max(np.abs(t.cpu().numpy()[cols] - ref_t)) / scale < 2e-4
    # adjointness <K v, w> = <v, K^T w>, accumulated in fp64 from the fp32 results
    lhs = float((u.double() * w.double()).sum())
    rhs = float((v.double() * t.double()).sum())
    norm = float(u.double().norm() * w.double().norm())
    assert abs(lhs - rhs) / norm < 1e-5
    # the multi-GPU decomposition of SURVEY 8e: 8 contiguous row shards, partials summed
    per = -(-N // 8)
    acc = torch.zeros(M, 1, dtype=torch.float64, device=dev())
    for g in range(8):
        acc += ops.kmn_matvec(spec, X[g * per:(g + 1) * per], Z, w[g * per:(g + 1) * per]).double()
    assert float((acc - t.double()).abs().max()) / float(t.double().abs().max()) < 2e-5
    # run-to-run determinism of the ordered reduction
    assert torch.equal(t, ops.kmn_matvec(spec, X, Z, w))


class _TorchCpuSgprOperator:
    """Dense fp64 S = s2 (Kmm + jI) + K_mn K_nm of the oracle (oracle/models.py:SgprNormalOperator),
    with the K_nm chunks evaluated by oracle/cpu_baseline.py on every host core."""

    def __init__(self, X, Z, ko, s2, jitter):
        from oracle import cpu_baseline
        self._apply = cpu_baseline.sgpr_operator_apply
        self.X, self.Z = torch.from_numpy(X), torch.from_numpy(Z)
        self.Kmm = torch.from_numpy(ok.Kuu(Z, ko, jitter=jitter))
        self.s2 = s2
        self.shape = (Z.shape[0], Z.shape[0])
        self.ls = torch.ones(Z.shape[1], dtype=torch.float64)

    def rmatmul(self, P):
        out = self._apply(self.X, self.Z, torch.from_numpy(np.ascontiguousarray(P.T)), self.Kmm, self.s2, 1.0,
                          self.ls, "se", chunk=8192)
        return out.numpy().T


def test_c4_cg_steps_against_oracle(c4):
    """3 CG steps of S alpha = K_mn y in fp32 on a slice of one rank's shard vs oracle/cg.py in fp64."""
    from cggp import ops
    from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient
    syn, X, Z, y, kern, ko = c4
    ns = 65536  # 5e8 pairs per dense fp64 operator application on the host: ~5 s each, four of them
    Xs, ys = X[:ns].contiguous(), y[:ns].contiguous()
    op = SgprNormalOperator(kern, Xs, Z, 0.1, jitter=1e-6)
    rhs = ops.kmn_matvec(kern.spec(2), Xs, Z, ys).t().contiguous()
    X64, Z64, y64 = syn.X[:ns].astype(np.float64), syn.Z.astype(np.float64), syn.y[:ns].astype(np.float64)
    oop = _TorchCpuSgprOperator(X64, Z64, ko, 0.1, 1e-6)
    rhs_o = np.zeros((1, Z64.shape[0]))
    for s in range(0, ns, 16384):
        rhs_o += (ko.K(Z64, X64[s:s + 16384]) @ y64[s:s + 16384]).T
    assert np.max(np.abs(rhs.cpu().numpy() - rhs_o)) / np.max(np.abs(rhs_o)) < 1e-5
    for steps in (3,):
        sol, (k, err) = conjugate_gradient(op, rhs, None, 0.0, max_iterations=steps, max_steps_cycle=10 ** 6,
                                           check_every=steps)
        sol_o, (k_o, err_o) = ocg.conjugate_gradient(oop, rhs_o, np.zeros_like(rhs_o), 0.0, max_iterations=steps,
                                                     max_steps_cycle=10 ** 6)
        assert int(k) == steps == int(k_o)
        d = np.max(np.abs(sol.cpu().numpy() - sol_o)) / np.max(np.abs(sol_o))
        assert d < 2e-3, (steps, d)  # fp32 recurrence against fp64: cond(S) amplifies 6e-8 per step
        assert abs(float(err) - float(err_o[0, 0])) / float(err_o[0, 0]) < 2e-2


def test_c4_shard_cg_fp32_against_fp64_hip_and_true_residual(c4):
    """The same three steps on the whole 1.25e6-row shard of one rank: fp32 against the fp64 HIP
    solve (the oracle needs ~100 s per operator application at this size), and the recurrence
    residual against the recomputed one."""
    from cggp import kernels, ops
    from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient
    syn, X, Z, y, kern, ko = c4
    ns = 1_250_000
    Xs, ys = X[:ns].contiguous(), y[:ns].contiguous()
    op = SgprNormalOperator(kern, Xs, Z, 0.1, jitter=1e-6)
    rhs = ops.kmn_matvec(kern.spec(2), Xs, Z, ys).t().contiguous()
    op64 = SgprNormalOperator(kern, Xs.double(), Z.double(), 0.1, jitter=1e-6)
    rhs64 = ops.kmn_matvec(kern.spec(2), Xs.double(), Z.double(), ys.double()).t().contiguous()
    assert float((rhs.double() - rhs64).abs().max()) / float(rhs64.abs().max()) < 1e-5
    kw = dict(max_iterations=3, max_steps_cycle=10 ** 6, check_every=3)
    sol, (k, err) = conjugate_gradient(op, rhs, None, 0.0, **kw)
    sol64, (k64, err64) = conjugate_gradient(op64, rhs64, None, 0.0, **kw)
    assert int(k) == int(k64) == 3
    assert float((sol.double() - sol64).abs().max()) / float(sol64.abs().max()) < 2e-3
    r = rhs64 - op64.rmatmul(sol.double())
    true_half = 0.5 * float((r * r).sum())
    assert abs(true_half - float(err)) / true_half < 5e-2
    assert true_half < 0.5 * float((rhs64 * rhs64).sum())


def test_c4_refreshed_pcg_reaches_fp32_floor(c4):
    """fp32 PCG with the subsampled normal-equation preconditioner: with the reference's residual
    refresh (`max_steps_cycle`, conjugate_gradient.py:71-84) every 4 steps the true residual reaches
    the fp32 floor of the system and stays there; without it the recurrence drifts."""
    from cggp import ops
    from cggp.conjugate_gradient import SgprNormalOperator, SubsampledNormalPreconditioner, conjugate_gradient
    syn, X, Z, y, kern, ko = c4
    ns = 1_250_000
    Xs, ys = X[:ns].contiguous(), y[:ns].contiguous()
    op = SgprNormalOperator(kern, Xs, Z, 0.1, jitter=1e-6)
    rhs = ops.kmn_matvec(kern.spec(2), Xs, Z, ys).t().contiguous()
    b2 = float((rhs.double() ** 2).sum())
    pre = SubsampledNormalPreconditioner(op, rows_per_inducing=16)

    def rel(cycle, cap):
        sol, (k, _) = conjugate_gradient(op, rhs, None, 1e-6, pre, max_iterations=cap, max_steps_cycle=cycle,
                                         check_every=16)
        assert int(k) == cap  # the absolute 1e-6 rule is far below the fp32 floor: runs to the cap
        r = rhs.double() - op.rmatmul(sol).double()
        return np.sqrt(float((r * r).sum()) / b2)

    refreshed = rel(4, 64)
    drifting = rel(10 ** 6, 256)
    assert refreshed < 5e-5, refreshed
    assert drifting > 10 * refreshed, (drifting, refreshed)


# --------------------------------------------------------------------------------------- C3 probes
@pytest.fixture(scope="module")
def c3_cdgp():
    from cggp import kernels, synthetic
    from cggp.conjugate_gradient import ConjugateGradient
    from cggp.models import CGGP
    from cggp.optimize import assign_inducing_parameters, oips_update_inducing_parameters
    N, D, M, dt, kname = synthetic.CONFIGS["C3"]
    syn = synthetic.make_inputs(N, D, M, dt)
    X, y, Z = (torch.from_numpy(a).to(dev()) for a in (syn.X, syn.y, syn.Z))
    kern = kernels.SquaredExponential(1.0, [1.0] * D)
    m = CGGP(kern, 0.1, Z, ConjugateGradient(1e-6), num_probes=64, num_data=N)
    assign_inducing_parameters(m, *oips_update_inducing_parameters(m, (X, y), Z))
    probes = synthetic.make_probes(M, 64)
    ko = ok.Kernel(kname, 1.0, np.ones(D))
    u, counts = m.pseudo_u.cpu().numpy(), m.cluster_counts.cpu().numpy()
    assert counts.sum() == N and counts.min() >= 1
    Kmm = ok.Kuu(syn.Z, ko, jitter=0.0)
    KL = om.add_diagonal(Kmm, (0.1 / counts)[:, 0])
    yield syn, m, probes, ko, u, counts, Kmm, KL
    del X, y, Z
    torch.cuda.empty_cache()


def test_c3_prior_kl_64_probes_m4096(c3_cdgp):
    """`CGGP.prior_kl` with P=64 injected Rademacher probes at M=4096 (`cggp/models.py:293-322`)."""
    from cggp.conjugate_gradient import ConjugateGradient
    syn, m, probes, ko, u, counts, Kmm, KL = c3_cdgp
    M = Kmm.shape[0]
    pt = torch.from_numpy(probes).to(dev())
    # (i) the closed form the estimator is defined by: direct solves instead of CG
    a = np.linalg.solve(KL, u)
    S = np.linalg.solve(KL, probes)
    trace = np.sum(S * (Kmm @ probes)) / 64.0
    exact = 0.5 * (np.sum((Kmm @ a) * a) - trace + 0.0 - np.sum(np.log(0.1 / counts)))
    m.conjugate_gradient = ConjugateGradient(1e-13, max_iterations=4 * M)
    kl_tight = m.prior_kl(probes=pt)
    assert abs(kl_tight - exact) / abs(exact) < 1e-9, (kl_tight, exact)
    # (ii) the reference algorithm at its own threshold (cli_utils.py:439: 1e-6), both through CG
    m.conjugate_gradient = ConjugateGradient(1e-6)
    kl = m.prior_kl(probes=pt)
    ref = om.CGGP(ko, 0.1, syn.Z, ocg.ConjugateGradient(1e-6), num_probes=64, pseudo_u=u, cluster_counts=counts,
                  num_data=syn.X.shape[0])
    kl_o = ref.prior_kl(probes=probes)
    assert abs(kl - kl_o) / abs(kl_o) < 1e-6, (kl, kl_o, exact)
    assert abs(kl - exact) / abs(exact) < 1e-5


def test_c3_logdet_gradient_64_probes_m4096(c3_cdgp):
    """`eval_logdet` backward with probes, (1/P) CG(K, Zp) (df Zp)^T (`cggp/models.py:37-44`)."""
    from cggp.conjugate_gradient import ConjugateGradient
    syn, m, probes, ko, u, counts, Kmm, KL = c3_cdgp
    pt = torch.from_numpy(probes).to(dev())
    exact = np.linalg.solve(KL, probes) @ probes.T / 64.0
    m.conjugate_gradient = ConjugateGradient(1e-13, max_iterations=4 * Kmm.shape[0])
    g_tight = m.logdet_gradient(probes=pt).cpu().numpy()
    assert np.linalg.norm(g_tight - exact) / np.linalg.norm(exact) < 1e-8  # measured 2.8e-9 at thr 1e-13
    m.conjugate_gradient = ConjugateGradient(1e-6)
    g = m.logdet_gradient(probes=pt).cpu().numpy()
    g_o = om.eval_logdet_grad(KL, ocg.ConjugateGradient(1e-6), 1.0, 64, probes)
    # both stop at 0.5||r||^2 <= 1e-6 per column; what that leaves in K^-1 Zp is bounded by
    # ||K^-1|| sqrt(2e-6) per column, on trajectories that differ in rounding
    assert np.linalg.norm(g - g_o) / np.linalg.norm(g_o) < 1e-4
    assert np.linalg.norm(g - exact) / np.linalg.norm(exact) < 1e-4


# --------------------------------------------------------------------------------------- C5 contraction
def test_c5_kmn_knm_full_size():
    """K_mn K_nm [M,M] on the matrix cores at N=2^20, M=4096, D=32, Matern-3/2 (GPflow SGPR's A A^T)."""
    from cggp import kernels, ops, synthetic
    N, D, M, dt, kname = synthetic.CONFIGS["C5"]
    syn = synthetic.make_inputs(N, D, M, dt, need_y=False)
    X, Z = torch.from_numpy(syn.X).to(dev()), torch.from_numpy(syn.Z).to(dev())
    kern = kernels.Matern32(1.0, [1.0] * D)
    KK = ops.kmn_knm(kern.spec(D), X, Z)
    assert KK.shape == (M, M) and torch.isfinite(KK).all()
    assert torch.equal(KK, KK.t())  # mirrored from the upper triangle: exactly symmetric
    ko = ok.Kernel(kname, 1.0, np.ones(D))
    idx = np.r_[0, 1, 127, 128, 2047, M - 1, np.random.default_rng(11).integers(0, M, 6)]
    Kz = np.zeros((len(idx), len(idx)))
    for s in range(0, N, 1 << 17):
        Kc = ko.K(syn.Z[idx], syn.X[s:s + (1 << 17)])
        Kz += Kc @ Kc.T
    got = KK.cpu().numpy()[np.ix_(idx, idx)]
    assert np.max(np.abs(got - Kz) / np.abs(Kz)) < 1e-10
    # the diagonal is the k^2 column sum of the fused sweep (mgp_kmn_sq_colsum): two code paths, one number
    d = ops.kmn_sq_colsum(kern.spec(D), X, Z).reshape(-1)
    assert float((KK.diagonal() - d).abs().max() / d.abs().max()) < 1e-11
