"""Model surface of the hot path -- host mirror of `cggp/models.py` (rows M1-M6, S1).

`CGGP` (the CDGP model: every (Kmm+Lambda)^-1 applied by CG), its Cholesky twin `ClusterGP`,
`eval_logdet`, and a CG form of SGPR.  Names, constructor keywords, method signatures and
returned shapes follow the reference (`cggp/models.py:21-48,176-354`, `cggp/cli_utils.py:439-446`);
kernel evaluation, the K_nm products, the dense `p @ A` and the CG loop run in libmgp.
"""

import math

import numpy as np
import torch

from . import ops
from .conjugate_gradient import ConjugateGradient, SgprNormalOperator, SubsampledNormalPreconditioner
from .kernels import InducingPoints, Kuf, Kuu, inducingpoint_wrapper  # noqa: F401 (InducingPoints re-exported)
from .likelihoods import Gaussian


def rademacher(shape, dtype, device, seed):
    """+-1 probes from a documented stream: numpy PCG64(seed), `2*integers(0,2)-1`.

    The reference draws them with `tfp.random.rademacher` (`cggp/models.py:39,310`), whose
    stream cannot be reproduced outside TFP, so parity tests inject `probes=`.
    """
    z = 2 * np.random.default_rng(seed).integers(0, 2, size=shape) - 1
    return torch.from_numpy(z.astype(np.float64)).to(device=device, dtype=dtype)


class _EvalLogdet(torch.autograd.Function):
    """`eval_logdet` custom gradient (`cggp/models.py:26-46`): forward 0.0, backward
    df * CG(K, I)^T (exact) or CG(K, Zp) (df Zp)^T / P (probes)."""

    @staticmethod
    def forward(ctx, matrix, cg, num_probes, probes):
        ctx.cg, ctx.num_probes, ctx.probes = cg, num_probes, probes
        ctx.save_for_backward(matrix)
        return torch.zeros((), dtype=matrix.dtype, device=matrix.device)  # :46

    @staticmethod
    def backward(ctx, df):
        (matrix,) = ctx.saved_tensors
        with torch.no_grad():
            grad = eval_logdet_grad(matrix, ctx.cg, df, ctx.num_probes, ctx.probes)
        return grad, None, None, None


def eval_logdet_grad(matrix, cg, df=1.0, num_probes=None, probes=None, seed=0):
    """The backward of `eval_logdet` as a plain function (`cggp/models.py:30-44`)."""
    n = matrix.shape[-1]
    if num_probes is None and probes is None:
        eye = torch.eye(n, dtype=matrix.dtype, device=matrix.device)  # :33
        inv = cg(matrix, eye)  # :34
        return df * inv.t()  # :35-36
    if probes is None:
        probes = rademacher((n, num_probes), matrix.dtype, matrix.device, seed)  # :38-39
    P = probes.shape[1]
    rv = df * probes  # :40
    lv = cg(matrix, probes)  # :41
    return (lv @ rv.t()) / P  # :42  ([n,P]x[P,n] rank-P library GEMM)


def eval_logdet(matrix, cg, num_probes=None, probes=None):
    """`cggp/models.py:21-48`: value is the constant 0.0, only the gradient carries log|K|."""
    return _EvalLogdet.apply(matrix, cg, num_probes, probes)


class LpSVGP:
    """`cggp/models.py:51-173` (Panos et al. SVGP with diagonal q-covariance): parameters `nu [M,1]`
    and `diag_variance [M,1]`; Cholesky solves on the [M,M] matrix libmgp builds.  The base of the
    reference's class tree; `ClusterGP` replaces `nu`/`diag_variance` by pseudo_u and sigma^2/counts."""

    def __init__(self, kernel, likelihood, inducing_variable, *, mean_function=None, num_latent_gps=1, nu=None,
                 diag_variance=None, num_data=None):
        assert num_latent_gps == 1, "One latent GP is allowed"  # :76
        self.kernel = kernel
        self.likelihood = likelihood if not isinstance(likelihood, (int, float)) else Gaussian(likelihood)
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)
        self.mean_function = mean_function
        self.num_data = num_data
        Z = self.inducing_variable.Z
        shape = (Z.shape[0], 1)
        self.nu = (torch.zeros(shape, dtype=Z.dtype, device=Z.device) if nu is None  # :93
                   else torch.as_tensor(nu, dtype=Z.dtype, device=Z.device).reshape(shape).clone())
        self.diag_variance = (torch.full(shape, 1e-4, dtype=Z.dtype, device=Z.device) if diag_variance is None  # :94
                              else torch.as_tensor(diag_variance, dtype=Z.dtype, device=Z.device).reshape(shape).clone())

    def _mean(self, Xnew):
        return 0.0 if self.mean_function is None else self.mean_function(Xnew)

    def _K(self):
        Kmm = Kuu(self.inducing_variable, self.kernel, jitter=0.0)  # :112 / :141
        return Kmm, Kuu(self.inducing_variable, self.kernel, jitter=0.0, diag_add=self.diag_variance[:, 0])

    def prior_kl(self):  # :107-120
        Kmm, K = self._K()
        nut = self.nu.t().contiguous()
        quad = ops.dot_all(ops.symm_matmul(Kmm, nut), nut)
        L = torch.linalg.cholesky(K)
        trace = torch.cholesky_solve(Kmm, L).diagonal().sum().item()
        logdet = (2.0 * torch.log(L.diagonal())).sum().item() - torch.log(self.diag_variance).sum().item()
        return 0.5 * (quad - trace + logdet)

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):  # :136-161
        assert not full_output_cov
        _, K = self._K()
        Kmn = Kuf(self.inducing_variable, self.kernel, Xnew)
        L = torch.linalg.cholesky(K)
        A = torch.linalg.solve_triangular(L, Kmn, upper=False)
        if not full_cov:
            fvar = (self.kernel.K_diag(Xnew) - ops.colwise_dot(A, A))[:, None]
        else:
            fvar = (self.kernel.K(Xnew) - A.t() @ A)[None, ...]
        fmu = ops.knm_matvec(self.kernel.spec(Xnew.shape[1]), Xnew, self.inducing_variable.Z, self.nu)  # :157
        return fmu + self._mean(Xnew), fvar

    def scale(self, batch_size, dtype=None):  # :163-169
        return 1.0 if self.num_data is None else float(self.num_data) / float(batch_size)

    def elbo(self, data):  # :125-134
        x, y = data
        kl = self.prior_kl()
        f_mean, f_var = self.predict_f(x)
        var_exp = self.likelihood.variational_expectations(x, f_mean, f_var, y)
        return var_exp.sum().item() * self.scale(x.shape[0]) - kl

    def maximum_log_likelihood_objective(self, data):  # :122-123
        return self.elbo(data)

    def q_moments(self, full_cov=False):  # :171-173
        return self.predict_f(self.inducing_variable.Z, full_cov=full_cov)


class ClusterGP:
    """Cluster-data GP with Cholesky solves (`cggp/models.py:176-276`); parameter container of
    row M6.  The Cholesky factorisation itself is a plain library call (torch.linalg) on the
    [M,M] matrix libmgp builds; it is the twin the CG model is checked against."""

    def __init__(self, kernel, likelihood, inducing_variable, *, mean_function=None, num_latent_gps=1,
                 cluster_counts=None, num_data=None, pseudo_u=None):
        assert num_latent_gps == 1, "One latent GP is allowed"  # :189
        self.kernel = kernel
        self.likelihood = likelihood if not isinstance(likelihood, (int, float)) else Gaussian(likelihood)
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)
        self.mean_function = mean_function
        self.num_latent_gps = num_latent_gps
        self.num_data = num_data
        Z = self.inducing_variable.Z
        M = Z.shape[0]
        shape = (M, num_latent_gps)
        if pseudo_u is not None:
            pseudo_u = torch.as_tensor(pseudo_u, dtype=Z.dtype, device=Z.device)
            if tuple(pseudo_u.shape) != shape:  # :204-205
                raise ValueError("Pseudo-u argument shape must match actual pseudo-u shape.")
            self.pseudo_u = pseudo_u.clone()
        else:
            self.pseudo_u = torch.zeros(shape, dtype=Z.dtype, device=Z.device)  # nu = 0, :93
        if cluster_counts is not None:
            cluster_counts = torch.as_tensor(cluster_counts, dtype=Z.dtype, device=Z.device)
            if tuple(cluster_counts.shape) != shape:  # :209-210
                raise ValueError("Cluster counts argument shape must match pseudo-u shape.")
            self.cluster_counts = cluster_counts.clone()
        else:
            self.cluster_counts = torch.ones(shape, dtype=Z.dtype, device=Z.device)  # :213

    # ---- parameters
    @property
    def nu(self):  # :222-224
        raise NotImplementedError(f"This property is not supported in {self.__class__}")

    @property
    def diag_variance(self):  # :226-228
        return self.likelihood.variance / self.cluster_counts

    def _mean(self, Xnew):
        if self.mean_function is None:
            return 0.0
        return self.mean_function(Xnew)

    def _Kmm_and_KmmLambda(self):
        iv, kernel = self.inducing_variable, self.kernel
        Kmm = Kuu(iv, kernel, jitter=0.0)  # :236 / :300
        KmmLambda = Kuu(iv, kernel, jitter=0.0, diag_add=self.diag_variance[:, 0])  # add_diagonal fused
        return Kmm, KmmLambda

    # ---- Cholesky versions
    def prior_kl(self):  # :230-248
        Kmm, K = self._Kmm_and_KmmLambda()
        L = torch.linalg.cholesky(K)
        a = torch.cholesky_solve(self.pseudo_u, L)
        quad = ops.dot_all(ops.symm_matmul(Kmm, a.t().contiguous()), a.t().contiguous())
        trace = torch.cholesky_solve(Kmm, L).diagonal().sum().item()
        logdet = (2.0 * torch.log(L.diagonal())).sum().item()
        const = torch.log(self.diag_variance).sum().item()
        return 0.5 * (quad - trace + logdet - const)

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):  # :250-276
        assert not full_output_cov
        iv, kernel = self.inducing_variable, self.kernel
        _, K = self._Kmm_and_KmmLambda()
        Kmn = Kuf(iv, kernel, Xnew)
        L = torch.linalg.cholesky(K)
        a = torch.cholesky_solve(self.pseudo_u, L)
        A = torch.linalg.solve_triangular(L, Kmn, upper=False)
        if not full_cov:
            fvar = (kernel.K_diag(Xnew) - ops.colwise_dot(A, A))[:, None]
        else:
            fvar = (kernel.K(Xnew) - A.t() @ A)[None, ...]
        fmu = ops.knm_matvec(kernel.spec(Xnew.shape[1]), Xnew, iv.Z, a)
        return fmu + self._mean(Xnew), fvar

    # ---- shared by both
    def scale(self, batch_size, dtype=None):  # :163-169
        if self.num_data is not None:
            return float(self.num_data) / float(batch_size)
        return 1.0

    def elbo(self, data):  # :125-134
        x, y = data
        kl = self.prior_kl()
        f_mean, f_var = self.predict_f(x, full_cov=False, full_output_cov=False)
        var_exp = self.likelihood.variational_expectations(x, f_mean, f_var, y)
        return var_exp.sum().item() * self.scale(x.shape[0]) - kl

    def maximum_log_likelihood_objective(self, data):  # :122-123
        return self.elbo(data)

    def training_loss(self, data):
        return -self.elbo(data)

    def q_moments(self, full_cov=False):  # :171-173
        return self.predict_f(self.inducing_variable.Z, full_cov=full_cov)

    def predict_f_batched(self, X, batch_size):
        """`batch_posterior_computation` (`cggp/cli_utils.py:426-436`): stream rows in batches."""
        means, variances = [], []
        for s in range(0, X.shape[0], batch_size):
            mu, var = self.predict_f(X[s:s + batch_size])
            means.append(mu)
            variances.append(var)
        return torch.cat(means, 0), torch.cat(variances, 0)


class CGGP(ClusterGP):
    """`cggp/models.py:279-354`."""

    def __init__(self, kernel, likelihood, inducing_variable, conjugate_gradient, num_probes=5, **kwargs):
        super().__init__(kernel, likelihood, inducing_variable, **kwargs)
        self.conjugate_gradient = conjugate_gradient
        self.num_probes = num_probes
        self.probe_seed = 0

    def prior_kl(self, probes=None):  # :293-322
        Kmm, KmmLambda = self._Kmm_and_KmmLambda()  # :300-301
        cg = self.conjugate_gradient
        a = cg(KmmLambda, self.pseudo_u)  # :303
        if self.num_probes is None and probes is None:
            trace = cg(KmmLambda, Kmm).diagonal().sum().item()  # :305-306
        else:
            if probes is None:
                probes = rademacher((Kmm.shape[0], self.num_probes), Kmm.dtype, Kmm.device, self.probe_seed)
                self.probe_seed += 1
            S = cg(KmmLambda, probes)  # :311
            Kp = ops.symm_matmul(Kmm, probes.t().contiguous())  # (Kmm Zp)^T, :312
            trace = ops.dot_all(S.t().contiguous(), Kp) / probes.shape[1]  # :313-314
        at = a.t().contiguous()
        quad = ops.dot_all(ops.symm_matmul(Kmm, at), at)  # :316-317
        logdet = 0.0  # eval_logdet forward value, :319 -> :46
        const = torch.log(self.diag_variance).sum().item()  # :321
        return 0.5 * (quad - trace + logdet - const)  # :322

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False, _shared=None):  # :324-354
        assert not full_output_cov
        iv, kernel = self.inducing_variable, self.kernel
        cg = self.conjugate_gradient
        if _shared is None:
            _, KmmLambda = self._Kmm_and_KmmLambda()  # :333,337
            a = cg(KmmLambda, self.pseudo_u)  # :339
        else:  # predict_f_batched: the N-free pieces are the same for every batch
            KmmLambda, a = _shared
        Kmn = Kuf(iv, kernel, Xnew)  # :334
        W = cg(KmmLambda, Kmn)  # :340
        if not full_cov:
            fvar = (kernel.K_diag(Xnew) - ops.colwise_dot(Kmn, W))[:, None]  # :343-345
        else:
            fvar = (kernel.K(Xnew) - Kmn.t() @ W)[None, ...]  # :347-349
        fmu = ops.knm_matvec(kernel.spec(Xnew.shape[1]), Xnew, iv.Z, a)  # Kmn^T a, :351 (row M1)
        return fmu + self._mean(Xnew), fvar

    def predict_f_batched(self, X, batch_size, shared_inverse=False, inverse_threshold=None):
        """`batch_posterior_computation` (`cggp/cli_utils.py:426-436`); (Kmm+Lambda) and its solve
        against pseudo_u do not depend on the batch and are formed once.

        `shared_inverse=True` (build-side option for predicting many rows): instead of one
        B-column CG per batch -- iterations x 2 B M^2 flops each, the dominant cost of the reference's
        prediction -- solve (Kmm+Lambda) Y = I ONCE with the same device CG
        (M right-hand sides) and apply Y to every batch as one GEMM.  For N rows that is
        `iterations x 2 M^3 + 2 N M^2` flops instead of `iterations x 2 N M^2`.  The columns of I are
        solved to `inverse_threshold` (default: the model's threshold squared, so that the product
        with a batch stays inside the per-batch solve's own tolerance)."""
        _, KmmLambda = self._Kmm_and_KmmLambda()
        cg = self.conjugate_gradient
        a = cg(KmmLambda, self.pseudo_u)
        means, variances = [], []
        if not shared_inverse:
            shared = (KmmLambda, a)
            for s in range(0, X.shape[0], batch_size):
                mu, var = self.predict_f(X[s:s + batch_size], _shared=shared)
                means.append(mu)
                variances.append(var)
            return torch.cat(means, 0), torch.cat(variances, 0)
        M = KmmLambda.shape[0]
        thr = inverse_threshold if inverse_threshold is not None else min(cg.error_threshold ** 2, 1e-12)
        tight = ConjugateGradient(thr, cg.preconditioner, cg.max_iterations, cg.max_steps_cycle,
                                  min_float=cg.min_float, check_every=cg.check_every)
        Y = tight(KmmLambda, torch.eye(M, dtype=KmmLambda.dtype, device=KmmLambda.device))
        self.inverse_stats = tight.last_stats
        Y = (0.5 * (Y + Y.t())).contiguous()
        kernel, Z = self.kernel, self.inducing_variable.Z
        spec = kernel.spec(Z.shape[1])
        for s in range(0, X.shape[0], batch_size):
            xb = X[s:s + batch_size]
            Knm = ops.k_dense(spec, xb, Z)  # [B, M] = Kmn^T
            Wt = ops.symm_matmul(Y, Knm)  # Kmn^T Y  (Y symmetric)
            variances.append((kernel.K_diag(xb) - (Knm * Wt).sum(dim=1))[:, None])
            means.append(ops.knm_matvec(spec, xb, Z, a) + self._mean(xb))
        return torch.cat(means, 0), torch.cat(variances, 0)

    def elbo(self, data, probes=None):
        x, y = data
        kl = self.prior_kl(probes=probes)
        f_mean, f_var = self.predict_f(x, full_cov=False, full_output_cov=False)
        var_exp = self.likelihood.variational_expectations(x, f_mean, f_var, y)
        return var_exp.sum().item() * self.scale(x.shape[0]) - kl

    def elbo_over_batches(self, data, batch_size, shared_inverse=True, probes=None):
        """sum_b elbo(batch_b) -- the quantity `make_metrics_callback` accumulates as "train/elbo"
        (`cggp/optimize.py:336-338`) -- without repeating the batch-independent work: the prior KL is
        evaluated once and counted once per batch, and the predictive moments of all rows come from
        `predict_f_batched` (with `shared_inverse`: one M-column CG for the whole data set instead of
        one B-column CG per batch).  With Hutchinson probes the reference draws fresh probes in every
        batch's KL; here the one evaluation stands for all of them."""
        x, y = data
        kl = self.prior_kl(probes=probes)
        mu, var = self.predict_f_batched(x, batch_size, shared_inverse=shared_inverse)
        ve = self.likelihood.variational_expectations(x, mu, var, y)
        total, nb = 0.0, 0
        for s in range(0, x.shape[0], batch_size):
            part = ve[s:s + batch_size]
            total += part.sum().item() * self.scale(part.shape[0])
            nb += 1
        return total - nb * kl

    def logdet_gradient(self, df=1.0, probes=None):
        """d/dK of the omitted log|Kmm+Lambda| term (`eval_logdet` backward, row M5)."""
        _, KmmLambda = self._Kmm_and_KmmLambda()
        return eval_logdet_grad(KmmLambda, self.conjugate_gradient, df, self.num_probes, probes,
                                seed=self.probe_seed)


class SGPR:
    """SGPR predictions and bound with the N-sized products done matrix-free (row S1).

    The reference reaches `gpflow.models.SGPR` through `sgpr_class` (`cggp/cli_utils.py:444-446`).
    Its two-Cholesky closed form is restated here in normal-equation form so that the only
    N-sized work is the fused sweeps:
        S = s2 (Kmm + jitter I) + K_mn K_nm,   alpha = S^-1 K_mn y            (CG, matrix-free S)
        mean* = K_*m alpha
        var*  = k_** - K_*m (Kmm+jI)^-1 K_m* + s2 K_*m S^-1 K_m*            (CG, two operators)
    `elbo()` needs log-determinants, so it forms K_mn K_nm explicitly on the matrix cores
    (`ops.kmn_knm`) and factorises the [M,M] result.  Rows of X may be this rank's shard: pass
    `allreduce` (parallel.make_allreduce) and every N-sized reduction is summed over ranks.
    """

    def __init__(self, data, kernel, inducing_variable, noise_variance, conjugate_gradient=None, *,
                 jitter=1e-6, allreduce=None, num_data=None, preconditioner="auto", explicit_rhs=8,
                 kmm_solver="cholesky"):
        """`preconditioner`: "auto" (the subsampled normal-equation preconditioner built from
        min(N, 32 M) rows of the data set -- all ranks together), None, or a
        `CGPreconditioner` for the [M,M] system.  `explicit_rhs`: solves on S with at least this
        many right-hand sides form S once on the matrix cores (`ops.kmn_knm`, 2NM^2 flops) and run
        the dense CG on it -- a matrix-free step costs two N x M sweeps per right-hand-side chunk,
        so beyond a handful of columns the explicit matrix is cheaper (0 disables).
        `kmm_solver`: how `K_*m (Kmm+jI)^-1 K_m*` of the predictive variance is formed --
        "cholesky" (GPflow's own `L = chol(Kmm + jitter I)`, one [M,M] factorisation, cached) or
        "cg" (the model's `conjugate_gradient`; Kmm + 1e-6 I is badly conditioned, so this takes
        hundreds of steps)."""
        self.X, self.Y = data
        self.kernel = kernel
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)
        self.likelihood = Gaussian(noise_variance)
        self.conjugate_gradient = conjugate_gradient or ConjugateGradient(1e-6)
        self.jitter = float(jitter)
        self.allreduce = allreduce
        if num_data is None:
            # rows of X may be this rank's shard: N of the bound and of the "auto" preconditioner rule
            # is the GLOBAL row count, agreed by one all-reduce (every rank must call the constructor)
            num_data = self.X.shape[0]
            if allreduce is not None:
                t = torch.tensor([float(num_data)], dtype=torch.float64, device=self.X.device)
                allreduce(t)
                num_data = int(round(t.item()))
        self.num_data = num_data
        self.preconditioner = preconditioner
        self.explicit_rhs = int(explicit_rhs)
        if kmm_solver not in ("cholesky", "cg"):
            raise ValueError(f"unknown kmm_solver {kmm_solver!r}")
        self.kmm_solver = kmm_solver
        self.invalidate()

    # ---- caches.  Everything below is a function of (X, Y, Z, kernel and likelihood parameters,
    # jitter); `update_fn` / `multiple_assign` / `assign_inducing_parameters` change those from
    # outside (cggp/cli_utils.py:394-411, paper_cli_uci.py:123-124), so every public method first
    # compares a fingerprint of them and drops the caches when it moved.
    def invalidate(self):
        self._Lmm = None
        self._alpha = None
        self._op = None
        self._cg_S = None
        self._S = None
        self._KK = None
        self._key = self._fingerprint()
        # keep the fingerprinted tensors alive: id() of a freed tensor can be handed to the next one
        self._key_refs = (self.inducing_variable.Z, self.X, self.Y)

    def _fingerprint(self):
        Z = self.inducing_variable.Z
        k = self.kernel
        return (id(Z), Z._version, tuple(Z.shape), id(self.X), self.X._version, id(self.Y), self.Y._version,
                type(k).__name__, float(k.variance), tuple(float(v) for v in k.lengthscales),
                float(self.likelihood.variance), float(self.jitter), id(self.conjugate_gradient),
                id(self.preconditioner) if not isinstance(self.preconditioner, str) else self.preconditioner)

    def _sync(self):
        if self._fingerprint() != self._key:
            self.invalidate()

    def operator(self):
        self._sync()
        if self._op is None:
            self._op = SgprNormalOperator(self.kernel, self.X, self.inducing_variable.Z,
                                          self.likelihood.variance, jitter=self.jitter,
                                          allreduce=self.allreduce)
        return self._op

    def solver(self):
        """The CG used on S: the model's `conjugate_gradient` settings plus the preconditioner."""
        self._sync()
        if self._cg_S is None:
            cg, pre = self.conjugate_gradient, self.preconditioner
            if isinstance(pre, str):
                if pre != "auto":
                    raise ValueError(f"unknown preconditioner {pre!r}")
                # Always built (its build contains collectives: no rank may decide otherwise from its
                # own shard).  The sample is min(N, 32 M) rows: a data set smaller than that gives
                # P = S itself, and CG then acts as iterative refinement of the factorised solve --
                # which is what brings the predictive variance to the 1e-6 of the closed form when
                # cond(S) ~ cond(Kmm)^2 leaves the un-preconditioned recurrence stuck at its guard floor
                pre = SubsampledNormalPreconditioner(self.operator())
            if pre is None:
                self._cg_S = cg
            else:
                cycle = cg.max_steps_cycle
                if cycle is None and self.X.dtype != torch.float64:
                    # below fp64 the preconditioned recurrence drifts from the true residual within
                    # ~16 steps; the reference's residual refresh (:71-84) every 4 keeps it honest
                    cycle = 4
                self._cg_S = ConjugateGradient(cg.error_threshold, pre, cg.max_iterations, cycle,
                                               min_float=cg.min_float, check_every=cg.check_every)
        return self._cg_S

    def dense_S(self):
        """S = s2 (Kmm + jitter I) + K_mn K_nm as an [M,M] matrix (formed once, cached)."""
        self._sync()
        if self._S is None:
            self._S = torch.add(self.kmn_knm(), self.operator().Kmm, alpha=self.likelihood.variance)
        return self._S

    def kmn_knm(self):
        """K_mn K_nm [M,M] on the matrix cores, summed over ranks (formed once, cached)."""
        self._sync()
        if self._KK is None:
            Z = self.inducing_variable.Z
            KK = ops.kmn_knm(self.kernel.spec(Z.shape[1]), self.X, Z)
            if self.allreduce is not None:
                self.allreduce(KK.view(-1))
            self._KK = KK
        return self._KK

    def solve_S(self, rhs):
        """S^-1 rhs for rhs [M, R]: matrix-free for a few columns, explicit S beyond `explicit_rhs`."""
        self._sync()
        if self._S is not None or (self.explicit_rhs > 0 and rhs.shape[1] >= self.explicit_rhs):
            return self.solver()(self.dense_S(), rhs)
        return self.solver()(self.operator(), rhs)

    def _Kmn_y(self):
        Z = self.inducing_variable.Z
        b = ops.kmn_matvec(self.kernel.spec(Z.shape[1]), self.X, Z, self.Y)  # K_mn y, [M,1]
        if self.allreduce is not None:
            self.allreduce(b.view(-1))
        return b

    def alpha(self):
        self._sync()
        if self._alpha is None:
            self._alpha = self.solve_S(self._Kmn_y())
        return self._alpha

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        assert not full_output_cov
        self._sync()
        iv, kernel = self.inducing_variable, self.kernel
        mean = ops.knm_matvec(kernel.spec(Xnew.shape[1]), Xnew, iv.Z, self.alpha())
        Kms = Kuf(iv, kernel, Xnew)
        if self.kmm_solver == "cholesky":
            if self._Lmm is None:
                self._Lmm = torch.linalg.cholesky(self.operator().Kmm)  # Kmm + jitter I
            W1 = torch.cholesky_solve(Kms, self._Lmm)
        else:
            W1 = self.conjugate_gradient(self.operator().Kmm, Kms)
        W2 = self.solve_S(Kms)
        if full_cov:  # GPflow SGPR.predict_f: [1, B, B]
            cov = kernel.K(Xnew) - Kms.t() @ W1 + self.likelihood.variance * (Kms.t() @ W2)
            return mean, cov[None, ...]
        var = kernel.K_diag(Xnew) - ops.colwise_dot(Kms, W1) + self.likelihood.variance * ops.colwise_dot(Kms, W2)
        return mean, var[:, None]

    def elbo(self):
        """Titsias' collapsed bound, GPflow `SGPR.elbo` (const + logdet + quad + trace)."""
        self._sync()
        iv, kernel = self.inducing_variable, self.kernel
        Z = iv.Z
        s2 = self.likelihood.variance
        N = self.num_data
        KK = self.kmn_knm()  # on the matrix cores; shared with the explicit-S solves
        yy = ops.dot_all(self.Y, self.Y)
        if self.allreduce is not None:
            t = torch.tensor([yy], dtype=torch.float64, device=Z.device)
            self.allreduce(t)
            yy = t.item()
        Kmn_y = self._Kmn_y()
        kuu = Kuu(iv, kernel, jitter=self.jitter)
        L = torch.linalg.cholesky(kuu)
        # A A^T = L^-1 (K_mn K_nm) L^-T / s2
        T1 = torch.linalg.solve_triangular(L, KK, upper=False)
        AAT = torch.linalg.solve_triangular(L, T1.t().contiguous(), upper=False) / s2
        B = AAT + torch.eye(Z.shape[0], dtype=Z.dtype, device=Z.device)
        LB = torch.linalg.cholesky(B)
        Aerr = torch.linalg.solve_triangular(L, Kmn_y, upper=False) / math.sqrt(s2)
        c = torch.linalg.solve_triangular(LB, Aerr, upper=False) / math.sqrt(s2)
        const = -0.5 * N * math.log(2.0 * math.pi)
        logdet = -torch.log(LB.diagonal()).sum().item() - 0.5 * N * math.log(s2)
        quad = -0.5 * yy / s2 + 0.5 * (c * c).sum().item()
        trace = -0.5 * N * kernel.variance / s2 + 0.5 * AAT.diagonal().sum().item()
        return const + logdet + quad + trace


def cdgp_class(kernel, likelihood, iv, error_threshold=1e-6, **kwargs):
    """`cggp/cli_utils.py:439-441`."""
    conjugate_gradient = ConjugateGradient(error_threshold)
    return CGGP(kernel, likelihood, iv, conjugate_gradient, **kwargs)


def sgpr_class(train_data, kernel, likelihood, iv, **kwargs):
    """`cggp/cli_utils.py:444-446` (GPflow SGPR there; the CG form here)."""
    return SGPR(train_data, kernel, iv, noise_variance=likelihood.variance, **kwargs)


def rmse_nlpd(model, test_data, batch_size=None):
    """Test RMSE / NLPD as `make_metrics_callback` computes them (`cggp/optimize.py:302-309,344-350`)."""
    x, y = test_data
    bs = batch_size or x.shape[0]
    sq, lpd, n = 0.0, 0.0, 0
    for s in range(0, x.shape[0], bs):
        xb, yb = x[s:s + bs], y[s:s + bs]
        mu, var = model.predict_f(xb)
        lpd += model.likelihood.predict_log_density(xb, mu, var, yb).sum().item()
        sq += ((yb - mu) ** 2).sum().item()
        n += xb.shape[0]
    return math.sqrt(sq / n), -lpd / n
