"""Oracle: model surface on the path.

Restates `cggp/utils.py:11-17`, `cggp/models.py:21-48,125-134,163-173,226-276,
293-354` and the GPflow SGPR / Gaussian-likelihood closed forms in numpy.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""

import numpy as np

from .kernels import Kuu, Kuf
from .cg import ConjugateGradient

LOG2PI = np.log(2.0 * np.pi)


def add_diagonal(matrix, diagonal):
    """`cggp/utils.py:11-17`: matrix + diag(diagonal)."""
    out = np.array(matrix, copy=True)
    idx = np.arange(out.shape[0])
    out[idx, idx] = out[idx, idx] + np.asarray(diagonal).reshape(-1)
    return out


# ---------------------------------------------------------------- likelihood
def gaussian_variational_expectations(f_mean, f_var, y, variance):
    """GPflow Gaussian.variational_expectations, summed over the last axis -> [B]."""
    ve = -0.5 * LOG2PI - 0.5 * np.log(variance) - 0.5 * (np.square(y - f_mean) + f_var) / variance
    return np.sum(ve, axis=-1)


def gaussian_predict_log_density(f_mean, f_var, y, variance):
    """GPflow Gaussian.predict_log_density: log N(y | mu, var + s2), summed over last axis."""
    v = f_var + variance
    ld = -0.5 * (LOG2PI + np.log(v) + np.square(y - f_mean) / v)
    return np.sum(ld, axis=-1)


def rmse_nlpd(mu, var, y, variance):
    """`cggp/optimize.py:302-309,344-350`: test RMSE and NLPD from (mu,var)."""
    err = y - mu
    lpd = np.sum(gaussian_predict_log_density(mu, var, y, variance))
    return float(np.sqrt(np.mean(err ** 2))), float(-lpd / y.shape[0])


# ---------------------------------------------------------------- eval_logdet
def eval_logdet_forward(matrix):
    """`cggp/models.py:46`: the forward value is the constant 0.0."""
    return np.asarray(matrix).dtype.type(0.0)


def eval_logdet_grad(matrix, cg, df=1.0, num_probes=None, probes=None):
    """Backward of `eval_logdet` (`cggp/models.py:30-44`).

    exact: df * CG(matrix, I)^T.  probes: (1/P) * CG(matrix, Zp) @ (df*Zp)^T,
    the reference's unsymmetrised rank-P estimate.  Probes are injected (TFP's
    Rademacher stream is not reproducible outside TFP).
    """
    n = matrix.shape[-1]
    dtype = matrix.dtype
    if num_probes is None and probes is None:
        eye = np.eye(n, dtype=dtype)  # :33
        inv = cg(matrix, eye)  # :34
        return dtype.type(df) * inv.T  # :35-36
    assert probes is not None, "inject the probe matrix"
    P = probes.shape[1]
    rv = dtype.type(df) * probes  # :40
    lv = cg(matrix, probes)  # :41
    return (lv @ rv.T) / dtype.type(P)  # :42


# ---------------------------------------------------------------- LpSVGP
class LpSVGP:
    """`cggp/models.py:51-173`: q-mean `Kmn^T nu`, diagonal `diag_variance`, Cholesky solves."""

    def __init__(self, kernel, noise_variance, Z, nu=None, diag_variance=None, num_data=None):
        self.kernel, dt = kernel, kernel.dtype
        self.noise_variance = dt.type(noise_variance)
        self.Z = np.asarray(Z, dtype=dt)
        M = self.Z.shape[0]
        self.nu = np.zeros((M, 1), dt) if nu is None else np.asarray(nu, dt).reshape(M, 1)  # :93
        self.diag_variance = (np.full((M, 1), 1e-4, dt) if diag_variance is None  # :94
                              else np.asarray(diag_variance, dt).reshape(M, 1))
        self.num_data = num_data

    def prior_kl(self):  # :107-120
        Kmm = Kuu(self.Z, self.kernel, jitter=0.0)
        quad = np.sum(self.nu * (Kmm @ self.nu))
        K = add_diagonal(Kmm, self.diag_variance[:, 0])
        L = np.linalg.cholesky(K)
        trace = np.trace(np.linalg.solve(L.T, np.linalg.solve(L, Kmm)))
        logdet = np.sum(2.0 * np.log(np.diag(L))) - np.sum(np.log(self.diag_variance))
        return 0.5 * (quad - trace + logdet)

    def predict_f(self, Xnew, full_cov=False):  # :136-161
        Kmm = Kuu(self.Z, self.kernel, jitter=0.0)
        Kmn = Kuf(self.Z, self.kernel, Xnew)
        Knn = self.kernel.K(Xnew) if full_cov else self.kernel.K_diag(Xnew)
        L = np.linalg.cholesky(add_diagonal(Kmm, self.diag_variance[:, 0]))
        A = np.linalg.solve(L, Kmn)
        fvar = (Knn - A.T @ A)[None, ...] if full_cov else (Knn - np.sum(np.square(A), axis=0))[:, None]
        return Kmn.T @ self.nu, fvar

    def elbo(self, data):  # :125-134
        x, y = data
        f_mean, f_var = self.predict_f(x)
        ve = gaussian_variational_expectations(f_mean, f_var, y, self.noise_variance)
        scale = 1.0 if self.num_data is None else self.num_data / x.shape[0]
        return np.sum(ve) * scale - self.prior_kl()


# ---------------------------------------------------------------- ClusterGP / CGGP
class ClusterGP:
    """Cholesky twin (`cggp/models.py:176-276`) -- a second oracle for CGGP."""

    def __init__(self, kernel, noise_variance, Z, pseudo_u=None, cluster_counts=None,
                 num_data=None):
        self.kernel = kernel
        dt = kernel.dtype
        self.noise_variance = dt.type(noise_variance)
        self.Z = np.asarray(Z, dtype=dt)
        M = self.Z.shape[0]
        self.pseudo_u = np.zeros((M, 1), dt) if pseudo_u is None else np.asarray(pseudo_u, dt).reshape(M, 1)
        self.cluster_counts = (np.ones((M, 1), dt) if cluster_counts is None
                               else np.asarray(cluster_counts, dt).reshape(M, 1))
        self.num_data = num_data

    @property
    def diag_variance(self):  # :226-228
        return self.noise_variance / self.cluster_counts

    def _KmmLambda(self):
        Kmm = Kuu(self.Z, self.kernel, jitter=0.0)
        return Kmm, add_diagonal(Kmm, self.diag_variance[:, 0])

    def prior_kl(self):  # :230-248
        Kmm, K = self._KmmLambda()
        L = np.linalg.cholesky(K)
        a = np.linalg.solve(L.T, np.linalg.solve(L, self.pseudo_u))
        quad = np.sum((Kmm @ a) * a)
        trace = np.trace(np.linalg.solve(L.T, np.linalg.solve(L, Kmm)))
        logdet = np.sum(2.0 * np.log(np.diag(L)))
        const = np.sum(np.log(self.diag_variance))
        return 0.5 * (quad - trace + logdet - const)

    def predict_f(self, Xnew, full_cov=False):  # :250-276
        Kmm, K = self._KmmLambda()
        Kmn = Kuf(self.Z, self.kernel, Xnew)
        Knn = self.kernel.K(Xnew) if full_cov else self.kernel.K_diag(Xnew)
        L = np.linalg.cholesky(K)
        a = np.linalg.solve(L.T, np.linalg.solve(L, self.pseudo_u))
        A = np.linalg.solve(L, Kmn)
        if not full_cov:
            fvar = (Knn - np.sum(np.square(A), axis=0))[:, None]
        else:
            fvar = (Knn - A.T @ A)[None, ...]
        fmu = Kmn.T @ a
        return fmu, fvar

    def scale(self, batch_size):  # :163-169
        if self.num_data is not None:
            return self.kernel.dtype.type(self.num_data) / self.kernel.dtype.type(batch_size)
        return self.kernel.dtype.type(1.0)

    def elbo(self, data):  # :125-134
        x, y = data
        kl = self.prior_kl()
        f_mean, f_var = self.predict_f(x)
        var_exp = gaussian_variational_expectations(f_mean, f_var, y, self.noise_variance)
        return np.sum(var_exp) * self.scale(x.shape[0]) - kl

    def q_moments(self, full_cov=False):  # :171-173
        return self.predict_f(self.Z, full_cov=full_cov)


class CGGP(ClusterGP):
    """`cggp/models.py:279-354`: every (Kmm+Lambda)^-1 applied by CG."""

    def __init__(self, kernel, noise_variance, Z, conjugate_gradient=None, num_probes=5, **kw):
        super().__init__(kernel, noise_variance, Z, **kw)
        self.conjugate_gradient = conjugate_gradient or ConjugateGradient(1e-6)
        self.num_probes = num_probes

    def prior_kl(self, probes=None):  # :293-322
        Kmm, KmmLambda = self._KmmLambda()
        a = self.conjugate_gradient(KmmLambda, self.pseudo_u)  # :303
        if self.num_probes is None:
            trace = np.trace(self.conjugate_gradient(KmmLambda, Kmm))  # :304-306
        else:
            assert probes is not None and probes.shape == (Kmm.shape[0], self.num_probes)
            S = self.conjugate_gradient(KmmLambda, probes)  # :311
            Kp = Kmm @ probes  # :312
            trace = np.sum(S * Kp) / self.kernel.dtype.type(self.num_probes)  # :313-314
        quad = np.sum((Kmm @ a) * a)  # :316-317
        logdet = eval_logdet_forward(KmmLambda)  # :319 -> 0.0 (:46)
        const = np.sum(np.log(self.diag_variance))  # :321
        return 0.5 * (quad - trace + logdet - const)  # :322

    def predict_f(self, Xnew, full_cov=False):  # :324-354
        Kmm, KmmLambda = self._KmmLambda()
        Kmn = Kuf(self.Z, self.kernel, Xnew)
        Knn = self.kernel.K(Xnew) if full_cov else self.kernel.K_diag(Xnew)
        a = self.conjugate_gradient(KmmLambda, self.pseudo_u)  # :339
        W = self.conjugate_gradient(KmmLambda, Kmn)  # :340
        if not full_cov:
            fvar = (Knn - np.sum(Kmn * W, axis=0))[:, None]  # :343-345
        else:
            fvar = (Knn - Kmn.T @ W)[None, ...]  # :347-349
        fmu = Kmn.T @ a  # :351
        return fmu, fvar

    def elbo(self, data, probes=None):
        x, y = data
        kl = self.prior_kl(probes=probes)
        f_mean, f_var = self.predict_f(x)
        var_exp = gaussian_variational_expectations(f_mean, f_var, y, self.noise_variance)
        return np.sum(var_exp) * self.scale(x.shape[0]) - kl


# ---------------------------------------------------------------- SGPR (row S1)
class SGPR:
    """GPflow `gpflow.models.SGPR` (Titsias collapsed bound), two-Cholesky closed form.

    Reached in the reference through `sgpr_class` (`cggp/cli_utils.py:444-446`).
    Zero mean function (the reference never passes one).
    """

    def __init__(self, data, kernel, Z, noise_variance, jitter=1e-6):
        self.X = np.asarray(data[0], dtype=kernel.dtype)
        self.Y = np.asarray(data[1], dtype=kernel.dtype)
        self.kernel = kernel
        self.Z = np.asarray(Z, dtype=kernel.dtype)
        self.noise_variance = kernel.dtype.type(noise_variance)
        self.jitter = jitter

    def _common(self):
        kuf = Kuf(self.Z, self.kernel, self.X)
        kuu = Kuu(self.Z, self.kernel, jitter=self.jitter)
        sigma = np.sqrt(self.noise_variance)
        L = np.linalg.cholesky(kuu)
        A = np.linalg.solve(L, kuf) / sigma
        AAT = A @ A.T
        B = AAT + np.eye(A.shape[0], dtype=A.dtype)
        LB = np.linalg.cholesky(B)
        Aerr = A @ self.Y
        c = np.linalg.solve(LB, Aerr) / sigma
        return L, LB, A, AAT, c, sigma

    def predict_f(self, Xnew, full_cov=False):
        L, LB, A, AAT, c, sigma = self._common()
        Kus = Kuf(self.Z, self.kernel, Xnew)
        tmp1 = np.linalg.solve(L, Kus)
        tmp2 = np.linalg.solve(LB, tmp1)
        mean = tmp2.T @ c
        if full_cov:
            var = (self.kernel.K(Xnew) + tmp2.T @ tmp2 - tmp1.T @ tmp1)[None, ...]
        else:
            var = (self.kernel.K_diag(Xnew) + np.sum(np.square(tmp2), 0)
                   - np.sum(np.square(tmp1), 0))[:, None]
        return mean, var

    def elbo(self):
        L, LB, A, AAT, c, sigma = self._common()
        N = self.X.shape[0]
        s2 = self.noise_variance
        const = -0.5 * N * LOG2PI
        logdet = -np.sum(np.log(np.diag(LB))) - 0.5 * N * np.log(s2)
        quad = -0.5 * np.sum(np.square(self.Y)) / s2 + 0.5 * np.sum(np.square(c))
        trace = -0.5 * np.sum(self.kernel.K_diag(self.X)) / s2 + 0.5 * np.trace(AAT)
        return const + logdet + quad + trace


class SgprNormalOperator:
    """S = s2 (Kmm + jitter I) + Kmn Knm, applied as a product (the build's CG form of S1).

    `rmatmul(P)` = P @ S for row-vector batches [Bt,M] (S symmetric).  `shards`>1
    evaluates Kmn(Knm v) as a sum over contiguous row shards of X (the multi-GPU
    decomposition of SURVEY §8e) so shard-sum invariance can be tested on CPU.
    """

    def __init__(self, X, Z, kernel, noise_variance, jitter=0.0, shards=1):
        self.X = np.asarray(X, kernel.dtype)
        self.Z = np.asarray(Z, kernel.dtype)
        self.kernel = kernel
        self.s2 = kernel.dtype.type(noise_variance)
        self.Kmm = Kuu(self.Z, kernel, jitter=jitter)
        self.shape = (self.Z.shape[0], self.Z.shape[0])
        self.dtype = kernel.dtype
        self.shards = np.array_split(np.arange(self.X.shape[0]), shards)

    def matmul(self, V):  # V [M,R]
        out = self.s2 * (self.Kmm @ V)
        for idx in self.shards:
            Knm = self.kernel.K(self.X[idx], self.Z)
            out = out + Knm.T @ (Knm @ V)
        return out

    def rmatmul(self, P):
        return self.matmul(P.T).T

    def dense(self):
        Knm = self.kernel.K(self.X, self.Z)
        return self.s2 * self.Kmm + Knm.T @ Knm

    def diag(self):
        Knm = self.kernel.K(self.X, self.Z)
        return self.s2 * np.diagonal(self.Kmm) + np.sum(np.square(Knm), axis=0)


class SGPRCG:
    """SGPR predictive equations in normal-equation form, solved by CG (row S1).

    mean* = K*m S^-1 Kmn y ;  var* = k** - K*m Kmm^-1 Km* + s2 K*m S^-1 Km*
    with S = s2 (Kmm + jitter I) + Kmn Knm.  Equals `SGPR.predict_f` exactly in
    exact arithmetic; checked against it in tests/test_oracle.py.
    """

    def __init__(self, data, kernel, Z, noise_variance, cg, jitter=1e-6, shards=1):
        self.X, self.Y = np.asarray(data[0], kernel.dtype), np.asarray(data[1], kernel.dtype)
        self.kernel, self.Z = kernel, np.asarray(Z, kernel.dtype)
        self.s2 = kernel.dtype.type(noise_variance)
        self.cg = cg
        self.jitter = jitter
        self.op = SgprNormalOperator(self.X, self.Z, kernel, noise_variance, jitter, shards)

    def rhs(self):
        return self.kernel.K(self.Z, self.X) @ self.Y  # Kmn y  [M,1]

    def predict_f(self, Xnew):
        alpha = self.cg(self.op, self.rhs())  # S^-1 Kmn y
        Kms = Kuf(self.Z, self.kernel, Xnew)
        mean = Kms.T @ alpha
        Kmm_j = Kuu(self.Z, self.kernel, jitter=self.jitter)
        W1 = self.cg(Kmm_j, Kms)
        W2 = self.cg(self.op, Kms)
        var = (self.kernel.K_diag(Xnew) - np.sum(Kms * W1, 0) + self.s2 * np.sum(Kms * W2, 0))[:, None]
        return mean, var


# ---------------------------------------------------------------- Hutchinson
def rademacher(M, P, seed, dtype=np.float64):
    """Documented probe stream: numpy PCG64 -> +-1 (`2*rng.integers(0,2)-1`)."""
    rng = np.random.default_rng(seed)
    return (2 * rng.integers(0, 2, size=(M, P)) - 1).astype(dtype)


def hutchinson_trace(KmmLambda, Kmm, probes, cg):
    """`cggp/models.py:308-314`: tr(KL^-1 Kmm) ~= sum(CG(KL, Zp) * (Kmm Zp)) / P."""
    S = cg(KmmLambda, probes)
    return np.sum(S * (Kmm @ probes)) / probes.dtype.type(probes.shape[1])
