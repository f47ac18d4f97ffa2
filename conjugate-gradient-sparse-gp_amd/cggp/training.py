"""Training step through CG (next row F2) -- host mirror of the reference's trainable path.

What the reference does under `tf.GradientTape` (`cggp/optimize.py:198-254`,
`train_using_adam_and_update`): minimise `-elbo(batch)` (`cggp/models.py:125-134`) of a CGGP model
with Adam over the kernel variance / lengthscales and the likelihood variance (Z and pseudo_u are
frozen, `models.py:219-220`), where every `(Kmm+Lambda)^-1` is the CG with its custom gradient
(`conjugate_gradient.py:100-118`) and `log|Kmm+Lambda|` enters only through `eval_logdet`'s
backward (`models.py:30-44`).

Here: the hyper-parameters are positive `Parameter`s (softplus, as gpflow.utilities.positive());
kernel blocks are an autograd node whose forward is `mgp_k_dense` and whose backward is the fused
`mgp_k_dense_vjp` reduction; the CG solves are `_CGFunction` (device loop forward and backward);
`eval_logdet` is the node of `cggp.models`; the remaining small dense algebra (`Kmm @ a`,
elementwise likelihood terms) is torch on the device so the tape is end to end.
"""

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .conjugate_gradient import ConjugateGradient
from .kernels import Stationary
from .models import eval_logdet, rademacher


class Parameter:
    """Positive parameter: value = softplus(raw) + lower (gpflow.utilities.positive())."""

    def __init__(self, value, lower=0.0):
        v = np.atleast_1d(np.asarray(value, dtype=np.float64)) - lower
        if np.any(v <= 0):
            raise ValueError("initial value must exceed the lower bound")
        raw = np.where(v > 30.0, v, np.log(np.expm1(np.minimum(v, 30.0))))
        self.raw = torch.tensor(raw, dtype=torch.float64, requires_grad=True)
        self.lower = float(lower)
        self.scalar = np.ndim(value) == 0

    def __call__(self):
        t = F.softplus(self.raw) + self.lower
        return t[0] if self.scalar else t

    @property
    def value(self):
        t = self().detach()
        return float(t) if self.scalar else t.tolist()


class _KBlock(torch.autograd.Function):
    """K = k(A, B) (+ jitter I): forward `mgp_k_dense`, backward `mgp_k_dense_vjp`."""

    @staticmethod
    def forward(ctx, variance, lengthscales, A, B, kind, jitter):
        D = A.shape[1]
        spec = ops.KernelSpec(kind, float(variance), [float(x) for x in lengthscales.reshape(-1)], D)
        ctx.spec, ctx.v_shape, ctx.l_shape = spec, variance.shape, lengthscales.shape
        ctx.save_for_backward(A, B)
        return ops.k_dense(spec, A, B, jitter=jitter)

    @staticmethod
    def backward(ctx, G):
        A, B = ctx.saved_tensors
        dvar, dls = ops.k_dense_vjp(ctx.spec, A, B, G.contiguous())
        gv = torch.tensor(dvar, dtype=torch.float64).reshape(ctx.v_shape)
        n_l = int(np.prod(ctx.l_shape)) if len(ctx.l_shape) else 1
        gl = torch.tensor(dls if n_l > 1 else [sum(dls)], dtype=torch.float64).reshape(ctx.l_shape)
        return gv, gl, None, None, None, None


class TrainableKernel:
    """A stationary kernel whose variance / lengthscales are `Parameter`s."""

    def __init__(self, kernel: Stationary):
        self.name = kernel.name
        self.variance_p = Parameter(kernel.variance)
        ls = kernel.lengthscales
        self.lengthscales_p = Parameter(ls if len(ls) > 1 else ls[0])

    def parameters(self):
        return [self.variance_p.raw, self.lengthscales_p.raw]

    def K(self, X, X2=None, jitter=0.0):
        X2 = X if X2 is None else X2
        ls = self.lengthscales_p()
        if ls.dim() == 0:
            ls = ls.reshape(1)
        return _KBlock.apply(self.variance_p(), ls, X, X2, self.name, float(jitter))

    def frozen(self):
        """Plain kernel with the current values (non-differentiable fast paths: predict, metrics)."""
        from . import kernels
        cls = {"se": kernels.SquaredExponential, "matern12": kernels.Matern12, "matern32": kernels.Matern32,
               "matern52": kernels.Matern52}[self.name]
        ls = self.lengthscales_p.value
        return cls(variance=self.variance_p.value, lengthscales=ls if isinstance(ls, list) else [ls])


class _LogdetFromSolution(torch.autograd.Function):
    """`eval_logdet` (`cggp/models.py:21-48`) when K^-1 Zp is already at hand: value 0.0, gradient
    (K^-1 Zp)(df Zp)^T / P (`:40-42`) -- the estimator the reference recomputes with a second CG."""

    @staticmethod
    def forward(ctx, matrix, solution, probes):
        ctx.save_for_backward(solution, probes)
        return torch.zeros((), dtype=matrix.dtype, device=matrix.device)

    @staticmethod
    def backward(ctx, df):
        lv, probes = ctx.saved_tensors
        return (lv @ (df * probes).t()) / probes.shape[1], None, None


class TrainableCGGP:
    """Differentiable `CGGP.elbo` (`cggp/models.py:125-134,293-354`).

    The log-det gradient reuses `K^-1 Zp` of the trace estimator's probe solve instead of running
    the reference's second probe solve in the backward pass (same estimator, one CG less per step).
    `fused_solves=True` additionally sends `pseudo_u`, `Kmn` and the probes through ONE device CG as
    columns of one right-hand side; measured slower at C2 sizes (72.9 vs 67.2 ms per Adam step)
    because the reference's "any column not converged" stopping rule then makes the 1000-column
    GEMM-regime solve run as long as the slowest (Rademacher) columns, so it is off by default."""

    def __init__(self, kernel, noise_variance, Z, conjugate_gradient=None, num_probes=5, *, pseudo_u, cluster_counts,
                 num_data=None, fused_solves=False, independent_logdet_probes=False):
        self.kernel = kernel if isinstance(kernel, TrainableKernel) else TrainableKernel(kernel)
        self.noise_p = Parameter(noise_variance)
        self.Z = Z
        self.pseudo_u = pseudo_u.reshape(-1, 1)
        self.cluster_counts = cluster_counts.reshape(-1, 1)
        self.conjugate_gradient = conjugate_gradient or ConjugateGradient(1e-6)
        self.num_probes = num_probes
        self.num_data = num_data
        self.probe_seed = 0
        self.fused_solves = bool(fused_solves)
        # True: the reference's estimator exactly -- `eval_logdet`'s backward draws its OWN Rademacher probes and
        # runs its own probe solve (`cggp/models.py:38-41`), independent of the trace estimator's (`:310`).
        # False (default): K^-1 Zp of the trace solve is reused -- same expectation, one CG less per step, but the
        # two estimates are then correlated
        self.independent_logdet_probes = bool(independent_logdet_probes)
        self.logdet_probe_seed = 1 << 20

    def parameters(self):
        return self.kernel.parameters() + [self.noise_p.raw]

    def elbo(self, data, probes=None):
        x, y = data
        dev, dt = self.Z.device, self.Z.dtype
        cg = self.conjugate_gradient
        s2 = self.noise_p().to(device=dev, dtype=dt)
        var_f = self.kernel.variance_p().to(device=dev, dtype=dt)
        Kmm = self.kernel.K(self.Z)  # :300 / :333
        lam = s2 / self.cluster_counts[:, 0]  # diag_variance, :226-228
        KL = Kmm + torch.diag(lam)  # add_diagonal, :301 / :337
        Kmn = self.kernel.K(self.Z, x)  # :334
        fused = self.fused_solves and not (self.num_probes is None and probes is None)
        reuse = fused
        if fused:
            if probes is None:
                probes = rademacher((Kmm.shape[0], self.num_probes), dt, dev, self.probe_seed)
                self.probe_seed += 1
            B = Kmn.shape[1]
            sol = cg(KL, torch.cat([self.pseudo_u, Kmn, probes], dim=1))  # :303, :339, :311 in one solve
            a, W, S = sol[:, :1], sol[:, 1:1 + B], sol[:, 1 + B:]
        else:
            a = cg(KL, self.pseudo_u)  # :303 / :339
            W = cg(KL, Kmn)  # :340
        fvar = (var_f - (Kmn * W).sum(dim=0))[:, None]  # :343-345
        fmu = Kmn.t() @ a  # :351
        var_exp = -0.5 * math.log(2.0 * math.pi) - 0.5 * torch.log(s2) - 0.5 * ((y - fmu) ** 2 + fvar) / s2
        # prior_kl, :293-322
        if fused:
            trace = (S * (Kmm @ probes)).sum() / probes.shape[1]  # :312-314
        elif self.num_probes is None and probes is None:
            trace = torch.diagonal(cg(KL, Kmm)).sum()  # :304-306
        else:
            if probes is None:
                probes = rademacher((Kmm.shape[0], self.num_probes), dt, dev, self.probe_seed)
                self.probe_seed += 1
            S = cg(KL, probes)  # :311
            trace = (S * (Kmm @ probes)).sum() / probes.shape[1]  # :312-314
            reuse = True
        quad = ((Kmm @ a) * a).sum()  # :316-317
        if reuse and self.independent_logdet_probes:
            P = probes.shape[1]
            own = rademacher((Kmm.shape[0], P), dt, dev, self.logdet_probe_seed)  # fresh draw, :38-39
            self.logdet_probe_seed += 1
            logdet = eval_logdet(KL, cg, P, own)  # second probe solve in the backward pass, :40-42
        elif reuse:
            logdet = _LogdetFromSolution.apply(KL, S.detach(), probes)  # :319 with K^-1 Zp reused
        else:
            logdet = eval_logdet(KL, cg, self.num_probes if probes is None else probes.shape[1], probes)  # :319
        const = torch.log(lam).sum()  # :321
        kl = 0.5 * (quad - trace + logdet - const)
        scale = 1.0 if self.num_data is None else float(self.num_data) / float(x.shape[0])  # :163-169
        return var_exp.sum() * scale - kl

    def training_loss(self, data, probes=None):
        return -self.elbo(data, probes=probes)

    def frozen_model(self):
        from .models import CGGP
        return CGGP(self.kernel.frozen(), self.noise_p.value, self.Z, self.conjugate_gradient,
                    num_probes=self.num_probes, pseudo_u=self.pseudo_u, cluster_counts=self.cluster_counts,
                    num_data=self.num_data)


def train_using_adam_and_update(data, model, iterations, batch_size, learning_rate, update_fn=None,
                                update_during_training=None, monitor=None, seed=0):
    """`cggp/optimize.py:198-254`: shuffled minibatches, one Adam step per iteration, optional
    inducing-parameter update after each step, monitor callback per iteration."""
    x, y = data
    n = x.shape[0]
    gen = torch.Generator().manual_seed(seed)
    opt = torch.optim.Adam(model.parameters(), lr=learning_rate)
    update_during_training = update_during_training and (update_fn is not None)

    if update_fn is not None:
        update_fn()
    if monitor is not None:
        monitor(0)
    perm, pos = torch.randperm(n, generator=gen), 0
    losses = []
    for iteration in range(iterations):
        if pos + batch_size > n:
            perm, pos = torch.randperm(n, generator=gen), 0
        idx = perm[pos:pos + batch_size].to(x.device)
        pos += batch_size
        opt.zero_grad()
        loss = model.training_loss((x[idx], y[idx]))
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if update_during_training:
            update_fn()
        if monitor is not None:
            monitor(iteration)
    return losses


def train_using_lbfgs_and_update(data, model, max_num_iters, update_fn=None, update_during_training=None,
                                 monitor=None, probe_seed=0):
    """`cggp/optimize.py:152-195`: full-batch L-BFGS (scipy `L-BFGS-B`, what `gpflow.optimizers.Scipy`
    drives) over the model's trainable parameters; `update_fn` / `monitor` are called before the
    first step and after every accepted step, as the reference's `step_callback`.

    L-BFGS needs a deterministic objective, so when the model estimates the trace / log-det terms
    with Hutchinson probes the SAME probes (drawn once from `probe_seed`) are used for every
    evaluation.  Returns scipy's `OptimizeResult` (None when `max_num_iters` is 0, as upstream)."""
    from scipy.optimize import minimize

    params = model.parameters()
    sizes = [p.numel() for p in params]
    x, y = data
    probes = None
    if model.num_probes is not None:
        probes = rademacher((model.Z.shape[0], model.num_probes), model.Z.dtype, model.Z.device, probe_seed)

    def assign(flat):
        off = 0
        with torch.no_grad():
            for p, n in zip(params, sizes):
                p.copy_(torch.from_numpy(flat[off:off + n].copy()).reshape(p.shape))
                off += n

    def value_and_grad(flat):
        assign(flat)
        for p in params:
            p.grad = None
        loss = model.training_loss((x, y), probes=probes)
        loss.backward()
        g = np.concatenate([p.grad.detach().cpu().numpy().reshape(-1) for p in params])
        return float(loss), g.astype(np.float64)

    state = {"iteration": 0}

    def internal_update_fn(iteration):
        if update_during_training and (update_fn is not None):
            update_fn()
        if monitor is not None:
            monitor(iteration)

    def callback(_xk):
        state["iteration"] += 1
        internal_update_fn(state["iteration"])

    internal_update_fn(0)
    if max_num_iters > 0:
        x0 = np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in params]).astype(np.float64)
        result = minimize(value_and_grad, x0, jac=True, method="L-BFGS-B", callback=callback,
                          options=dict(maxiter=int(max_num_iters)))
        assign(result.x)
        return result
    internal_update_fn(-1)
    if monitor is not None and hasattr(monitor, "close"):
        monitor.close()
    return None


def train_vanilla_using_lbfgs(data, model, clustering_fn, max_num_iters, probe_seed=0):
    """`cggp/optimize.py:127-150`: full-batch L-BFGS over the model's trainable parameters, no inducing-point
    update (`clustering_fn` is accepted and unused, as upstream)."""
    return train_using_lbfgs_and_update(data, model, max_num_iters, update_fn=None, update_during_training=False,
                                        probe_seed=probe_seed)


def train_vanilla_using_lbfgs_and_standard_ip_update(data, model, clustering_fn, max_num_iters, probe_seed=0):
    """`cggp/optimize.py:101-124`: as above, with `model.Z <- clustering_fn()` after every L-BFGS step (the
    reference's own comment notes that this converges to poor minima; kept for call compatibility)."""
    def update_fn():
        model.Z = clustering_fn().to(model.Z.dtype)

    return train_using_lbfgs_and_update(data, model, max_num_iters, update_fn=update_fn, update_during_training=True,
                                        probe_seed=probe_seed)
