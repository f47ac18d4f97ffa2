"""Batched preconditioned conjugate gradient -- host mirror of `cggp/conjugate_gradient.py`.

Same names, argument order and shapes as the reference:

* `conjugate_gradient(matrix, rhs[Bt,n], initial_solution[Bt,n], error_threshold,
  preconditioner=None, max_iterations=None, max_steps_cycle=100)
  -> (solution[Bt,n], (steps, error[Bt,1]))`                      (reference :24-32,120)
* `ConjugateGradient(error_threshold, preconditioner=None, max_iterations=None,
  max_steps_cycle=None)`; `__call__(matrix[n,n], rhs[n,Bt], initial_solution=None) -> [n,Bt]`
  (reference :160-212)
* `CGPreconditioner / EyePreconditioner / BlockPreconditioner`     (reference :125-157)

The loop itself (`cg_step`, the stopping rule, the breakdown guards, the residual refresh)
runs in libmgp (`csrc/cg.hip`); this module only validates, lays tensors out and wires the
custom gradient (reference :100-118).  Build-side additions, none replacing reference
behaviour: `matrix` may be a `LinearOperator` (matrix-free forms), `JacobiPreconditioner`,
`ConjugateGradient.solve_with_stats`, `min_float` / `check_every` knobs.
"""

import ctypes

import numpy as np
import torch

from . import _hip, ops

MIN_FLOAT = 1e-16  # reference :50


# --------------------------------------------------------------------------- operators
class LinearOperator:
    """Symmetric n x n operator the device CG can apply (`matrix` generalised)."""

    shape = None
    dtype = None
    device = None

    def _struct(self):
        """(MgpOperator, keepalive objects)"""
        raise NotImplementedError

    def rmatmul(self, P):
        """P [Bt,n] @ Op -> [Bt,n] (the layout of `state.p @ A`, reference :65)."""
        P = _hip.check_tensor(P, "P", dtype=self.dtype)
        if P.dim() != 2 or P.shape[1] != self.shape[0]:
            raise ValueError(f"P shape {tuple(P.shape)} does not match n={self.shape[0]}")
        out = torch.empty_like(P)
        if P.shape[0] == 0:
            return out
        hd = _hip.get_handle(self.device)
        st, keep = self._struct()
        hd.check(hd.lib.mgp_operator_apply(hd.h, ctypes.byref(st), _hip.ptr(P), P.shape[0], _hip.ptr(out)))
        del keep
        return out

    def matmul(self, V):
        """Op @ V [n,R] -> [n,R]."""
        return self.rmatmul(V.t().contiguous()).t().contiguous()

    def diag(self):
        raise NotImplementedError

    def dense(self):
        """Explicit [n,n] matrix (tests, small n)."""
        eye = torch.eye(self.shape[0], dtype=self.dtype, device=self.device)
        return self.rmatmul(eye)


class DenseOperator(LinearOperator):
    def __init__(self, matrix):
        matrix = _hip.check_tensor(matrix, "matrix")
        if matrix.dim() != 2 or matrix.shape[0] != matrix.shape[1]:
            raise ValueError(f"matrix must be [n,n], got {tuple(matrix.shape)}")
        self.A = matrix
        self.shape = tuple(matrix.shape)
        self.dtype = matrix.dtype
        self.device = matrix.device

    def _struct(self):
        st = _hip.MgpOperator()
        st.kind = _hip.OP_DENSE
        st.dtype = _hip.dtype_code(self.A)
        st.n = self.shape[0]
        st.A = self.A.data_ptr()
        return st, (self.A,)

    def diag(self):
        return self.A.diagonal().clone()

    def dense(self):
        return self.A


class KmmLambdaOperator(LinearOperator):
    """(k(Z,Z) + diag(lam)) applied matrix-free (row M2, matrix-free alternative)."""

    def __init__(self, kernel, Z, lam):
        self.Z = _hip.check_tensor(Z, "Z")
        self.lam = _hip.check_tensor(lam, "lam", dtype=self.Z.dtype).reshape(-1)
        M = self.Z.shape[0]
        if self.lam.shape[0] != M:
            raise ValueError("lam must have M entries")
        self.kernel = kernel
        self.spec = kernel.spec(self.Z.shape[1])
        self.shape = (M, M)
        self.dtype = self.Z.dtype
        self.device = self.Z.device

    def _struct(self):
        k = self.spec.struct(_hip.dtype_code(self.Z))
        st = _hip.MgpOperator()
        st.kind = _hip.OP_KMM_LAMBDA
        st.dtype = k.dtype
        st.n = self.shape[0]
        st.kernel = ctypes.pointer(k)
        st.Z = self.Z.data_ptr()
        st.M = self.shape[0]
        st.lam = self.lam.data_ptr()
        return st, (k, self.Z, self.lam)

    def diag(self):
        return self.lam + self.kernel.variance


class SgprNormalOperator(LinearOperator):
    """S = s2 (Kmm + jitter I) + K_mn K_nm over this rank's rows of X (row S1).

    K_nm is never materialised: each application is two fused N x M sweeps.  With
    `allreduce` set (see `parallel.make_allreduce`) the local [Bt,M] partial K_mn(K_nm p) is
    summed over ranks once per application; Kmm, Z and the CG state are replicated.
    """

    def __init__(self, kernel, X, Z, noise_variance, jitter=0.0, allreduce=None, max_rhs=1, kmm_rows=None):
        self.X = _hip.check_tensor(X, "X")
        self.Z = _hip.check_tensor(Z, "Z", dtype=self.X.dtype)
        if self.X.dim() != 2 or self.Z.dim() != 2 or self.X.shape[1] != self.Z.shape[1]:
            raise ValueError("X [N,D] and Z [M,D] must share D")
        self.kernel = kernel
        self.spec = kernel.spec(self.Z.shape[1])
        self.s2 = float(noise_variance)
        M = self.Z.shape[0]
        self.Kmm = ops.k_dense(self.spec, self.Z, self.Z, jitter=float(jitter))
        self.shape = (M, M)
        self.dtype = self.Z.dtype
        self.device = self.Z.device
        self.allreduce = allreduce
        # rows of Kmm whose s2*Kmm.p term this rank contributes before the all-reduce
        # (parallel.kmm_slab(M)); the slabs of all ranks must tile [0, M)
        self.kmm_rows = kmm_rows if allreduce is not None else None
        self._partial = None
        self._cb = None
        self.reserve(max_rhs)

    @property
    def _comm(self):
        """libmgp RCCL communicator when the exchange is native (parallel.AllReduce on backend nccl)."""
        return getattr(self.allreduce, "comm", None)

    def _world(self):
        w = getattr(self.allreduce, "world_size", None)
        if w is None:
            import torch.distributed as dist
            w = dist.get_world_size() if dist.is_initialized() else 1
        return int(w)

    def _ensure_partial(self, Bt):
        # hook path only (gloo rehearsal): the collective addresses this buffer as a torch tensor.
        # Bt*M partial + 1 agreement word (csrc/cg.hip: every rank must have taken part in a step)
        need = Bt * self.shape[0] + 1
        if self._partial is None or self._partial.numel() < need:
            self._partial = torch.empty((need,), dtype=self.dtype, device=self.device)
            buf = self._partial

            def _cb(ctx, ptr_, count, dtype_c, stream):
                try:
                    # libmgp enqueued the partial on `stream` (its handle's stream == torch's current one,
                    # _hip.get_handle); make that explicit for the collective
                    if stream:  # NULL = the legacy default stream, which is torch's default stream too
                        with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=buf.device)):
                            self.allreduce(buf[:count])
                    else:
                        with torch.cuda.stream(torch.cuda.default_stream(buf.device)):
                            self.allreduce(buf[:count])
                    return 0
                except Exception:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1

            self._cb = _hip.ALLREDUCE_FN(_cb)

    def reserve(self, Bt):
        if self.allreduce is not None and self._comm is None:
            self._ensure_partial(Bt)

    def _struct(self):
        k = self.spec.struct(_hip.dtype_code(self.Z))
        st = _hip.MgpOperator()
        st.kind = _hip.OP_SGPR
        st.dtype = k.dtype
        st.n = self.shape[0]
        st.kernel = ctypes.pointer(k)
        st.X = self.X.data_ptr() if self.X.shape[0] > 0 else None
        st.N = self.X.shape[0]
        st.Z = self.Z.data_ptr()
        st.M = self.shape[0]
        st.Kmm = self.Kmm.data_ptr()
        st.s2 = self.s2
        keep = (k, self.X, self.Z, self.Kmm)
        if self.allreduce is not None:
            if self._comm is not None:  # native RCCL: no callback, no staging buffer
                st.comm = self._comm.ptr
                keep += (self._comm,)
            else:
                st.allreduce = self._cb
                st.partial_buf = self._partial.data_ptr()
                st.world_size = self._world()
                keep += (self._partial, self._cb)
            if self.kmm_rows is not None:
                st.kmm_row_begin, st.kmm_row_end = int(self.kmm_rows[0]), int(self.kmm_rows[1])
            else:  # no slab given: rank 0 alone adds the replicated term
                import torch.distributed as dist
                st.kmm_row_begin, st.kmm_row_end = (0, self.shape[0]) if dist.get_rank() == 0 else (0, -1)
        return st, keep

    def rmatmul(self, P):
        self.reserve(P.shape[0])
        return super().rmatmul(P)

    def diag(self):
        """diag(S) = s2 diag(Kmm_j) + sum_i k(x_i, z_m)^2 (one fused sweep, summed over ranks)."""
        d = ops.kmn_sq_colsum(self.spec, self.X, self.Z)
        if self.allreduce is not None:
            self.allreduce(d)
        return d + self.s2 * self.Kmm.diagonal()


def as_operator(matrix):
    if isinstance(matrix, LinearOperator):
        return matrix
    if isinstance(matrix, torch.Tensor):
        return DenseOperator(matrix)
    raise TypeError("matrix must be a [n,n] CUDA tensor or a LinearOperator")


# --------------------------------------------------------------------------- preconditioners
class CGPreconditioner:
    """Protocol of reference :125-128: `__call__(vec [Bt,n], mat) -> (z, rz)`.

    The CG loop is device resident.  The preconditioners libmgp knows (Eye / Jacobi / Block / Dense)
    describe themselves through `_native(op)` and run inside its kernels.  ANY OTHER subclass -- a user's
    own `__call__`, as the reference allows -- is served through libmgp's callback kind: once per step the
    library hands the residual batch over, `__call__(r, mat)` is run on the solve's stream (torch ops), and
    its `z` goes back into the loop; `rz` is recomputed by the library exactly as `sum(z * r, -1)`."""

    def _native(self, op):
        return None

    def _native_for(self, op, Bt):
        nat = self._native(op)
        if nat is not None:
            return nat
        n = op.shape[0]
        r_buf = torch.empty((Bt, n), dtype=op.dtype, device=op.device)
        z_buf = torch.empty_like(r_buf)
        mat = op.A if isinstance(op, DenseOperator) else op  # what the reference passes as `mat` (:77,:89)

        def _cb(ctx, r_ptr, z_ptr, bt, nn, stream):
            try:
                st = torch.cuda.ExternalStream(stream, device=r_buf.device) if stream else \
                    torch.cuda.default_stream(r_buf.device)
                with torch.cuda.stream(st):
                    z = self(r_buf, mat)[0]
                    z_buf.copy_(z.reshape(z_buf.shape))
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        fn = _hip.PRECOND_FN(_cb)
        st = _hip.MgpPrecond()
        st.kind = _hip.PRE_CALLBACK
        st.apply = fn
        st.cb_r = r_buf.data_ptr()
        st.cb_z = z_buf.data_ptr()
        return st, (r_buf, z_buf, fn, mat)

    def __call__(self, vec, mat):
        raise NotImplementedError


class EyePreconditioner(CGPreconditioner):
    """reference :131-134: z = vec, rz = sum(vec^2, -1, keepdims)."""

    def _native(self, op):
        st = _hip.MgpPrecond()
        st.kind = _hip.PRE_EYE
        return st, ()

    def __call__(self, vec, mat):
        return vec, (vec * vec).sum(dim=-1, keepdim=True)


class JacobiPreconditioner(CGPreconditioner):
    """z = vec / diag(A) (build-side addition)."""

    def __init__(self, diagonal=None):
        self.diagonal = diagonal

    def _dinv(self, op):
        d = self.diagonal if self.diagonal is not None else op.diag()
        return (1.0 / d.reshape(-1)).to(op.dtype).contiguous()

    def _native(self, op):
        dinv = self._dinv(op)
        st = _hip.MgpPrecond()
        st.kind = _hip.PRE_JACOBI
        st.diag_inv = dinv.data_ptr()
        return st, (dinv,)

    def __call__(self, vec, mat):
        z = vec * self._dinv(as_operator(mat))[None, :]
        return z, (z * vec).sum(dim=-1, keepdim=True)


class BlockPreconditioner(CGPreconditioner):
    """Block-Jacobi with the reference's constructor `BlockPreconditioner(block_indices[nb,bs])`.

    The reference implementation (:137-157) gathers on the batch axis and never scatters back
    (shape-inconsistent with the solver, no call sites, untested); this is the intended
    operation -- per block solve A[idx,idx] z[idx] = vec[idx] -- and is parity-unpinned.
    Block inverses are formed once per solve on the host (nb small dense Cholesky problems).
    """

    def __init__(self, block_indices):
        bi = torch.as_tensor(block_indices)
        if bi.dim() != 2:
            raise ValueError("block_indices must be [num_blocks, block_size]")
        self.block_indices = bi.to(torch.int64)

    def _setup(self, op):
        A = op.dense()
        idx = self.block_indices.to(A.device)
        flat = idx.reshape(-1)
        if flat.numel() != torch.unique(flat).numel():
            raise ValueError("block indices must not overlap")
        if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= A.shape[0]):
            raise ValueError("block index out of range")
        blocks = A[idx[:, :, None], idx[:, None, :]].cpu().numpy().astype(np.float64)
        inv = np.empty_like(blocks)
        for k in range(blocks.shape[0]):
            L = np.linalg.cholesky(blocks[k])
            Li = np.linalg.inv(L)
            inv[k] = Li.T @ Li
        binv = torch.from_numpy(inv).to(device=A.device, dtype=A.dtype).contiguous()
        return idx.contiguous(), binv

    def _native(self, op):
        idx, binv = self._setup(op)
        st = _hip.MgpPrecond()
        st.kind = _hip.PRE_BLOCK
        st.block_size = idx.shape[1]
        st.num_blocks = idx.shape[0]
        st.block_index = idx.data_ptr()
        st.block_inv = binv.data_ptr()
        return st, (idx, binv)

    def __call__(self, vec, mat):
        idx, binv = self._setup(as_operator(mat))
        z = vec.clone()
        for k in range(idx.shape[0]):
            z[:, idx[k]] = vec[:, idx[k]] @ binv[k].t()
        return z, (z * vec).sum(dim=-1, keepdim=True)


class DensePreconditioner(CGPreconditioner):
    """z = vec @ Pinv for a symmetric positive-definite `Pinv` [n, n] held on the device
    (build-side addition; the product runs on the same symmetric-product kernels as `p @ A`)."""

    def __init__(self, inverse):
        self.inverse = _hip.check_tensor(inverse, "inverse")
        if self.inverse.dim() != 2 or self.inverse.shape[0] != self.inverse.shape[1]:
            raise ValueError("inverse must be a square [n, n] tensor")

    def _native(self, op):
        if self.inverse.shape[0] != op.shape[0] or self.inverse.dtype != op.dtype:
            raise ValueError(f"preconditioner is {tuple(self.inverse.shape)} {self.inverse.dtype}, "
                             f"operator is n={op.shape[0]} {op.dtype}")
        st = _hip.MgpPrecond()
        st.kind = _hip.PRE_DENSE
        st.dense_inv = self.inverse.data_ptr()
        return st, (self.inverse,)

    def __call__(self, vec, mat):
        z = ops.symm_matmul(self.inverse, vec)
        return z, (z * vec).sum(dim=-1, keepdim=True)


class SubsampledNormalPreconditioner(DensePreconditioner):
    """Dense preconditioner for the SGPR normal matrix S = s2 Kmm + Kmn Knm (`SgprNormalOperator`):

        P = s2 Kmm + (N / n_s) Ks^T Ks,   Ks = k(X[sample], Z),

    i.e. the same matrix with the N-row Gram term replaced by an n_s-row Monte-Carlo estimate
    (n_s = `rows_per_inducing` * M rows drawn without replacement, per rank from its own shard and
    all-reduced, so every rank holds the same P).  `P^-1` comes from one Cholesky on the device.
    Cost: one `kmn_knm` contraction over n_s rows -- a few CG steps' worth -- against a cut in
    the step count from O(cond) to tens (PCG section)."""

    def __init__(self, operator, rows_per_inducing=32, seed=0, jitter=0.0):
        if not isinstance(operator, SgprNormalOperator):
            raise TypeError("SubsampledNormalPreconditioner needs an SgprNormalOperator")
        X, Z = operator.X, operator.Z
        M, N_local = Z.shape[0], X.shape[0]
        world = 1
        if operator.allreduce is not None:  # the ranks the operator's own exchange spans (it may be a sub-group)
            world = int(getattr(operator.allreduce, "world_size", 0) or 0)
            if world < 1:
                import torch.distributed as dist
                world = dist.get_world_size() if dist.is_initialized() else 1
        n_s = min(N_local, max(1, (int(rows_per_inducing) * M + world - 1) // world))
        # shards hold different rows, so one seed gives independent samples per rank
        gen = torch.Generator(device=X.device).manual_seed(int(seed))
        sel = torch.randperm(N_local, generator=gen, device=X.device)[:n_s]  # on the device: 10 ms less
        G = ops.kmn_knm(operator.spec, X[sel].contiguous(), Z)  # Ks^T Ks  [M, M]
        tot = torch.tensor([float(n_s), float(N_local)], dtype=torch.float64, device=X.device)
        if operator.allreduce is not None:
            operator.allreduce(G.view(-1))
            operator.allreduce(tot)
        n_tot, N_tot = float(tot[0]), float(tot[1])
        # factorise in fp64 whatever the operator's dtype: P inherits cond(S) ~ cond(Kmm)^2
        P = operator.s2 * operator.Kmm.double() + (N_tot / n_tot) * G.double()
        P = 0.5 * (P + P.t())
        eye = torch.eye(M, dtype=P.dtype, device=P.device)
        # In a dtype narrower than fp64 the operator itself is only known to eps ||S||: eigenvalues of
        # P below that are noise, and P^-1 rounded to that dtype would not stay positive definite
        # (measured: the fp32 solve diverges).  Lift them to the rounding level -- nothing is lost.
        eps = float(torch.finfo(operator.dtype).eps)
        bump = float(jitter) if operator.dtype == torch.float64 else max(float(jitter), eps * float(P.diagonal().sum()))
        floor = eps * float(P.diagonal().mean())
        for _ in range(12):
            L, info = torch.linalg.cholesky_ex(P + bump * eye if bump else P)
            if int(info) == 0:
                break
            # not numerically positive definite (rounding of the Gram term in the operator's dtype):
            # lift the spectrum by a step relative to that rounding and retry
            bump = max(10.0 * bump, floor)
        else:
            raise RuntimeError("SubsampledNormalPreconditioner: P is not positive definite")
        self.jitter_used = bump
        Pinv = torch.cholesky_inverse(L)
        self.sample_rows = int(n_tot)
        super().__init__((0.5 * (Pinv + Pinv.t())).to(operator.dtype).contiguous())


# --------------------------------------------------------------------------- solver
def _solve_device(op, rhs, initial_solution, error_threshold, preconditioner, max_iterations,
                  max_steps_cycle, min_float, check_every):
    rhs = _hip.check_tensor(rhs, "rhs", dtype=op.dtype)
    if rhs.dim() != 2 or rhs.shape[1] != op.shape[0]:
        raise ValueError(f"rhs must be [Bt, n={op.shape[0]}], got {tuple(rhs.shape)}")
    Bt, n = rhs.shape
    v0 = None
    if initial_solution is not None:
        v0 = _hip.check_tensor(initial_solution, "initial_solution", dtype=op.dtype, shape=(Bt, n))
    if preconditioner is None:
        preconditioner = EyePreconditioner()
    if max_iterations is None:
        max_iterations = n  # reference :47-48
    max_iterations = int(max_iterations)
    max_steps_cycle = int(max_steps_cycle)
    if max_steps_cycle < 1:
        raise ValueError("max_steps_cycle must be >= 1")
    sol = torch.empty_like(rhs)
    err = torch.empty((Bt, 1), dtype=op.dtype, device=op.device)
    stats = _hip.MgpCgStats()
    if Bt > 0:
        if isinstance(op, SgprNormalOperator):
            op.reserve(Bt)
        hd = _hip.get_handle(op.device)
        st, keep = op._struct()
        pst, pkeep = preconditioner._native_for(op, Bt)
        hd.check(hd.lib.mgp_pcg_solve(
            hd.h, ctypes.byref(st), ctypes.byref(pst), _hip.ptr(rhs), _hip.ptr(v0), Bt,
            float(error_threshold), max_iterations, max_steps_cycle, float(min_float), int(check_every),
            _hip.ptr(sol), _hip.ptr(err), ctypes.byref(stats)))
        del keep, pkeep
    return sol, stats, err


class _CGFunction(torch.autograd.Function):
    """Custom gradient of reference :100-118: db = CG(A, dx) from zero, dA = -solution^T @ db.

    Shortcut: when every row of the incoming `dx` is a multiple of the same row of the forward
    right-hand side -- `dx_b = c_b rhs_b`, which is what a loss that touches the solution only through
    `sum(rhs * solution)` sends back (the predictive-variance term `sum(Kmn * W)` of the ELBO,
    `cggp/models.py:343`) -- then `c_b solution_b` solves `db A = dx` with residual `c_b r_b`, i.e.
    `0.5||r||^2 = c_b^2 err_b`.  It is returned as it stands only when that already meets the reference's
    absolute rule `<= error_threshold` for every row (the second solve would then stop at step 0 or
    land within the same tolerance); otherwise it is the INITIAL SOLUTION of the second solve, which
    then needs a few steps instead of a full solve.  `conjugate_gradient.backward_shortcuts` counts the
    first case, `backward_warm_starts` the second."""

    @staticmethod
    def forward(ctx, matrix, rhs, initial_solution, cfg):
        op = as_operator(matrix)
        sol, stats, err = _solve_device(op, rhs, initial_solution, *cfg)
        ctx.cfg = cfg
        ctx.matrix = matrix
        ctx.zero_start = initial_solution is None
        ctx.save_for_backward(sol, rhs, err)
        ctx.mark_non_differentiable(err)
        ctx.stats = stats
        return sol, err

    @staticmethod
    def backward(ctx, dx, _derr):
        sol, rhs, err = ctx.saved_tensors
        op = as_operator(ctx.matrix)
        dx = dx.contiguous()
        db = None
        warm = None
        if ctx.zero_start:
            rr = (rhs * rhs).sum(dim=1, keepdim=True)
            c = (dx * rhs).sum(dim=1, keepdim=True) / torch.where(rr > 0, rr, torch.ones_like(rr))
            tol = 1e-12 if dx.dtype == torch.float64 else 1e-5
            if bool(((dx - c * rhs).abs().max() <= tol * dx.abs().max()).item()):
                # err = 0.5 rz of the forward solve (= 0.5||r||^2 for the identity preconditioner; for
                # others recompute the true residual of c*sol below through the warm start)
                thr = float(ctx.cfg[0])
                plain = ctx.cfg[1] is None or isinstance(ctx.cfg[1], EyePreconditioner)
                if plain and bool(((c * c) * err <= thr).all().item()):
                    db = c * sol
                    conjugate_gradient.backward_shortcuts += 1
                else:
                    warm = (c * sol).contiguous()
                    conjugate_gradient.backward_warm_starts += 1
        if db is None:
            db, _, _ = _solve_device(op, dx, warm, *ctx.cfg)
        dA = None
        if isinstance(ctx.matrix, torch.Tensor) and ctx.needs_input_grad[0]:
            dA = -(sol.t() @ db)  # [n,Bt]x[Bt,n] library GEMM (rank-Bt update)
        return dA, db, None, None


def conjugate_gradient(matrix, rhs, initial_solution, error_threshold, preconditioner=None,
                       max_iterations=None, max_steps_cycle=100, *, min_float=MIN_FLOAT, check_every=10):
    """Solve `V A = B` for row batches (reference `conjugate_gradient`, :24-122).

    Returns `(solution [Bt,n], (steps, error [Bt,1]))` with `steps` an int32 scalar tensor and
    `error = 0.5 * rz_final` (reference :96-98,120).
    """
    cfg = (error_threshold, preconditioner, max_iterations, max_steps_cycle, min_float, check_every)
    needs_grad = torch.is_grad_enabled() and (
        (isinstance(matrix, torch.Tensor) and matrix.requires_grad) or rhs.requires_grad)
    if needs_grad:
        sol, err = _CGFunction.apply(matrix, rhs, initial_solution, cfg)
        steps = torch.tensor(-1, dtype=torch.int32)  # not tracked on the differentiable path
        return sol, (steps, err)
    op = as_operator(matrix)
    sol, stats, err = _solve_device(op, rhs, initial_solution, *cfg)
    steps = torch.tensor(stats.iterations, dtype=torch.int32)
    conjugate_gradient.last_stats = stats
    return sol, (steps, err)


conjugate_gradient.last_stats = None
conjugate_gradient.backward_shortcuts = 0
conjugate_gradient.backward_warm_starts = 0


class ConjugateGradient:
    """Callable facade of reference :160-212 (column layout in and out, stats dropped)."""

    def __init__(self, error_threshold, preconditioner=None, max_iterations=None, max_steps_cycle=None,
                 *, min_float=MIN_FLOAT, check_every=10):
        self.error_threshold = error_threshold
        if preconditioner is None:
            preconditioner = EyePreconditioner()
        self.preconditioner = preconditioner
        self.max_iterations = max_iterations
        self.max_steps_cycle = max_steps_cycle
        self.min_float = min_float
        self.check_every = check_every
        self.last_stats = None

    def solve_with_stats(self, matrix, rhs, initial_solution=None):
        """`stats_conjugate_gradient` of `cggp/paper_condition_wasserstein.py:262-294`."""
        rhs_t = rhs.t().contiguous()  # :183
        init_t = None if initial_solution is None else initial_solution.t().contiguous()  # :185-188
        n = matrix.shape[-1]
        max_iterations = self.max_iterations if self.max_iterations is not None else n  # :190-192
        max_steps_cycle = self.max_steps_cycle if self.max_steps_cycle is not None else max_iterations + 1  # :194-196
        sol, stats = conjugate_gradient(
            matrix, rhs_t, init_t, self.error_threshold, preconditioner=self.preconditioner,
            max_iterations=max_iterations, max_steps_cycle=max_steps_cycle, min_float=self.min_float,
            check_every=self.check_every)
        self.last_stats = stats
        return sol.t().contiguous(), stats  # :211

    def __call__(self, matrix, rhs, initial_solution=None):
        return self.solve_with_stats(matrix, rhs, initial_solution)[0]
