"""How far does un-preconditioned / Jacobi / subsampled-normal PCG get on the C3 SGPR normal equations?

  python tools/sgpr_convergence.py C3 [eye|jacobi|sub:<rows_per_inducing>]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np, torch
from cggp import kernels, ops, synthetic
from cggp.conjugate_gradient import (SgprNormalOperator, conjugate_gradient, JacobiPreconditioner,
                                     SubsampledNormalPreconditioner)

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
N, D, M, dt, kname = synthetic.CONFIGS[cfg]
syn = synthetic.make_inputs(N, D, M, dt)
dev = torch.device("cuda:0")
X, Z, y = (torch.from_numpy(a).to(dev) for a in (syn.X, syn.Z, syn.y))
kern = kernels.SquaredExponential(1.0, [1.0] * D)
op = SgprNormalOperator(kern, X, Z, 0.1, jitter=1e-6)
rhs = ops.kmn_matvec(kern.spec(D), X, Z, y).t().contiguous()
print("0.5||b||^2 =", 0.5 * float((rhs * rhs).sum()))
mode = sys.argv[2] if len(sys.argv) > 2 else "eye"
pre = None
if mode == "jacobi":
    pre = JacobiPreconditioner()
elif mode.startswith("sub"):
    torch.cuda.synchronize()
    t = time.perf_counter()
    pre = SubsampledNormalPreconditioner(op, rows_per_inducing=int(mode.split(":")[1]) if ":" in mode else 16)
    torch.cuda.synchronize()
    print("preconditioner build", time.perf_counter() - t, "s, rows", pre.sample_rows)
print("preconditioner", pre)
for cap in ((100, 300, 1000, 3000) if pre is None or mode == 'jacobi' else (10, 20, 50, 100, 300, 1000)):
    t = time.perf_counter()
    sol, (steps, err) = conjugate_gradient(op, rhs, None, 1e-6, preconditioner=pre, max_iterations=cap, max_steps_cycle=cap + 1, check_every=50 if cap > 100 else 10)
    torch.cuda.synchronize()
    r = rhs - op.rmatmul(sol)
    print(cap, int(steps), float(err), 0.5 * float((r * r).sum()), time.perf_counter() - t, flush=True)
    if int(steps) < cap:
        break
