import os, sys
sys.path.insert(0, "/root/repo/conjugate-gradient-sparse-gp_amd")
import numpy as np, torch
from cggp import kernels, ops, synthetic
from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient, SubsampledNormalPreconditioner
N, D, M, dt, kname = synthetic.CONFIGS["C4"]
mode = sys.argv[1] if len(sys.argv) > 1 else "slice"
if mode == "slice":
    syn = synthetic.make_inputs(N, D, M, dt)
else:
    syn = synthetic.make_inputs(1250000, D, M, dt)
dev = torch.device("cuda:0")
X, Z, y = (torch.from_numpy(a).to(dev) for a in (syn.X, syn.Z, syn.y))
ns = 1250000
Xs, ys = X[:ns].contiguous(), y[:ns].contiguous()
kern = kernels.SquaredExponential(1.0, [1.0] * D)
op = SgprNormalOperator(kern, Xs, Z, 0.1, jitter=1e-6)
rhs = ops.kmn_matvec(kern.spec(D), Xs, Z, ys).t().contiguous()
b2 = float((rhs.double() ** 2).sum())
print("mode", mode, "0.5||b||^2", 0.5 * b2, "rhs finite", bool(torch.isfinite(rhs).all()))
pre = SubsampledNormalPreconditioner(op, rows_per_inducing=16)
print("jitter used", pre.jitter_used, "Pinv finite", bool(torch.isfinite(pre.inverse).all()), float(pre.inverse.abs().max()))
for cyc, cap in ((4, 8), (4, 64), (10**6, 64)):
    sol, (steps, err) = conjugate_gradient(op, rhs, None, 1e-6, pre, max_iterations=cap, max_steps_cycle=cyc, check_every=16)
    r = rhs.double() - op.rmatmul(sol).double()
    print(cyc, cap, "steps", int(steps), "err", float(err), "sol finite", bool(torch.isfinite(sol).all()),
          "rel", np.sqrt(float((r * r).sum()) / b2), flush=True)
