import os, sys, time
sys.path.insert(0, "/root/repo/conjugate-gradient-sparse-gp_amd")
import numpy as np, torch
from cggp import kernels, ops, synthetic
from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient, SubsampledNormalPreconditioner
N, D, M, dt, kname = synthetic.CONFIGS["C4"]
N = 1250000
syn = synthetic.make_inputs(N, D, M, dt)
dev = torch.device("cuda:0")
X, Z, y = (torch.from_numpy(a).to(dev) for a in (syn.X, syn.Z, syn.y))
kern = kernels.SquaredExponential(1.0, [1.0] * D)
op = SgprNormalOperator(kern, X, Z, 0.1, jitter=1e-6)
rhs = ops.kmn_matvec(kern.spec(D), X, Z, y).t().contiguous()
b2 = 0.5 * float((rhs.double() ** 2).sum())
print("dtype", X.dtype, "0.5||b||^2", b2)
for rpi in (16, 64):
    pre = SubsampledNormalPreconditioner(op, rows_per_inducing=rpi)
    print("rows/ind", rpi, "jitter used", pre.jitter_used)
    for cyc in (1000, 16, 4):
        for cap in (16, 64, 256):
            sol, (steps, err) = conjugate_gradient(op, rhs, None, 1e-6, pre, max_iterations=cap, max_steps_cycle=cyc, check_every=16)
            r = rhs - op.rmatmul(sol)
            print(f"  cycle {cyc} cap {cap}: rec {float(err):.3e} true {0.5*float((r.double()**2).sum()):.3e} rel {np.sqrt(0.5*float((r.double()**2).sum())/b2):.2e}", flush=True)
