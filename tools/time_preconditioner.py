import sys, time
sys.path.insert(0, "/root/repo/conjugate-gradient-sparse-gp_amd")
import numpy as np, torch
from cggp import kernels, ops, synthetic
from cggp.conjugate_gradient import SgprNormalOperator, SubsampledNormalPreconditioner
N, D, M, dt, kname = synthetic.CONFIGS["C3"]
syn = synthetic.make_inputs(N, D, M, dt)
dev = torch.device("cuda:0")
X, Z = (torch.from_numpy(a).to(dev) for a in (syn.X, syn.Z))
kern = kernels.SquaredExponential(1.0, [1.0] * D)
op = SgprNormalOperator(kern, X, Z, 0.1, jitter=1e-6)
def timed(label, fn, reps=3):
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        print(f"{label}: {1e3*(time.perf_counter()-t):.1f} ms", flush=True)
    return r
timed("whole preconditioner", lambda: SubsampledNormalPreconditioner(op, 16))
sel = torch.randperm(N)[:16*M].to(dev)
Xs = timed("gather rows", lambda: X[sel].contiguous())
G = timed("kmn_knm 65536 rows", lambda: ops.kmn_knm(op.spec, Xs, Z))
P = 0.1 * op.Kmm + 16.0 * G; P = 0.5 * (P + P.t())
L = timed("cholesky_ex", lambda: torch.linalg.cholesky_ex(P)[0])
Pi = timed("cholesky_inverse", lambda: torch.cholesky_inverse(L))
timed("inv via 2 triangular solves", lambda: (lambda Li: Li.t() @ Li)(torch.linalg.solve_triangular(L, torch.eye(M, dtype=L.dtype, device=dev), upper=False)))
timed("randperm host + to(dev)", lambda: torch.randperm(N, generator=torch.Generator().manual_seed(0))[:16*M].to(dev))
