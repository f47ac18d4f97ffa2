import sys
sys.path[:0] = ["/root/repo", "/root/repo/conjugate-gradient-sparse-gp_amd"]
import numpy as np, torch
from cggp import kernels, synthetic
from cggp.conjugate_gradient import ConjugateGradient
from cggp.models import SGPR
from oracle import kernels as ok, models as om, extended as ox
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def rel(a, b): return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
N, D, M, dt, kname = synthetic.CONFIGS["C1"]
syn = synthetic.make_inputs(N, D, M, dt)
Xs = syn.X[::8]
ko = ok.Kernel("se", 1.0, np.ones(1))
r = om.SGPR((syn.X, syn.y), ko, syn.Z, 0.1, jitter=1e-6)
rmu, rvar = r.predict_f(Xs)
lmu, lvar = ox.sgpr_predict_se(syn.X, syn.y, syn.Z, Xs, 1.0, np.ones(1), 0.1, 1e-6)
lmu, lvar = lmu.astype(np.float64), lvar.astype(np.float64)
print("oracle64 vs longdouble: mean", rel(rmu, lmu), "var", rel(rvar, lvar))
for pre in (None, "auto"):
    for thr, cap in ((1e-15, 2500), (1e-20, 200), (1e-30, 50)):
        for ex in (0, 8):
            m = SGPR((T(syn.X), T(syn.y)), kernels.SquaredExponential(1.0, [1.0]), T(syn.Z), 0.1,
                     ConjugateGradient(thr, max_iterations=cap), jitter=1e-6, preconditioner=pre, explicit_rhs=ex)
            mu, var = m.predict_f(T(Xs))
            st = m.solver().last_stats
            print(f"C1 pre={pre} thr={thr} cap={cap} explicit={ex}: steps {int(st[0])} mean {rel(mu.cpu().numpy(), lmu):.2e} var {rel(var.cpu().numpy(), lvar):.2e} elbo {abs(m.elbo()-r.elbo())/abs(r.elbo()):.2e}", flush=True)
