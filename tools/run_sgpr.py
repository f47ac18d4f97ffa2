"""SGPR end to end at a BASELINE config: alpha solve (matrix-free PCG), predict (explicit-S PCG), ELBO.

  python tools/run_sgpr.py C3 [test_rows]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np, torch
from cggp import kernels, synthetic
from cggp.conjugate_gradient import ConjugateGradient
from cggp.models import SGPR

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N, D, M, dt, kname = synthetic.CONFIGS[cfg]
syn = synthetic.make_inputs(N, D, M, dt)
dev = torch.device("cuda:0")
X, Z, y = (torch.from_numpy(a).to(dev) for a in (syn.X, syn.Z, syn.y))
kern = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32}[kname](1.0, [1.0] * D)


def timed(label, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print(f"{label}: {time.perf_counter() - t:.3f} s", flush=True)
    return r


m = SGPR((X, y), kern, Z, 0.1, ConjugateGradient(1e-6, check_every=8), jitter=1e-6)
timed("operator + preconditioner build", lambda: m.solver())
a = timed("alpha = S^-1 K_mn y (matrix-free PCG)", lambda: m.alpha())
print("  iterations", int(m.solver().last_stats[0]))
Xs = X[:B] + 0.01
mu, var = timed(f"predict_f on {B} rows (explicit S + dense PCG, {B} RHS)", lambda: m.predict_f(Xs))
print("  iterations", int(m.solver().last_stats[0]), " mean|mu|", float(mu.abs().mean()), " min var", float(var.min()))
mu2, var2 = timed("predict_f again (S cached)", lambda: m.predict_f(Xs))
e = timed("elbo (K_mn K_nm cached)", lambda: m.elbo())
print("  elbo", e)
rm = float(torch.sqrt(((mu - y[:B]) ** 2).mean()))
print("  rmse vs noisy targets on the perturbed training rows", rm)
