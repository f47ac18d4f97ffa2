"""CDGP prediction of many rows at a BASELINE config: per-batch CG vs one shared CG against I.

  python tools/run_predict_all.py C3 [rows] [batch]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np, torch
from cggp import kernels, synthetic
from cggp.conjugate_gradient import ConjugateGradient
from cggp.models import CGGP, ClusterGP
from cggp.optimize import oips_update_inducing_parameters

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
N, D, M, dt, kname = synthetic.CONFIGS[cfg]
syn = synthetic.make_inputs(N, D, M, dt)
dev = torch.device("cuda:0")
X, Z, y = (torch.from_numpy(a).to(dev) for a in (syn.X, syn.Z, syn.y))
kern = kernels.SquaredExponential(1.0, [1.0] * D)
_, means, counts = oips_update_inducing_parameters(ClusterGP(kern, 0.1, Z), (X, y), Z)
m = CGGP(kern, 0.1, Z, ConjugateGradient(1e-6, check_every=25), num_probes=None, pseudo_u=means,
         cluster_counts=counts, num_data=N)


def timed(label, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    dt_ = time.perf_counter() - t
    print(f"{label}: {dt_:.3f} s", flush=True)
    return r, dt_


Xs = X[:rows]
(mu1, var1), t1 = timed(f"per-batch CG, {rows} rows in batches of {batch}", lambda: m.predict_f_batched(Xs, batch))
(mu2, var2), t2 = timed(f"shared inverse (one {M}-RHS CG + one GEMM per batch), {rows} rows", lambda: m.predict_f_batched(Xs, batch, shared_inverse=True))
print("  inverse CG iterations", int(m.inverse_stats[0]))
print("  max |mu diff|", float((mu1 - mu2).abs().max()), " max |var diff|", float((var1 - var2).abs().max()), " min var", float(var2.min()))
(mu3, var3), t3 = timed(f"shared inverse, all {N} rows", lambda: m.predict_f_batched(X, batch, shared_inverse=True))
print(f"  projected per-batch CG for all {N} rows: {t1 * N / rows:.1f} s")
