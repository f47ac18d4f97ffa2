"""Randomised stress of the register-resident dense CG with several right-hand sides (csrc/cg_dense1.hip: full-matrix
form for n <= 2048, super-blocks of the triangle above): random n in [1024, 4096] (ragged and whole), 1..8 columns of
very different size, random step counts, identity / Jacobi, fp64 / fp32, with and without an initial solution -- every
case against oracle/cg.py (the reference's loop restated, conjugate_gradient.py:44-122).
Usage: python tools/stress_dense_cols.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd")]
import numpy as np, torch
from oracle import cg as ocg
from cggp.conjugate_gradient import JacobiPreconditioner, conjugate_gradient

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for c in range(cases):
    n = int(rng.choice([1024, 2048, 2112, 3072, 4096])) if rng.random() < 0.3 else int(rng.integers(1024, 4097))
    Bt = int(rng.integers(1, 9))
    f32 = rng.random() < 0.2
    k = int(rng.integers(1, 10))
    jac = rng.random() < 0.4
    use_v0 = rng.random() < 0.3
    Q = rng.standard_normal((n, 20))
    A = Q @ Q.T / 20 + np.diag(0.5 + rng.random(n) * (3.0 if jac else 1.0))
    b = rng.standard_normal((Bt, n)) * (10.0 ** rng.integers(-2, 3, (Bt, 1)))
    v0 = 0.01 * rng.standard_normal((Bt, n)) if use_v0 else None
    dt = torch.float32 if f32 else torch.float64
    s, (ks, es) = conjugate_gradient(torch.from_numpy(A).to(dt).to(dev), torch.from_numpy(b).to(dt).to(dev),
                                     None if v0 is None else torch.from_numpy(v0).to(dt).to(dev), 0.0,
                                     JacobiPreconditioner() if jac else None, max_iterations=k, max_steps_cycle=k + 1)
    o, (ko, eo) = ocg.conjugate_gradient(A, b, np.zeros((Bt, n)) if v0 is None else v0, 0.0,
                                         ocg.JacobiPreconditioner() if jac else None, max_iterations=k, max_steps_cycle=k + 1)
    assert int(ks) == k == ko, (n, Bt, k, int(ks), ko)
    got = s.cpu().numpy().astype(np.float64)
    tol = 5e-3 if f32 else 1e-9
    for col in range(Bt):
        rel = float(np.max(np.abs(got[col] - o[col])) / np.max(np.abs(o[col])))
        worst = max(worst, rel / tol)
        assert np.isfinite(got).all() and rel < tol, (n, Bt, col, k, jac, use_v0, f32, rel)
print(f"stress_dense_cols: {cases} cases ok, worst relative-to-tolerance {worst:.3f}")
