"""Random shapes through the sweeps and the symmetric products, each checked against a float64 numpy product.
Not a benchmark: a bounded search for shape-dependent faults (ragged tails, pad rows, slice counts)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np, torch
from cggp import kernels, ops
from oracle import kernels as ok
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
KINDS = {"se": kernels.SquaredExponential, "matern12": kernels.Matern12, "matern32": kernels.Matern32,
         "matern52": kernels.Matern52}
t0, n_sweep, n_dense, worst = time.time(), 0, 0, 0.0
while time.time() - t0 < budget:
    if rng.random() < 0.5:
        name = rng.choice(list(KINDS))
        D = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 31, 32]))
        N = int(rng.choice([1, 2, 63, 64, 65, 511, 513, 1023, 1025, 2049, 4097, 8193, 20001]))
        M = int(rng.choice([1, 2, 3, 63, 65, 255, 257, 511, 1025, 2047]))
        R = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17]))
        ls = rng.uniform(0.5, 2.0, D)
        var = float(rng.uniform(0.5, 2.0))
        k, ko = KINDS[name](var, list(ls)), ok.Kernel(name, var, ls)
        X, Z = rng.standard_normal((N, D)), rng.standard_normal((M, D))
        V, W = rng.standard_normal((M, R)), rng.standard_normal((N, R))
        K = ko.K(X, Z)
        a = ops.knm_matvec(k.spec(D), T(X), T(Z), T(V)).cpu().numpy()
        b = ops.kmn_matvec(k.spec(D), T(X), T(Z), T(W)).cpu().numpy()
        ra, rb = K @ V, K.T @ W
        tol = 1e-10 if name != "matern12" else 1e-8
        ea = np.max(np.abs(a - ra)) / max(np.max(np.abs(ra)), 1e-300)
        eb = np.max(np.abs(b - rb)) / max(np.max(np.abs(rb)), 1e-300)
        worst = max(worst, ea, eb)
        assert ea < tol and eb < tol, ("sweep", name, D, N, M, R, ea, eb)
        n_sweep += 1
    else:
        n = int(rng.choice([1, 3, 63, 64, 65, 255, 256, 257, 511, 1000, 1023, 1025, 2047, 3001]))
        Bt = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 47, 63, 64, 65, 127, 128, 129, 200, 513]))
        dt = torch.float64 if rng.random() < 0.75 else torch.float32
        A = rng.standard_normal((n, n)); A = A + A.T
        P = rng.standard_normal((Bt, n))
        out = ops.symm_matmul(T(A, dt), T(P, dt)).double().cpu().numpy()
        ref = P @ A
        e = np.max(np.abs(out - ref)) / max(np.max(np.abs(ref)), 1e-300)
        worst = max(worst, e) if dt == torch.float64 else worst
        assert e < (1e-11 if dt == torch.float64 else 3e-4), ("dense", n, Bt, dt, e)
        n_dense += 1
torch.cuda.synchronize()
print(f"fuzz seed {seed}: {n_sweep} sweep cases, {n_dense} dense cases in {time.time() - t0:.0f} s, worst fp64 rel err {worst:.2e}: ok")
