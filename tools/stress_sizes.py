"""Largest-size sanity pass (not a benchmark): C4's full N on one GPU, a 16384^2 dense operator in all three
product regimes, each spot-checked against the oracle / a library product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np, torch
from cggp import kernels, ops, synthetic
from oracle import kernels as ok
dev = torch.device("cuda:0")
N, D, M = 10_000_000, 2, 8192
syn = synthetic.make_inputs(N, D, M, "float32", need_y=False)
X, Z = torch.from_numpy(syn.X).to(dev), torch.from_numpy(syn.Z).to(dev)
k = kernels.SquaredExponential(1.0, [1.0, 1.0])
v = torch.from_numpy(synthetic.make_vectors(M, 1, "float32")).to(dev)
torch.cuda.synchronize(); t = time.perf_counter()
u = ops.knm_matvec(k.spec(D), X, Z, v)
w = ops.kmn_matvec(k.spec(D), X, Z, u)
torch.cuda.synchronize(); print(f"C4 full N on one GPU: K_nm v then K_mn u in {time.perf_counter()-t:.3f} s")
rows = np.random.default_rng(0).integers(0, N, 32)
ko = ok.Kernel("se", 1.0, np.ones(D))
ref = ko.K(syn.X[rows].astype(np.float64), syn.Z.astype(np.float64)) @ v.cpu().numpy().astype(np.float64)
print("  spot rows rel err", float(np.max(np.abs(u.cpu().numpy()[rows] - ref)) / np.max(np.abs(ref))), "finite", bool(torch.isfinite(w).all()))
idx = ops.nearest_center(k.spec(D), X, Z, return_distance=False)
s, c = ops.cluster_stats(idx, u, M)
print("  assignment + stats: counts sum", int(c.sum().item()), "of", N)
del X, u, w, idx
n = 16384
A = torch.randn(n, n, dtype=torch.float64, device=dev); A = A + A.t()
for Bt in (1, 64, 1024):
    P = torch.randn(Bt, n, dtype=torch.float64, device=dev)
    out = ops.symm_matmul(A, P)
    ref = P @ A
    print(f"n={n} Bt={Bt}: rel err {float((out-ref).abs().max()/ref.abs().max()):.2e}")
print("ok")
