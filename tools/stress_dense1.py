"""Randomised stress of the two-launch dense CG (csrc/cg_dense1.hip): random n in [1024, 8192] (ragged and whole),
random step counts, identity / Jacobi, fp64 / fp32, with and without an initial solution -- every case against the
several-right-hand-side path (the same column twice: skinny product + fused update, other kernels entirely), which
tests/test_gpu_parity.py pins to the oracle.  Usage: python tools/stress_dense1.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np, torch
from cggp.conjugate_gradient import JacobiPreconditioner, conjugate_gradient

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for c in range(cases):
    n = int(rng.choice([1024, 2048, 4096, 8192])) if rng.random() < 0.3 else int(rng.integers(1024, 8193))
    dt = torch.float64 if rng.random() < 0.75 else torch.float32
    k = int(rng.integers(1, 12))
    jac = rng.random() < 0.4
    use_v0 = rng.random() < 0.3
    g = torch.Generator(device="cpu").manual_seed(int(rng.integers(1 << 30)))
    Q = torch.randn(n, 24, generator=g, dtype=torch.float64)
    d = 0.5 + torch.rand(n, generator=g, dtype=torch.float64) * (3.0 if jac else 1.0)
    A = ((Q @ Q.t()) / 24 + torch.diag(d)).to(dt).to(dev)
    b = torch.randn(1, n, generator=g, dtype=torch.float64).to(dt).to(dev)
    v0 = (0.01 * torch.randn(1, n, generator=g, dtype=torch.float64)).to(dt).to(dev) if use_v0 else None
    pre = JacobiPreconditioner() if jac else None
    s1, (k1, e1) = conjugate_gradient(A, b, v0, 0.0, pre, max_iterations=k, max_steps_cycle=k + 1)
    b2 = torch.cat([b, b], 0)
    v2 = torch.cat([v0, v0], 0) if use_v0 else None
    s2, (k2, e2) = conjugate_gradient(A, b2, v2, 0.0, pre, max_iterations=k, max_steps_cycle=k + 1)
    assert int(k1) == int(k2) == k, (n, k, int(k1), int(k2))
    rel = float((s1[0] - s2[0]).abs().max() / s2[0].abs().max())
    rele = abs(float(e1[0]) - float(e2[0])) / max(abs(float(e2[0])), 1e-300)
    tol = 1e-9 if dt == torch.float64 else 2e-3
    worst = max(worst, rel / tol)
    assert torch.isfinite(s1).all() and rel < tol and rele < (1e-6 if dt == torch.float64 else 5e-2), (n, dt, k, jac, use_v0, rel, rele)
    # converged solve: stopping quantity on the true residual
    if c % 6 == 0 and dt == torch.float64:
        thr = 1e-10
        s, (ks, es) = conjugate_gradient(A, b, None, thr, pre, max_iterations=n, max_steps_cycle=n + 1, check_every=16)
        r = b - s @ A
        assert 0.5 * float((r * r).sum()) <= thr * (1 + 1e-6) and int(ks) < n, (n, int(ks))
print(f"stress_dense1: {cases} cases ok, worst relative-to-tolerance {worst:.3f}")
