"""Per-iteration time of the dense multi-right-hand-side CG (the 64-probe solves of models.py:308-314) at fixed
iteration count: n = 4096 by default, Bt right-hand sides, threshold 0 so that exactly `iters` steps run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import torch
from cggp.conjugate_gradient import conjugate_gradient
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Q = torch.randn(n, n, dtype=torch.float64, device=dev, generator=g)
A = Q @ Q.t() / n + torch.eye(n, dtype=torch.float64, device=dev)
B = torch.randn(Bt, n, dtype=torch.float64, device=dev, generator=g)
best = 1e9
for rep in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    sol, (k, err) = conjugate_gradient(A, B, None, 0.0, max_iterations=iters, max_steps_cycle=iters + 1, check_every=iters)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t) / iters)
print(f"dense CG n={n} Bt={Bt}: {best * 1e6:.1f} us per iteration ({int(k)} steps); checksum {float(sol.abs().sum()):.12e}")
