"""One right-hand side on a dense matrix (the reference's literal CG loop, conjugate_gradient.py:65-84): time of the
upper-triangle product alone and of a whole CG iteration, per n.  Variants are selected by environment switches
read at handle creation (one process per variant):
    MGP_TRI_FORM=0|1|2   tile kernel: round-1 LDS row sums | cross-lane reduce-scatter | the same, two tiles per workgroup
    MGP_CG_DENSE1=0|1   product (tile kernel + slot reduce) + fused update launch | the two-launch iteration of cg_dense1.hip
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import torch
from cggp import kernels, ops
from cggp.conjugate_gradient import conjugate_gradient

dev = torch.device("cuda:0")


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tag = f"TRI_FORM={os.environ.get('MGP_TRI_FORM', 'default')} CG_DENSE1={os.environ.get('MGP_CG_DENSE1', 'default')}"
for n in (int(a) for a in (sys.argv[1:] or ["2048", "4096", "8192", "4001"])):
    Z = torch.randn(n, 8, dtype=torch.float64, device=dev)
    A = kernels.SquaredExponential(1.0, [1.0] * 8).K(Z) + 0.1 * torch.eye(n, dtype=torch.float64, device=dev)
    P = torch.randn(1, n, dtype=torch.float64, device=dev)
    us = 1e3 * timeit(lambda: ops.symm_matmul(A, P))
    B = torch.randn(1, n, dtype=torch.float64, device=dev)
    k = 400
    ms = timeit(lambda: conjugate_gradient(A, B, None, 0.0, max_iterations=k, max_steps_cycle=k + 1, check_every=k),
                reps=5, warm=2)
    it = ms / k * 1e3
    print(f"[{tag}] n={n}: product {us:.1f} us ({4.0 * n * n / us / 1e6:.2f} TB/s on the upper triangle), "
          f"CG iteration {it:.1f} us ({4.0 * n * n / it / 1e6:.2f} TB/s)", flush=True)
