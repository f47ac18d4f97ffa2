"""Per-iteration timeline of the register-resident dense CG (csrc/cg_dense1.hip): MGP_D1_TRACE=<file> makes libmgp stamp
eight points of the first 64 iterations in workgroups 0 (a chunk owner) and 1.  usage: python tools/d1_trace.py out n ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
os.environ["MGP_D1_TRACE"] = out
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import torch
from cggp import kernels
from cggp.conjugate_gradient import conjugate_gradient
dev = torch.device("cuda:0")
for n in (int(a) for a in sys.argv[2:]):
    g = torch.Generator(device="cpu").manual_seed(n)
    Z = torch.randn(n, 8, generator=g, dtype=torch.float64).to(dev)
    A = kernels.SquaredExponential(1.0, [1.0] * 8).K(Z) + 0.1 * torch.eye(n, dtype=torch.float64, device=dev)
    B = torch.randn(int(os.environ.get("D1_TRACE_BT", "1")), n, generator=g, dtype=torch.float64).to(dev)
    for rep in range(2):  # the second solve is the warm one
        conjugate_gradient(A, B, None, 0.0, max_iterations=40, max_steps_cycle=41, check_every=40)
    torch.cuda.synchronize()
