"""A few launches of the NT GEMM (p @ A, Bt > 128 regime) for rocprofv3 passes: python tools/run_gemm.py [n] [Bt]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import torch
from cggp import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
A = torch.randn(n, n, dtype=torch.float64, device=dev); A = A + A.t()
P = torch.randn(Bt, n, dtype=torch.float64, device=dev)
for _ in range(6):
    out = ops.symm_matmul(A, P)
torch.cuda.synchronize()
print("ok", float(out[0, 0]))
