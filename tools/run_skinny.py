"""Repeats the skinny symmetric product out[Bt,n] = P[Bt,n] A[n,n] (the 64-probe CG's `p @ A`) for rocprofv3 passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import torch
from cggp import ops
n, Bt, reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 64, 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(n, n, dtype=torch.float64, device=dev, generator=g); A = A + A.t()
P = torch.randn(Bt, n, dtype=torch.float64, device=dev, generator=g)
if os.environ.get("SKINNY_DATA") == "zeros":  # power experiment: the same instruction stream on all-zero operands
    A.zero_(); P.zero_()
elif os.environ.get("SKINNY_DATA") == "ones":
    A.fill_(1.0); P.fill_(1.0)
for _ in range(3):
    out = ops.symm_matmul(A, P)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    out = ops.symm_matmul(A, P)
b.record()
torch.cuda.synchronize()
print(f"skinny n={n} Bt={Bt}: {a.elapsed_time(b) / reps * 1e3:.1f} us per product; check {float((out - P @ A).abs().max()):.2e}")
