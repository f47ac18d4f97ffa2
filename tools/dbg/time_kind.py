"""Time K_nm.v / K_mn.u at C3's shape for a given kernel family (fast vs LDS-tile via MGP_SWEEP_FAST)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import torch
from cggp import kernels, ops, synthetic
name, D = sys.argv[1], int(sys.argv[2])
N, M = 1 << 20, 4096
syn = synthetic.make_inputs(N, D, M, "float64", need_y=False)
dev = torch.device("cuda:0")
X, Z = torch.from_numpy(syn.X).to(dev), torch.from_numpy(syn.Z).to(dev)
k = {"se": kernels.SquaredExponential, "matern12": kernels.Matern12, "matern32": kernels.Matern32, "matern52": kernels.Matern52}[name](1.0, [1.0] * D)
V = torch.from_numpy(synthetic.make_vectors(M, 1, "float64")).to(dev)
u = ops.knm_matvec(k.spec(D), X, Z, V); w = ops.kmn_matvec(k.spec(D), X, Z, u)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(5):
    u = ops.knm_matvec(k.spec(D), X, Z, V)
b.record(); torch.cuda.synchronize(); t1 = a.elapsed_time(b) / 5
a.record()
for _ in range(5):
    w = ops.kmn_matvec(k.spec(D), X, Z, u)
b.record(); torch.cuda.synchronize(); t2 = a.elapsed_time(b) / 5
print(f"{name} D={D} FAST={os.environ.get('MGP_SWEEP_FAST', 'default')}: knm {t1:.3f} ms kmn {t2:.3f} ms")
