import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import torch
from cggp import kernels, ops, synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
N, D, M, dt, kname = synthetic.CONFIGS[cfg]
syn = synthetic.make_inputs(N, D, M, dt, need_y=False)
dev = torch.device("cuda:0")
X, Z = torch.from_numpy(syn.X).to(dev), torch.from_numpy(syn.Z).to(dev)
k = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32}[kname](1.0, [1.0] * D)
for _ in range(2):
    out = ops.kmn_knm(k.spec(D), X, Z)
torch.cuda.synchronize()
print("ok", float(out[0, 0]))
