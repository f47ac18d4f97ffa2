"""Runs a few launches of the fused sweep at a BASELINE config (for rocprofv3 passes)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))

import torch  # noqa: E402

from cggp import kernels, ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--launches", type=int, default=3)
ap.add_argument("--rhs", type=int, default=1)
args = ap.parse_args()
N, D, M, dt, kname = synthetic.CONFIGS[args.config]
syn = synthetic.make_inputs(N, D, M, dt, need_y=False)
dev = torch.device("cuda:0")
X, Z = torch.from_numpy(syn.X).to(dev), torch.from_numpy(syn.Z).to(dev)
V = torch.from_numpy(synthetic.make_vectors(M, args.rhs, dt)).to(dev)
kern = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32}[kname](1.0, [1.0] * D)
spec = kern.spec(D)
for _ in range(args.launches):
    u = ops.knm_matvec(spec, X, Z, V)
    w = ops.kmn_matvec(spec, X, Z, u)
torch.cuda.synchronize()
print("ok", float(w.abs().sum()))
