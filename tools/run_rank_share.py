"""Per-rank share of C3's two sweeps (rows of X sharded 8 / 4 / 2 ways): K_nm.p and K_mn.u timed separately with the
streamed-set pack held (as inside mgp_pcg_solve), for A/B runs of the chunking switches (one process per variant:
the switches are read at handle creation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np, torch
from cggp import kernels, ops, synthetic
from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient

dev = torch.device("cuda:0")
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MGP_")) or "defaults"
N0, D, M, dt, kname = synthetic.CONFIGS["C3"]
syn = synthetic.make_inputs(N0, D, M, dt)
kern = kernels.SquaredExponential(1.0, [1.0] * D)
for rows in (int(a) for a in (sys.argv[1:] or ["131072", "262144", "524288", "1048576"])):
    X = torch.from_numpy(syn.X[:rows]).to(dev)
    y = torch.from_numpy(syn.y[:rows]).to(dev)
    Z = torch.from_numpy(syn.Z).to(dev)
    op = SgprNormalOperator(kern, X, Z, 0.1, jitter=1e-6, max_rhs=1)
    rhs = ops.kmn_matvec(kern.spec(D), X, Z, y).t().contiguous()
    k = 40

    def run():
        conjugate_gradient(op, rhs, None, 0.0, max_iterations=k, max_steps_cycle=k + 1, check_every=k)

    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        run()
    b.record()
    torch.cuda.synchronize()
    step_us = a.elapsed_time(b) / (3 * k) * 1e3
    pairs = rows * M
    ideal = pairs / 64 * 18.41 * 4 / 1024 / 2.4e9 * 1e6  # us per sweep at 100 % VALU issue, 2.4 GHz
    print(f"[{tag}] rows={rows}: CG step {step_us:.1f} us (two sweeps at 100% issue / 2.4 GHz: {2 * ideal:.1f} us -> "
          f"{2 * ideal / step_us:.3f} of the step)", flush=True)
