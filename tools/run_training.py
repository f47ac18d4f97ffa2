"""Time the F2 training step (ELBO + backward + Adam) at C2 sizes: N=100k, D=8, M=2048, batch 1000."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np, torch
from cggp import kernels, synthetic
from cggp.conjugate_gradient import ConjugateGradient
from cggp.models import ClusterGP
from cggp.optimize import oips_update_inducing_parameters
from cggp.training import TrainableCGGP, train_using_adam_and_update

N, D, M, dt, kname = synthetic.CONFIGS["C2"]
syn = synthetic.make_inputs(N, D, M, dt)
dev = torch.device("cuda:0")
X, y, Z = (torch.from_numpy(a).to(dev) for a in (syn.X, syn.y, syn.Z))
k0 = kernels.SquaredExponential(1.0, [1.0] * D)
_, means, counts = oips_update_inducing_parameters(ClusterGP(k0, 0.1, Z), (X, y), Z)
m = TrainableCGGP(k0, 0.1, Z, ConjugateGradient(1e-6, check_every=16), num_probes=5, pseudo_u=means,
                  cluster_counts=counts, num_data=N)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
train_using_adam_and_update((X, y), m, iterations=2, batch_size=bs, learning_rate=0.01)
torch.cuda.synchronize()
t = time.perf_counter()
losses = train_using_adam_and_update((X, y), m, iterations=10, batch_size=bs, learning_rate=0.01)
torch.cuda.synchronize()
print(f"batch {bs}: {(time.perf_counter() - t) / 10 * 1e3:.1f} ms per Adam step; loss {losses[0]:.1f} -> {losses[-1]:.1f}")
print("variance", m.kernel.variance_p.value, "noise", m.noise_p.value)
