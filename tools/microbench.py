"""Per-kernel timings of the non-headline kernels (HIP events via torch on the current stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np, torch
from cggp import kernels, ops, synthetic

dev = torch.device("cuda:0")


def timeit(fn, reps=5, warm=2):
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


which = sys.argv[1:] or ["gemm", "gemv", "kdense", "contract", "sweeps"]
if "gemm" in which:
    for n, Bt, dt in [(2048, 4096, torch.float64), (4096, 4096, torch.float64), (4096, 1024, torch.float64), (2048, 1000, torch.float64), (2048, 300, torch.float64),
                      (4096, 64, torch.float64), (4096, 4096, torch.float32)]:
        A = torch.randn(n, n, dtype=dt, device=dev); A = A + A.t()
        P = torch.randn(Bt, n, dtype=dt, device=dev)
        ms = timeit(lambda: ops.symm_matmul(A, P))
        print(f"gemm n={n} Bt={Bt} {dt}: {ms:.3f} ms  {2.0*Bt*n*n/ms/1e9:.1f} TFLOP/s", flush=True)
        ms2 = timeit(lambda: P @ A)
        print(f"   torch matmul (library): {ms2:.3f} ms  {2.0*Bt*n*n/ms2/1e9:.1f} TFLOP/s", flush=True)
if "gemv" in which:
    for n, Bt in [(4096, 1), (4096, 2), (4096, 5), (4096, 8), (4096, 16), (4096, 64), (4096, 128), (8192, 1), (8192, 64), (2048, 1), (2048, 64)]:
        A = torch.randn(n, n, dtype=torch.float64, device=dev); A = A + A.t()
        P = torch.randn(Bt, n, dtype=torch.float64, device=dev)
        ms = timeit(lambda: ops.symm_matmul(A, P), reps=20)
        print(f"gemv n={n} Bt={Bt}: {ms*1e3:.1f} us  {8.0*n*n/ms/1e6:.0f} GB/s", flush=True)
if "stream" in which:  # how fast can one read-only pass go?  (torch reductions as the yardstick)
    for mb in (134, 537, 2147):
        x = torch.randn(mb * 1000 * 1000 // 8, dtype=torch.float64, device=dev)
        ms = timeit(lambda: x.sum(), reps=20)
        print(f"stream sum {mb} MB: {ms*1e3:.1f} us  {x.numel()*8/ms/1e6:.0f} GB/s", flush=True)
    for n in (4096, 8192, 16384):
        A = torch.randn(n, n, dtype=torch.float64, device=dev)
        P = torch.randn(1, n, dtype=torch.float64, device=dev)
        ms = timeit(lambda: ops.symm_matmul(A, P), reps=20)
        print(f"gemv n={n}: {ms*1e3:.1f} us  {8.0*n*n/ms/1e6:.0f} GB/s", flush=True)
        ms = timeit(lambda: A @ P[0], reps=20)
        print(f"   torch mv: {ms*1e3:.1f} us  {8.0*n*n/ms/1e6:.0f} GB/s", flush=True)
if "cg64" in which:  # per-iteration cost of the 64-probe CG at C3's M
    from cggp.conjugate_gradient import conjugate_gradient
    n = 4096
    Z = torch.randn(n, 8, dtype=torch.float64, device=dev)
    A = kernels.SquaredExponential(1.0, [1.0] * 8).K(Z) + 0.1 * torch.eye(n, dtype=torch.float64, device=dev)
    for Bt in (1, 8, 64, 128, 1024):
        B = torch.randn(Bt, n, dtype=torch.float64, device=dev)
        k = 200
        ms = timeit(lambda: conjugate_gradient(A, B, None, 0.0, max_iterations=k, max_steps_cycle=k + 1, check_every=k), reps=3, warm=1)
        print(f"cg n={n} Bt={Bt}: {ms/k*1e3:.1f} us/iteration", flush=True)
if "kuu" in which:
    Z = torch.randn(4096, 8, dtype=torch.float64, device=dev)
    kk = kernels.SquaredExponential(1.0, [1.0] * 8)
    lam = torch.rand(4096, dtype=torch.float64, device=dev)
    ms = timeit(lambda: kernels.Kuu(Z, kk, jitter=0.0, diag_add=lam))
    print(f"Kuu M=4096 with diag_add: {ms*1e3:.1f} us", flush=True)
    ms = timeit(lambda: kk.K(Z))
    print(f"K(Z) M=4096: {ms*1e3:.1f} us", flush=True)
if "kdense" in which:
    for M, D in [(4096, 8), (8192, 2)]:
        Z = torch.randn(M, D, dtype=torch.float64, device=dev)
        k = kernels.SquaredExponential(1.0, [1.0] * D)
        ms = timeit(lambda: k.K(Z))
        print(f"k_dense M={M} D={D}: {ms*1e3:.1f} us ({8.0*M*M/ms/1e6:.0f} GB/s written)", flush=True)
if "contract" in which:
    for cfg in ["C2", "C3", "C5"]:
        N, D, M, dt, kname = synthetic.CONFIGS[cfg]
        syn = synthetic.make_inputs(N, D, M, dt, need_y=False)
        X, Z = torch.from_numpy(syn.X).to(dev), torch.from_numpy(syn.Z).to(dev)
        k = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32}[kname](1.0, [1.0] * D)
        ms = timeit(lambda: ops.kmn_knm(k.spec(D), X, Z), reps=2, warm=1)
        print(f"kmn_knm {cfg}: {ms:.1f} ms  {2.0*N*M*M/ms/1e9:.1f} TFLOP/s (full 2NM^2 count; half executed)", flush=True)
if "sweeps" in which:
    for cfg in ["C2", "C3", "C5"]:
        N, D, M, dt, kname = synthetic.CONFIGS[cfg]
        syn = synthetic.make_inputs(N, D, M, dt, need_y=False)
        X, Z = torch.from_numpy(syn.X).to(dev), torch.from_numpy(syn.Z).to(dev)
        k = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32}[kname](1.0, [1.0] * D)
        for R in (1, 2, 4, 8):
            V = torch.randn(M, R, dtype=X.dtype, device=dev)
            W = torch.randn(N, R, dtype=X.dtype, device=dev)
            ms = timeit(lambda: ops.knm_matvec(k.spec(D), X, Z, V))
            ms2 = timeit(lambda: ops.kmn_matvec(k.spec(D), X, Z, W))
            print(f"sweep {cfg} R={R}: knm {ms:.3f} ms kmn {ms2:.3f} ms  ({N*M/ms/1e6:.2f} Gpair/s)", flush=True)
    N, D, M = 1_250_000, 2, 8192
    syn = synthetic.make_inputs(N, D, M, "float32", need_y=False)
    X, Z = torch.from_numpy(syn.X).to(dev), torch.from_numpy(syn.Z).to(dev)
    k = kernels.SquaredExponential(1.0, [1.0] * D)
    V = torch.randn(M, 1, dtype=X.dtype, device=dev)
    ms = timeit(lambda: ops.knm_matvec(k.spec(D), X, Z, V))
    print(f"sweep C4-shard fp32 N={N}: knm {ms:.3f} ms ({N*M/ms/1e6:.2f} Gpair/s)", flush=True)
if "nearest" in which:
    for cfg in ["C2", "C3"]:
        N, D, M, dt, kname = synthetic.CONFIGS[cfg]
        syn = synthetic.make_inputs(N, D, M, dt)
        X, Z, y = (torch.from_numpy(a).to(dev) for a in (syn.X, syn.Z, syn.y))
        k = kernels.SquaredExponential(1.0, [1.0] * D)
        ms = timeit(lambda: ops.nearest_center(k.spec(D), X, Z, distance_type="sqeuclidean", return_distance=False), reps=5)
        idx = ops.nearest_center(k.spec(D), X, Z, distance_type="sqeuclidean", return_distance=False)
        ms2 = timeit(lambda: ops.cluster_stats(idx, y, M, method="sweep"), reps=5)
        ms3 = timeit(lambda: ops.cluster_stats(idx, y, M, method="sorted"), reps=5)
        ms4 = timeit(lambda: ops.cluster_stats(idx, X, M, method="sorted"), reps=5)
        print(f"nearest_center {cfg}: {ms:.3f} ms ({N*M/ms/1e6:.0f} Gpair/s); cluster_stats sweep {ms2:.3f} ms, "
              f"sorted {ms3:.3f} ms, sorted with {D} columns {ms4:.3f} ms", flush=True)
