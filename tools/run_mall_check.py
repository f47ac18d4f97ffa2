"""Is Kmm + Lambda served from the Infinity Cache (256 MiB, memory-side) after the first CG iteration?  The PMC
counters at the L2 / fabric boundary cannot tell (an Infinity-Cache hit and an HBM read are the same request to them),
so this is a timing experiment: the one-RHS upper-triangle product (the tile kernel of the dense CG iteration; 67 MB of
tiles at n = 4096) timed back to back -- the state inside a CG solve -- and with 1 GiB of other data streamed between
two calls, which evicts the matrix from every cache level.  Also the average fabric read latency in cycles from
TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ when run under rocprofv3 --pmc (tools/... see DESIGN.md 4.3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import torch
from cggp import ops

dev = torch.device("cuda:0")
flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)  # 1 GiB
for n in (2048, 4096, 5792, 8192):
    A = torch.randn(n, n, dtype=torch.float64, device=dev)
    A = A + A.t()
    P = torch.randn(1, n, dtype=torch.float64, device=dev)
    for _ in range(5):
        ops.symm_matmul(A, P)
    torch.cuda.synchronize()
    res = {}
    for mode in ("back to back", "1 GiB streamed in between"):
        ts = []
        for _ in range(30):
            if mode != "back to back":
                flush.sum()  # read-only: leaves no dirty lines whose write-back would compete with the product
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.symm_matmul(A, P)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    tri = 4.0 * n * (n + 64)
    print(f"n={n}: upper triangle {tri / 1e6:.0f} MB; product (tile kernel + slot reduce) median "
          f"{res['back to back']:.1f} us back to back ({tri / res['back to back'] / 1e6:.2f} TB/s), "
          f"{res['1 GiB streamed in between']:.1f} us after a 1 GiB flush ({tri / res['1 GiB streamed in between'] / 1e6:.2f} TB/s)",
          flush=True)
