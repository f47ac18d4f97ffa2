"""Cover-tree construction time, device-assisted vs host (csrc/covertree_dev.hip vs csrc/covertree.cpp), same inputs.
usage: python tools/time_covertree.py N D resolution [--host]   (--host: also time the host construction and compare)"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
import numpy as np
from cggp.covertree import CoverTree

N, D, res = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rng = np.random.default_rng(0)
x = rng.standard_normal((N, D))
y = np.sin(x[:, :1])
warnings.simplefilter("ignore")
CoverTree(None, (x[:2000], y[:2000]), spatial_resolution=res)  # warm: library, handle
t0 = time.perf_counter()
t = CoverTree(None, (x, y), spatial_resolution=res)
dt = time.perf_counter() - t0
print(f"N={N} D={D} resolution={res}: {t.built_on} construction {dt:.2f} s, levels {[len(l) for l in t.levels]}", flush=True)
if "--host" in sys.argv:
    t0 = time.perf_counter()
    th = CoverTree(None, (x, y), spatial_resolution=res, device=False)
    dth = time.perf_counter() - t0
    same = all(np.array_equal(a.rows, b.rows) and np.array_equal(a.point, b.point)
               for la, lb in zip(t.levels, th.levels) for a, b in zip(la, lb))
    print(f"  host construction {dth:.2f} s on one core; identical trees: {same}; speed-up {dth / dt:.1f}x", flush=True)
