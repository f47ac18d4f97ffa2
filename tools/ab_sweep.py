"""A/B of sweep-kernel variants in ONE process on one device (cdna_hip_programming.md rule 24): variants are
handle settings read from the environment (MGP_SWEEP_FAST, MGP_SWEEP_RPT, MGP_PF_TRIPS, MGP_PF_AHEAD), a fresh
libmgp handle is made per variant, rounds are interleaved; per launch HIP-event times of the K_nm.p and K_mn.u
sweeps of the C3 SGPR-CG step are reported as median / min over all rounds.
Usage: python tools/ab_sweep.py "name:VAR=val,VAR=val" ... [--config C3] [--rounds 5] [--steps 10]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd")]
import numpy as np, torch
from cggp import _hip, kernels, ops, synthetic
from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--config", default="C3")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
KEYS = ("MGP_SWEEP_FAST", "MGP_SWEEP_RPT", "MGP_PF_TRIPS", "MGP_PF_AHEAD", "MGP_SWEEP", "MGP_NOSPLIT_PER_CU", "MGP_SWEEP_RPT32", "MGP_SWEEP_TARGET", "MGP_SWEEP_WFOLD")
N, D, M, dt, kname = synthetic.CONFIGS[args.config]
syn = synthetic.make_inputs(N, D, M, dt)
dev = torch.device("cuda:0")
X, Z, y = (torch.from_numpy(a).to(dev) for a in (syn.X, syn.Z, syn.y))
kern = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32}[kname](1.0, [1.0] * D)
variants = []
for v in args.variants:
    name, _, kv = v.partition(":")
    variants.append((name, dict(p.split("=") for p in kv.split(",") if p)))
res = {name: {"knm": [], "kmn": [], "step": []} for name, _ in variants}
ref = None
for rnd in range(args.rounds):
    for name, env in variants:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        _hip._handles.clear()  # next get_handle() makes a handle that reads the environment again
        op = SgprNormalOperator(kern, X, Z, syn.noise_variance, jitter=1e-6)
        rhs = ops.kmn_matvec(kern.spec(D), X, Z, y).t().contiguous()
        hd = _hip.get_handle(dev)
        conjugate_gradient(op, rhs, None, 0.0, max_iterations=2, max_steps_cycle=10 ** 6, check_every=2)  # warm
        hd.check(hd.lib.mgp_profile_enable(hd.h, 1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sol, _ = conjugate_gradient(op, rhs, None, 0.0, max_iterations=args.steps, max_steps_cycle=10 ** 6,
                                    check_every=args.steps)
        e1.record()
        n = ctypes.c_int64(0)
        buf = (ctypes.c_double * (2 * args.steps + 8))()
        hd.check(hd.lib.mgp_profile_read_each(hd.h, buf, len(buf), ctypes.byref(n)))
        hd.check(hd.lib.mgp_profile_enable(hd.h, 0))
        d = np.array(buf[:n.value])
        res[name]["knm"] += list(d[0::2])
        res[name]["kmn"] += list(d[1::2])
        res[name]["step"].append(e0.elapsed_time(e1) / args.steps)
        if ref is None:
            ref = sol.clone()
        else:
            res[name]["maxdiff"] = max(res[name].get("maxdiff", 0.0), float((sol - ref).abs().max() / ref.abs().max()))
for name, _ in variants:
    r = res[name]
    print(f"{name:28s} knm med {np.median(r['knm']):.4f} min {np.min(r['knm']):.4f} | kmn med {np.median(r['kmn']):.4f} "
          f"min {np.min(r['kmn']):.4f} | step med {np.median(r['step']):.4f} ms | sol diff vs first {r.get('maxdiff', 0.0):.1e}")
