"""bench.py -- CG iterations/s of the matrix-free SGPR normal-equation solve on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU
over RCCL.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json `metric`: "CG iters/sec & K_nm.v achieved-HBM-GB/s, N=1M M=4096 D=8 fp64"):
config C3 -- N = 2^20 rows, D = 8, M = 4096, SE kernel, fp64, synthetic inputs of SURVEY §8(d).
A step is ONE iteration of the preconditioned CG loop of cggp/conjugate_gradient.py:64-85 on the
system  S alpha = K_mn y,  S = s2 (Kmm + jitter I) + K_mn K_nm  applied matrix-free:
one fused K_nm.p sweep, one fused K_mn.u sweep, the replicated dense Kmm.p product, one
all-reduce of the [1, M] partial (N > 1), and the fused vector update -- all inside libmgp.
The rows of X are sharded over the ranks (strong scaling: N is the TOTAL row count).
Inputs are resident in HBM before the timed region.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X datasheet (== fp64 matrix peak); MI355X_MICROARCH.md: 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
SURVEY_FLOPS_PER_PAIR = lambda D, R: 3 * D + 35 + 2 * R  # N M (3D + C_SE + 2R), C_SE = 35
# ALGORITHMIC flops per pair of the fused SE product, frozen at round 1's count so that rounds
# compare: distance D fma (2D) + exp2 on a reduced argument (3 add, 2 fma, mul, fma, scale = 11) + R accumulate fma
# (2R).  The kernel may execute fewer instructions than this (round 2 does); time is what is measured.
ALG_FLOPS_PER_PAIR = lambda D, R: 2 * D + 11 + 2 * R
# VALU wave-instructions the current kernels issue per pair, by element size (csrc/sweep.hip, fp64 SE fast
# kernel: D fma + 3 add + 2 fma + mul + fma + R fma = D + 7 + R fp64 and 2 integer -- PMC: 18.3 per pair at
# D = 8, R = 1 with the per-point overheads; fp32 SE: D fma + v_exp_f32 + R fma), and the cycles one
# wave-instruction holds a SIMD (fp64 16 lanes/clk -> 4; fp32 and 32-bit integer 32 lanes/clk -> 2 nominal;
# profiles/r02_valu_issue_probe.txt: next to fp64 work an integer instruction costs ~0.75 of an fp64 slot)
# fp32 (C4): D + R fp32 fma slots (v_fma_f32 holds a SIMD 2 cycles per wave, v_pk_fma_f32 4 cycles for two pairs) and one
# v_exp_f32, which holds it 8 cycles -- tools/micro/valu_issue.hip, profiles/r02_valu_issue_probe_fp32.txt: 16 fma32
# 9.8 ms, 16 pk_fma32 18.1 ms, 16 exp32 34.7 ms against 16 fma64 18.6 ms; mixes add up, nothing overlaps.  The second
# entry of each pair is "the other kind": 32-bit integer instructions for fp64, the transcendental for fp32.
VALU_INSTR_PER_PAIR = {8: lambda D, R: (D + 7 + R, 2), 4: lambda D, R: (D + R, 1)}  # (float, other)
NUM_SIMDS, MAX_CLOCK_HZ = 1024, 2.4e9
CYCLES_PER_WAVE_INSTR = {8: (4, 2), 4: (2, 8)}  # (float, other) by element size
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X datasheet (packed fp32); used when the config computes in fp32 (C4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=393216)  # ~13 s of host work on 16 cores
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--convergence-cap", type=int, default=0, help="iteration cap of the convergence leg (0 = M)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1 only: run the all-reduce hook on a 1-rank group (measures its fixed per-step cost)")
    ap.add_argument("--rows", type=int, default=0, help="override the TOTAL row count (0 = the config's)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): the config's N is the TOTAL row count, sharded over the ranks -- the "
                         "size BASELINE.json's metric is quoted on; weak: every rank holds the config's N rows")
    args = ap.parse_args()

    # Only the JSON line may reach stdout (RCCL prints a version banner there): park fd 1 on stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from cggp import _hip, kernels, ops, parallel, synthetic
    from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient

    N, D, M, dtype_name, kname = synthetic.CONFIGS[args.config]
    if args.rows:
        N = args.rows
    if args.scaling == "weak":
        N = N * world  # per-GPU work fixed: the job grows with the number of ranks
    tdtype = torch.float64 if dtype_name == "float64" else torch.float32
    esize = 8 if dtype_name == "float64" else 4
    syn = synthetic.make_inputs(N, D, M, dtype_name)
    lo, hi = parallel.shard_bounds(N, world, rank)
    X = torch.from_numpy(syn.X[lo:hi]).to(dev)
    y = torch.from_numpy(syn.y[lo:hi]).to(dev)
    Z = torch.from_numpy(syn.Z).to(dev)
    n_local = hi - lo
    kern = {"se": kernels.SquaredExponential, "matern32": kernels.Matern32}[kname](
        variance=syn.variance, lengthscales=syn.lengthscales)
    spec = kern.spec(D)
    allreduce = parallel.make_allreduce(force=args.force_collective)
    op = SgprNormalOperator(kern, X, Z, syn.noise_variance, jitter=1e-6, allreduce=allreduce, max_rhs=1,
                            kmm_rows=parallel.kmm_slab(M))
    rhs = ops.kmn_matvec(spec, X, Z, y)  # K_mn y  [M,1]
    if allreduce is not None:
        allreduce(rhs.view(-1))
    rhs_rows = rhs.t().contiguous()  # [1, M]

    def barrier():
        if world > 1:
            dist.barrier()

    def run_steps(k):
        # error_threshold 0 is never met, so exactly k iterations run (device gating keeps count)
        _, (steps, _) = conjugate_gradient(op, rhs_rows, None, 0.0, max_iterations=k, max_steps_cycle=k + 1,
                                           check_every=k)
        assert int(steps) == k, (int(steps), k)

    if args.warmup > 0:
        run_steps(args.warmup)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if args.backend != "nccl":
            tmax = tmax.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- dominant kernel: the fused sweep (K_nm.p and K_mn.u are the same kernel symbol),
    # HIP events bracketing every launch of it over a second pass of the same K steps
    hd = _hip.get_handle(dev)
    hd.check(hd.lib.mgp_profile_enable(hd.h, 1))
    run_steps(args.steps)
    import ctypes
    launches = ctypes.c_int64(0)
    each = (ctypes.c_double * (2 * args.steps + 8))()
    hd.check(hd.lib.mgp_profile_read_each(hd.h, each, len(each), ctypes.byref(launches)))
    hd.check(hd.lib.mgp_profile_enable(hd.h, 0))
    durs = np.array(each[:min(launches.value, len(each))], dtype=np.float64)
    sweep_ms = float(durs.mean()) if durs.size else float("nan")
    sweep_pct = {"median_ms": float(np.median(durs)), "p10_ms": float(np.percentile(durs, 10)),
                 "p90_ms": float(np.percentile(durs, 90))} if durs.size else None
    R = 1
    pairs_launch = float(n_local) * M
    flops_launch = pairs_launch * ALG_FLOPS_PER_PAIR(D, R)
    n_fl, n_int = VALU_INSTR_PER_PAIR[esize](D, R)
    c_fl, c_int = CYCLES_PER_WAVE_INSTR[esize]
    issue_s = (pairs_launch / 64.0) * (n_fl * c_fl + n_int * c_int) / (NUM_SIMDS * MAX_CLOCK_HZ)
    bytes_launch = float(esize) * (n_local * D + M * D + M * R + n_local * R)
    ach_tflops = flops_launch / (sweep_ms * 1e-3) / 1e12
    vector_peak = FP64_VECTOR_PEAK_TFLOPS if esize == 8 else FP32_VECTOR_PEAK_TFLOPS
    flop_model_exact = kname == "se"
    ach_gbps = bytes_launch / (sweep_ms * 1e-3) / 1e9
    equiv_gemv_gbps = float(esize) * n_local * M / (sweep_ms * 1e-3) / 1e9

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{args.config}/gpus{world}", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # ---- bounded convergence report (informational): real stopping rule, thr = 1e-6
    conv = None
    if rank == 0 or world > 1:
        cap = args.convergence_cap or M  # reference default cap: max_iterations = n (conjugate_gradient.py:190-192)
        tc = time.perf_counter()
        sol, (steps, err) = conjugate_gradient(op, rhs_rows, None, 1e-6, max_iterations=cap, max_steps_cycle=cap + 1,
                                               check_every=64)
        torch.cuda.synchronize()
        rres = rhs_rows - op.rmatmul(sol)
        conv = {"error_threshold": 1e-6, "iteration_cap": cap, "iterations": int(steps),
                "converged": bool(int(steps) < cap or float(err.max().item()) <= 1e-6),
                "half_rz_final": float(err.max().item()),
                "true_half_residual_sq": 0.5 * float((rres * rres).sum().item()),
                "half_rhs_sq": 0.5 * float((rhs_rows * rhs_rows).sum().item()),
                "seconds": time.perf_counter() - tc,
                "note": "absolute criterion 0.5||r||^2 <= 1e-6 of the reference, no preconditioner"}

    # ---- the same solve with the subsampled normal-equation preconditioner (build-side addition,
    # DESIGN.md 4.6): P = s2 Kmm + (N/n_s) Ks^T Ks from n_s = 16 M sampled rows, z = r @ P^-1
    pcg = None
    try:
        from cggp.conjugate_gradient import SubsampledNormalPreconditioner
        torch.cuda.synchronize()
        tb = time.perf_counter()
        pre = SubsampledNormalPreconditioner(op, rows_per_inducing=32, seed=0)
        torch.cuda.synchronize()
        t_build_cold = time.perf_counter() - tb  # includes the one-off load of the factorisation library
        tb = time.perf_counter()
        pre = SubsampledNormalPreconditioner(op, rows_per_inducing=32, seed=0)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - tb
        tc = time.perf_counter()
        pcap = min(M, 256)
        # fp32: the recurrence residual drifts from the true one within ~16 preconditioned steps, so
        # use the reference's residual refresh (max_steps_cycle, conjugate_gradient.py:71-84) every 4
        pcycle = pcap + 1 if esize == 8 else 4
        sol, (steps, err) = conjugate_gradient(op, rhs_rows, None, 1e-6, pre, max_iterations=pcap,
                                               max_steps_cycle=pcycle, check_every=8)
        torch.cuda.synchronize()
        t_solve = time.perf_counter() - tc
        rres = rhs_rows - op.rmatmul(sol)
        pcg = {"preconditioner": "SubsampledNormalPreconditioner(rows_per_inducing=32)",
               "sample_rows": pre.sample_rows, "build_seconds": t_build, "build_seconds_first_call": t_build_cold,
               "error_threshold": 1e-6,
               "iteration_cap": pcap, "max_steps_cycle": pcycle, "iterations": int(steps), "converged": bool(int(steps) < pcap),
               "solve_seconds": t_solve,
               "half_rz_final": float(err.max().item()),
               "true_half_residual_sq": 0.5 * float((rres * rres).sum().item())}
    except Exception as e:
        if world > 1:
            raise  # the leg contains collectives: a rank must not drop out of them silently
        pcg = {"error": repr(e)}

    # ---- CDGP leg at the same size (informational; SURVEY §8e: no per-iteration collective):
    # assignment + cluster statistics over the local rows, one [2,M] all-reduce, then the M x M
    # system (Kmm + Lambda) a = u solved by the device CG with the reference's stopping rule
    cdgp = None
    try:
        from cggp.conjugate_gradient import ConjugateGradient
        from cggp.optimize import nearest_centre_statistics

        def timed(fn):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            return r, 1e3 * (time.perf_counter() - t)

        (_, _, _), t_assign_first = timed(lambda: nearest_centre_statistics(kern, Z, (X, y), "sqeuclidean", allreduce))
        (_, sums, counts), t_assign = timed(lambda: nearest_centre_statistics(kern, Z, (X, y), "sqeuclidean", allreduce))
        counts = torch.where(counts != 0, counts, torch.ones_like(counts))
        u = (sums / counts)[:, None]
        _, t_k_first = timed(lambda: kernels.Kuu(Z, kern, jitter=0.0, diag_add=syn.noise_variance / counts))
        KL, t_k = timed(lambda: kernels.Kuu(Z, kern, jitter=0.0, diag_add=syn.noise_variance / counts))  # warm: the first call pays the M x M allocation
        cgm = ConjugateGradient(1e-6, check_every=25)
        (a, (csteps, cerr)), t_cg = timed(lambda: cgm.solve_with_stats(KL, u))
        res = KL @ a - u
        (_, t_mean) = timed(lambda: ops.knm_matvec(spec, X, Z, a))
        # C3's "64 Hutchinson log-det probe vectors": trace estimator of models.py:308-314 and the
        # log-det-gradient estimator of models.py:37-44, both one 64-RHS CG on (Kmm + Lambda)
        from cggp.models import CGGP
        mdl = CGGP(kern, syn.noise_variance, Z, cgm, num_probes=64, pseudo_u=u, cluster_counts=counts[:, None],
                   num_data=N)
        probes = torch.from_numpy(synthetic.make_probes(M, 64, dtype_name)).to(dev)
        (kl, t_kl) = timed(lambda: mdl.prior_kl(probes=probes))
        (_, t_ldg) = timed(lambda: mdl.logdet_gradient(1.0, probes=probes))
        cdgp_probe = {"prior_kl_64_probes_ms": t_kl, "prior_kl": kl, "logdet_gradient_64_probes_ms": t_ldg,
                      "probe_cg_iterations": int(cgm.last_stats[0])}
        cdgp = {"assign_and_stats_ms": t_assign, "assign_and_stats_first_call_ms": t_assign_first, "kuu_lambda_ms": t_k, "kuu_lambda_first_call_ms": t_k_first,
                "cg_iterations": int(csteps),
                "cg_ms": t_cg, "cg_half_rz_final": float(cerr.max().item()),
                "true_half_residual_sq": 0.5 * float((res * res).sum().item()),
                "predict_mean_all_local_rows_ms": t_mean, **cdgp_probe}
    except Exception as e:  # the headline number must not depend on this leg
        if world > 1:
            raise  # collectives inside: see above
        cdgp = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline
        ns = min(N, args.cpu_sample_rows)
        sec, threads = cpu_baseline.time_cg_iteration(syn.X[:ns], syn.Z, syn.variance, syn.lengthscales,
                                                      syn.noise_variance, kname,
                                                      torch.float64 if esize == 8 else torch.float32)
        cpu = {"value": 1.0 / (sec * (N / ns)), "unit": "CG iters/s", "cores": threads, "kind": "port",
               "sample": f"one CG iteration of the same operator on the first {ns} of {N} rows (dense "
                         f"chunked K build + GEMV, torch-CPU fp{esize * 8}), {sec:.2f} s, scaled by N/sample"}

    if rank == 0:
        out = {
            "metric": "CG iters/sec (matrix-free SGPR-CG, N=2^20 M=4096 D=8 fp64)"
                      if args.config == "C3" and not args.rows and (args.scaling == "strong" or world == 1)
                      else f"CG iters/sec ({args.config}, N={N} total)",
            "value": args.steps / elapsed,
            "unit": "CG iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64" if esize == 8 else "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: CG on S=s2(Kmm+jI)+KmnKnm, K_nm matrix-free, {kname} kernel",
                       "N": N, "D": D, "M": M, "rhs": 1, "rows_per_gpu": n_local,
                       "parallelism": f"rows of X sharded over {world} GPU(s), one all-reduce of [1,M]+1 per step",
                       "collective": ("none (one rank)" if allreduce is None else
                                      "libmgp ncclAllReduce on the solve's stream (mgp_operator.comm)"
                                      if getattr(allreduce, "comm", None) is not None else
                                      "callback hook -> torch.distributed (" + args.backend + ")")},
            "roofline": {
                "bound": "valu",
                "bound_note": "vector-ALU issue roofline: the kernel issues no MFMA (DESIGN.md 4.1: on MI355X the fp64 "
                              "matrix and vector peaks are the same 78.6 TFLOP/s on the same ALUs, measured and "
                              "rejected) and is not HBM-bound (SURVEY 8d) -- the hbm figures BASELINE.json asks for "
                              "are in the nested object",
                "kernel": (f"sweep_fast_kernel<{D},{kname},RC=1,RPT=4,512 threads,2^13-entry table>"
                           if esize == 8 and D <= 8 else
                           f"sweep_fast_kernel<{D},{kname},RC=1>" if esize == 8 else
                           f"sweep_kernel<{'double' if esize == 8 else 'float'},{D},{kname},1>") +
                          " (K_nm.p and K_mn.u are the same symbol)",
                "achieved": ach_tflops, "peak": vector_peak, "unit": "TFLOP/s",
                "frac": ach_tflops / vector_peak,
                "flop_model": "algorithmic flops of the fused SE product, DESIGN.md 4.1 (2D + 11 + 2R per pair)"
                              if flop_model_exact else "the SE count applied to another kernel: approximate",
                "flop_per_pair": ALG_FLOPS_PER_PAIR(D, R), "pairs_per_launch": pairs_launch,
                "valu_instr_per_pair": {"float": n_fl, ("int32" if esize == 8 else "transcendental"): n_int},
                "valu_issue_frac_at_2.4GHz": issue_s / (sweep_ms * 1e-3),
                "survey_flop_per_pair": SURVEY_FLOPS_PER_PAIR(D, R),
                "frac_at_survey_flop_count": pairs_launch * SURVEY_FLOPS_PER_PAIR(D, R) / (sweep_ms * 1e-3) / 1e12
                                             / vector_peak,
                "avg_launch_ms": sweep_ms, "launches_timed": int(launches.value), "launch_percentiles": sweep_pct,
                "gpair_evals_per_s": pairs_launch / (sweep_ms * 1e-3) / 1e9,
                "traffic": traffic,
                "hbm": {"bound": "hbm", "achieved": ach_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": ach_gbps / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": bytes_launch,
                        "equiv_dense_gemv_GBps_derived": equiv_gemv_gbps},
            },
            "cpu_baseline": cpu,
            # "converging to residual <= 1e-6" (north star): the reference recurrence (identity preconditioner)
            # and the preconditioned one, each with iterations and wall time to the reference's stopping rule
            "iterations_to_threshold": {
                "reference_recurrence": (conv["iterations"] if conv and conv["converged"] else None),
                "preconditioned": (pcg.get("iterations") if pcg and pcg.get("converged") else None)},
            "time_to_solution_s": {
                "reference_recurrence": (conv["seconds"] if conv and conv["converged"] else None),
                "preconditioned": ((pcg["build_seconds"] + pcg["solve_seconds"])
                                   if pcg and pcg.get("converged") else None),
                "note": "None = the iteration cap (the reference's default n = M) was hit before 0.5||r||^2 <= 1e-6"},
            "convergence": conv,
            "convergence_preconditioned": pcg,
            "cdgp_same_size": cdgp,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
