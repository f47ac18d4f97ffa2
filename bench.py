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

`roofline` (dominant kernel: the fused sweep, a vector-ALU-issue-bound kernel -- DESIGN.md 4.1).  Every number
can be recomputed from this line and ONE committed file, profiles/valu_issue_model.json (PMC, per config):
    busy  = pairs_per_launch / 64 * active_valu_quadcycles_per_wave_pair          [VALU-busy quad-cycles, all SIMDs]
    frac  = busy * 4 / (1024 SIMDs * 2.4e9 Hz) / avg_launch_s                     [issue-slot fraction, datasheet clock]
    frac_at_sustained_clock = the same with the shader clock MEASURED inside the timed launches
                              (mgp_profile_read_clocks: workgroups stamp s_memtime / s_memrealtime at start and end)
"""

import argparse
import ctypes
import datetime
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 16 fp64 lanes per SIMD and clock, 2.4 GHz peak engine clock
NUM_SIMDS, MAX_CLOCK_HZ = 1024, 2.4e9
FP64_VECTOR_PEAK_TFLOPS = 78.6   # = 1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz (== the fp64 matrix peak, same ALUs)
FP32_VECTOR_PEAK_TFLOPS = 157.3
HBM_PEAK_GBPS = 8000.0           # 8.0 TB/s spec (6.29 TB/s measured copy)
# One VALU wave-instruction of fp64 class holds its SIMD for 4 cycles ("quad-cycle"; SQ_ACTIVE_INST_VALU counts
# them): the issue peak of the chip is 1024 x 2.4e9 / 4 quad-cycles per second.
PEAK_GQUAD_PER_S = NUM_SIMDS * MAX_CLOCK_HZ / 4 / 1e9
# flops the kernels EXECUTE per pair, from their instruction lists (csrc/sweep.hip), by (element size, kernel):
#   fp64 SE (fast kernel): D fma + 3 add (magic) + 2 fma + mul + fma (table) + R fma (accumulate) = 2D + 10 + 2R
#   fp64 Matern-3/2: the same + rsq/Goldschmidt sqrt (8 instr, 14 flop) + polynomial fma + mul = 2D + 27 + 2R (approx.)
#   fp32 SE (LDS-tile kernel, v_exp_f32 counted as ONE flop): D fma + exp + R fma = 2D + 1 + 2R
EXECUTED_FLOPS_PER_PAIR = {(8, "se"): lambda D, R: 2 * D + 10 + 2 * R, (8, "matern32"): lambda D, R: 2 * D + 27 + 2 * R,
                           (4, "se"): lambda D, R: 2 * D + 1 + 2 * R}
SURVEY_FLOPS_PER_PAIR = lambda D, R: 3 * D + 35 + 2 * R  # N M (3D + C_SE + 2R) -- a libm-exp MODEL


def _pct(a):
    a = np.asarray(a, dtype=np.float64)
    return {"median": float(np.median(a)), "p10": float(np.percentile(a, 10)), "p90": float(np.percentile(a, 90))}


class Watchdog:
    """Bounded waits for the N > 1 launch: a rank that is still in `stage` when the deadline passes prints where it
    is and leaves with a non-zero status (os._exit: no exec, no re-launch -- the process has touched the GPU), so a
    bootstrap or collective problem of a node shows up as a failed run, never as a hang."""

    def __init__(self, rank):
        self.rank, self.stage, self.deadline, self.lock = rank, "start", None, threading.Lock()
        self.thread = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
        self.thread.start()

    def arm(self, stage, seconds):
        with self.lock:
            self.stage, self.deadline = stage, (time.monotonic() + seconds if seconds and seconds > 0 else None)

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                stage, deadline = self.stage, self.deadline
            if deadline is not None and time.monotonic() > deadline:
                try:
                    os.write(2, f"[bench.py] rank {self.rank}: timed out in stage '{stage}' -- exiting 124\n".encode())
                finally:
                    os._exit(124)


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
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="N=1 only: what ONE rank of a W-GPU run does per step -- N/W rows, the 1/W row slab of the "
                         "replicated Kmm.p term, the collective on a 1-rank communicator.  Timing only: the operator is "
                         "a rank's PARTIAL, so the legs that need the whole system are skipped")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the convergence / CDGP / CPU legs")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): the config's N is the TOTAL row count, sharded over the ranks -- the "
                         "size BASELINE.json's metric is quoted on; weak: every rank holds the config's N rows")
    ap.add_argument("--bootstrap-timeout", type=float, default=180.0,
                    help="N > 1: seconds the rendezvous + RCCL communicator creation may take before the rank exits 124")
    ap.add_argument("--run-timeout", type=float, default=1200.0,
                    help="N > 1: seconds everything after the bootstrap may take before the rank exits 124 (0 = no limit)")
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
    emu = args.emulate_world if world == 1 else 0
    if emu:
        args.force_collective = True
        args.no_extra_legs = True
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dog = Watchdog(rank) if world > 1 else None
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if dog:
            dog.arm("rendezvous / init_process_group", args.bootstrap_timeout)
        tmo = datetime.timedelta(seconds=max(30.0, args.bootstrap_timeout))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)

    from cggp import _hip, kernels, ops, parallel, synthetic
    from cggp.conjugate_gradient import SgprNormalOperator, conjugate_gradient

    N, D, M, dtype_name, kname = synthetic.CONFIGS[args.config]
    N_config = N
    if args.rows:
        N = args.rows
    if emu:
        N = N // emu
    if args.scaling == "weak":
        N = N * world  # per-GPU work fixed: the job grows with the number of ranks
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
    if dog:
        dog.arm("libmgp RCCL communicator (mgp_comm_init_rank)", args.bootstrap_timeout)
    allreduce = parallel.make_allreduce(force=args.force_collective, timeout_s=0.75 * args.bootstrap_timeout)
    if dog:
        dog.arm("run", args.run_timeout)
    rccl_ranks, collective = allreduce.describe() if allreduce is not None else (1, "none (one rank)")
    if world > 1 and rccl_ranks != world:
        raise SystemExit(f"the collective spans {rccl_ranks} ranks but WORLD_SIZE={world}")
    kmm_rows = parallel.kmm_slab(M, emu, 0) if emu else parallel.kmm_slab(M)
    op = SgprNormalOperator(kern, X, Z, syn.noise_variance, jitter=1e-6, allreduce=allreduce, max_rhs=1,
                            kmm_rows=kmm_rows)
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

    # ---- dominant kernel: the fused sweep (K_nm.p and K_mn.u are the same kernel symbol).  A second pass of the
    # same K steps with HIP events bracketing every launch of it on the solve's stream; 16 workgroups of every
    # bracketed launch stamp the shader-clock and constant counters (mgp_profile_read_clocks)
    hd = _hip.get_handle(dev)
    clock = None
    hd.check(hd.lib.mgp_profile_enable(hd.h, 1))
    run_steps(args.steps)
    launches = ctypes.c_int64(0)
    each = (ctypes.c_double * (2 * args.steps + 8))()
    hd.check(hd.lib.mgp_profile_read_each(hd.h, each, len(each), ctypes.byref(launches)))
    cap = 16 * (2 * args.steps + 8)
    mhz = (ctypes.c_double * cap)()
    nclk = ctypes.c_int64(0)
    if hd.lib.mgp_profile_read_clocks(hd.h, mhz, cap, ctypes.byref(nclk)) == 0 and nclk.value > 0:
        c = np.array(mhz[:min(nclk.value, cap)], dtype=np.float64)
        clock = {"mean_mhz": float(c.mean()), **{k + "_mhz": v for k, v in _pct(c).items()},
                 "min_mhz": float(c.min()), "max_mhz": float(c.max()), "workgroups_sampled": int(nclk.value),
                 "method": "up to 16 workgroups of every timed sweep launch stamp s_memrealtime (100 MHz) and s_memtime "
                           "(shader clock) at their start and at the end of their loop; clock = ratio of the differences "
                           "(mgp_profile_read_clocks)"}
    hd.check(hd.lib.mgp_profile_enable(hd.h, 0))
    durs = np.array(each[:min(launches.value, len(each))], dtype=np.float64)
    sweep_ms = float(durs.mean()) if durs.size else float("nan")
    sweep_pct = {k + "_ms": v for k, v in _pct(durs).items()} if durs.size else None
    R = 1
    pairs_launch = float(n_local) * M
    bytes_launch = float(esize) * (n_local * D + M * D + M * R + n_local * R)
    ach_gbps = bytes_launch / (sweep_ms * 1e-3) / 1e9
    equiv_gemv_gbps = float(esize) * n_local * M / (sweep_ms * 1e-3) / 1e9
    vector_peak = FP64_VECTOR_PEAK_TFLOPS if esize == 8 else FP32_VECTOR_PEAK_TFLOPS
    flops_pair = EXECUTED_FLOPS_PER_PAIR.get((esize, kname), EXECUTED_FLOPS_PER_PAIR[(esize, "se")])(D, R)
    tflops_exec = pairs_launch * flops_pair / (sweep_ms * 1e-3) / 1e12

    # PMC model of this config's kernel instantiation (committed; tools/make_valu_model.py builds it from the
    # rocprofv3 --pmc passes of tools/pmc_sweep.sh)
    model, model_src, model_stale = None, os.path.join("profiles", "valu_issue_model.json"), None
    try:
        model_file = json.load(open(os.path.join(ROOT, model_src)))
        entries = model_file["configs"]
        # the counters belong to ONE version of the kernel source: a later edit makes `frac` stale
        import hashlib
        model_stale = []
        for path, sha in (model_file.get("measured_sources") or {}).items():
            data = open(os.path.join(ROOT, path), "rb").read()
            if hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest() != sha:
                model_stale.append(path)
        model = entries.get(args.config)
        if model is None:  # another config that runs the same kernel instantiation (C2, C3r: the C3 kernel)
            same = [e for e in entries.values()
                    if (e.get("D"), e.get("dtype"), e.get("kernel_kind")) == (D, dtype_name, kname)]
            model = same[0] if same else None
    except Exception:
        model = None
    roof = {"bound": "valu",
            "bound_note": "vector-ALU issue roofline: the kernel issues no MFMA (on MI355X the fp64 matrix and vector "
                          "peaks are the same ALUs: measured and rejected, DESIGN.md 4.1) and is not HBM-bound (SURVEY 8d); "
                          "the hbm figures BASELINE.json asks for are in the nested object",
            "kernel": (f"sweep_fast_kernel<{D},{kname},RC=1,RPT=4,512 threads,2^13-entry table>" if esize == 8 and D <= 8
                       else f"sweep_fast_kernel<{D},{kname},RC=1>" if esize == 8
                       else f"sweep_kernel<float,{D},{kname},1>") + " (K_nm.p and K_mn.u are the same symbol)",
            "unit": "G VALU quad-cycles/s", "peak": PEAK_GQUAD_PER_S,
            "peak_note": "1024 SIMDs x 2.4e9 Hz / 4: one fp64-class VALU wave-instruction holds its SIMD for 4 cycles "
                         "(SQ_ACTIVE_INST_VALU counts these quad-cycles; fp32 transcendentals count 2)"}
    if model:
        q = float(model["active_valu_quadcycles_per_wave_pair"])
        busy = pairs_launch / 64.0 * q
        roof["achieved"] = busy / (sweep_ms * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / PEAK_GQUAD_PER_S
        roof["model"] = {"file": model_src, "config": args.config,
                         "active_valu_quadcycles_per_wave_pair": q,
                         "valu_instructions_per_pair": model.get("valu_instructions_per_pair"),
                         "pmc_source": model.get("source"), "kernel": model.get("kernel"),
                         "stale": bool(model_stale),
                         "stale_note": ("kernel sources edited since the PMC passes: " + ", ".join(model_stale)
                                        if model_stale else "the PMC passes ran on the kernel sources of this tree "
                                        "(git blob hashes in the model file)" if model_stale is not None else None),
                         "recompute": "frac = pairs_per_launch / 64 * active_valu_quadcycles_per_wave_pair * 4 / "
                                      "(1024 * 2.4e9) / (avg_launch_ms * 1e-3)"}
        if clock:
            roof["frac_at_sustained_clock"] = busy * 4.0 / (NUM_SIMDS * clock["mean_mhz"] * 1e6) / (sweep_ms * 1e-3)
    else:
        roof["achieved"] = None
        roof["frac"] = None
        roof["model"] = {"error": f"no entry for {args.config} in {model_src}"}
    traffic = None
    if model and model.get("hbm_bytes_per_launch") and not args.rows and not emu and world == 1:
        traffic = float(model["hbm_bytes_per_launch"])
    roof.update({
        "sustained_clock": clock,
        "avg_launch_ms": sweep_ms, "launches_timed": int(launches.value), "launch_percentiles": sweep_pct,
        "pairs_per_launch": pairs_launch, "gpair_evals_per_s": pairs_launch / (sweep_ms * 1e-3) / 1e9,
        "traffic": traffic,
        "traffic_note": "HBM bytes per launch from PMC (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 correction, separate --pmc "
                        "passes), mean of the K_nm and K_mn launches; null for row counts the PMC passes did not run",
        "flops": {"note": "secondary: flops the kernel EXECUTES per pair (instruction list, csrc/sweep.hip) over the "
                          "vector peak of its dtype; <= the issue fraction because adds/muls/integer ops fill a slot with "
                          "one flop or none", "executed_flop_per_pair": flops_pair, "tflops": tflops_exec,
                  "peak_tflops": vector_peak, "frac": tflops_exec / vector_peak,
                  "survey_model_flop_per_pair": SURVEY_FLOPS_PER_PAIR(D, R),
                  "frac_at_survey_flop_count": pairs_launch * SURVEY_FLOPS_PER_PAIR(D, R) / (sweep_ms * 1e-3) / 1e12
                                               / vector_peak,
                  "survey_note": "SURVEY 8d prices a 20-instruction libm exp the kernel does not execute: a model, "
                                 "not a bound (it exceeds 1)"},
        "hbm": {"bound": "hbm", "achieved": ach_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ach_gbps / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": bytes_launch,
                "equiv_dense_gemv_GBps_derived": equiv_gemv_gbps}})

    extra = not args.no_extra_legs
    # ---- bounded convergence report (informational): real stopping rule, thr = 1e-6
    conv = None
    if extra and (rank == 0 or world > 1):
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
    # DESIGN.md 4.3b): P = s2 Kmm + (N/n_s) Ks^T Ks from n_s = 32 M sampled rows, z = r @ P^-1
    pcg = None
    if extra:
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
                   "iteration_cap": pcap, "max_steps_cycle": pcycle, "iterations": int(steps),
                   "converged": bool(int(steps) < pcap), "solve_seconds": t_solve,
                   "half_rz_final": float(err.max().item()),
                   "true_half_residual_sq": 0.5 * float((rres * rres).sum().item())}
        except Exception as e:
            if world > 1:
                raise  # the leg contains collectives: a rank must not drop out of them silently
            pcg = {"error": repr(e)}

    # ---- CDGP leg at the same size (informational; SURVEY §8e: no per-iteration collective):
    # assignment + cluster statistics over the local rows, one [2,M] all-reduce, then the M x M
    # system (Kmm + Lambda) a = u solved by the device CG with the reference's stopping rule -- the reference's
    # literal hot loop (conjugate_gradient.py:65) -- and the predictive mean AND variance of every local row
    cdgp = None
    if extra:
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
            cgm.solve_with_stats(KL, u)  # warm (tile table, arenas)
            (a, (csteps, cerr)), t_cg = timed(lambda: cgm.solve_with_stats(KL, u))
            res = KL @ a - u
            (_, t_mean) = timed(lambda: ops.knm_matvec(spec, X, Z, a))
            # C3's "64 Hutchinson log-det probe vectors": trace estimator of models.py:308-314 and the
            # log-det-gradient estimator of models.py:37-44, both one 64-RHS CG on (Kmm + Lambda)
            from cggp.models import CGGP
            mdl = CGGP(kern, syn.noise_variance, Z, cgm, num_probes=64, pseudo_u=u, cluster_counts=counts[:, None],
                       num_data=N)
            probes = torch.from_numpy(synthetic.make_probes(M, 64, dtype_name)).to(dev)
            (kl, t_kl_first) = timed(lambda: mdl.prior_kl(probes=probes))  # pays two fresh M x M allocations
            (kl, t_kl) = timed(lambda: mdl.prior_kl(probes=probes))
            (_, t_ldg) = timed(lambda: mdl.logdet_gradient(1.0, probes=probes))
            it_us = 1e3 * t_cg / max(int(csteps), 1)
            # C3's 64-probe solve on its own (the CG inside prior_kl / logdet_gradient): per iteration, against both of
            # its rooflines -- 2 * 64 * M^2 flops on the fp64 matrix cores, M^2 elements of A streamed once
            cgm.solve_with_stats(KL, probes)
            ((_, (psteps, _)), t_pcg) = timed(lambda: cgm.solve_with_stats(KL, probes))
            p_us = 1e3 * t_pcg / max(int(psteps), 1)
            p_flops, p_bytes = 2.0 * 64 * M * M, float(esize) * M * M
            # ... and with the reference's DEFAULT probe count (CGGP(num_probes=5), models.py:286): what every prior_kl /
            # eval_logdet of a training step runs
            probes5 = probes[:, :5].contiguous()
            cgm.solve_with_stats(KL, probes5)
            ((_, (p5steps, _)), t_p5) = timed(lambda: cgm.solve_with_stats(KL, probes5))
            p5_us = 1e3 * t_p5 / max(int(p5steps), 1)
            # the same one-RHS solve at C2's size (M = 2048)
            M2 = min(M, 2048)
            counts2 = counts[:M2].clone()
            KL2 = kernels.Kuu(Z[:M2].contiguous(), kern, jitter=0.0, diag_add=syn.noise_variance / counts2)
            u2 = u[:M2].contiguous()
            cgm.solve_with_stats(KL2, u2)
            ((_, (c2steps, _)), t_cg2) = timed(lambda: cgm.solve_with_stats(KL2, u2))
            it2_us = 1e3 * t_cg2 / max(int(c2steps), 1)
            probes5_2 = probes5[:M2].contiguous()
            cgm.solve_with_stats(KL2, probes5_2)
            ((_, (p52steps, _)), t_p52) = timed(lambda: cgm.solve_with_stats(KL2, probes5_2))
            p52_us = 1e3 * t_p52 / max(int(p52steps), 1)
            tri2_bytes = esize * (M2 * (M2 + 64) / 2.0)
            d1_form = {"3": "register-resident: the whole solve in one launch, A held on the chip (csrc/cg_dense1.hip): the "
                            "upper triangle in 3 x 3 super-blocks of tiles for 2048 < n <= 4096, the full matrix for "
                            "n <= 2048", "4": "register-resident, super-blocks of the triangle at every n <= 4096",
                       "1": "two launches per iteration"}.get(
                os.environ.get("MGP_CG_DENSE1", "3"), "MGP_CG_DENSE1=" + os.environ.get("MGP_CG_DENSE1", ""))
            tri_bytes = esize * (M * (M + 64) / 2.0)  # the upper triangle's 64 x 64 tiles: what an iteration streams
            cdgp = {"assign_and_stats_ms": t_assign, "assign_and_stats_first_call_ms": t_assign_first,
                    "kuu_lambda_ms": t_k, "kuu_lambda_first_call_ms": t_k_first,
                    "cg_iterations": int(csteps), "cg_ms": t_cg, "cg_us_per_iteration": it_us,
                    "cg_form": d1_form,
                    "cg_hbm": {"bytes_per_iteration_if_streamed": tri_bytes,
                               "equivalent_GBps": tri_bytes / (it_us * 1e-6) / 1e9,
                               "equivalent_frac_of_8TBps": tri_bytes / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                               "note": "dense one-RHS CG (csrc/cg_dense1.hip).  bytes = the upper-triangle tiles of "
                                       "Kmm+Lambda, what the two-launch form streams from HBM in every iteration; the "
                                       "register-resident form reads them ONCE per solve, so for it this is an equivalent "
                                       "rate (what a streaming implementation would have to sustain), not traffic: its "
                                       "iteration is bound by two hand-offs between resident workgroups, not by HBM.  "
                                       "Wall time of the whole solve / steps, start-up included"},
                    "cg_c2_size": {"M": M2, "cg_iterations": int(c2steps), "cg_ms": t_cg2, "cg_us_per_iteration": it2_us,
                                   "equivalent_GBps": tri2_bytes / (it2_us * 1e-6) / 1e9,
                                   "probe5_iterations": int(p52steps), "probe5_ms": t_p52,
                                   "probe5_us_per_iteration": p52_us},
                    "probe5_cg": {"columns": 5, "iterations": int(p5steps), "ms": t_p5, "us_per_iteration": p5_us,
                                  "note": "the reference's default num_probes = 5 (models.py:286) on Kmm+Lambda: "
                                          "register-resident form with 5 columns (round 3: skinny MFMA product + fused "
                                          "update, 35 us per iteration)"},
                    "probe_cg": {"columns": 64, "iterations": int(psteps), "ms": t_pcg, "us_per_iteration": p_us,
                                 "tflops": p_flops / (p_us * 1e-6) / 1e12,
                                 "frac_of_fp64_mfma_peak": p_flops / (p_us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                 "GBps_on_A_read_once": p_bytes / (p_us * 1e-6) / 1e9,
                                 "frac_of_8TBps": p_bytes / (p_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                 "note": "64-column CG on Kmm+Lambda (skinny MFMA product + fused update): both rooflines "
                                         "coincide near 27 us per iteration at M = 4096 (2.1 GFLOP at 78.6 TFLOP/s; "
                                         "134 MB at 5 TB/s)"},
                    "cg_half_rz_final": float(cerr.max().item()),
                    "true_half_residual_sq": 0.5 * float((res * res).sum().item()),
                    "predict_mean_all_local_rows_ms": t_mean,
                    "prior_kl_64_probes_ms": t_kl, "prior_kl_64_probes_first_call_ms": t_kl_first, "prior_kl": kl,
                    "logdet_gradient_64_probes_ms": t_ldg,
                    "probe_cg_iterations": int(psteps)}
            # predictive VARIANCE of every local row (models.py:340, SURVEY row M3 "dominant cost today"), both ways
            B = 4096
            mdl.num_probes = None
            # (a) build-side option: (Kmm+Lambda) Y = I once (M-column CG to thr^2), then one [B,M].[M,M] GEMM per batch
            (mv, t_shared) = timed(lambda: mdl.predict_f_batched(X, B, shared_inverse=True))
            inv_steps = int(mdl.inverse_stats[0])
            Y = torch.empty((M, M), dtype=X.dtype, device=dev).normal_()
            Knm = torch.empty((B, M), dtype=X.dtype, device=dev).normal_()
            ops.symm_matmul(Y, Knm)
            nb_gemm = 16
            (_, t_gemm) = timed(lambda: [ops.symm_matmul(Y, Knm) for _ in range(nb_gemm)])
            gemm_tflops = 2.0 * B * M * M * nb_gemm / (t_gemm * 1e-3) / 1e12
            # (b) the reference's form: a B-column CG per batch -- on a bounded sample of batches
            ns = min(n_local, 4 * B)
            ((mu_b, var_b), t_batch) = timed(lambda: mdl.predict_f_batched(X[:ns], B, shared_inverse=False))
            batch_steps = int(cgm.last_stats[0])
            dv = float((mv[1][:ns] - var_b).abs().max().item())
            cdgp["predict_mean_and_variance_all_local_rows"] = {
                "rows": n_local, "batch": B,
                "shared_inverse": {"seconds": t_shared * 1e-3, "inverse_cg_iterations": inv_steps,
                                   "gemm_flops": 2.0 * n_local * M * M,
                                   "gemm_tflops_measured_on_this_shape": gemm_tflops,
                                   "gemm_frac_of_fp64_mfma_peak": gemm_tflops / FP64_VECTOR_PEAK_TFLOPS,
                                   "note": "one M-column CG (threshold^2) + per batch k_dense, [B,M].[M,M] on the matrix "
                                           "cores, column dots; GEMM rate from 16 calls of the same shape"},
                "per_batch_cg": {"sample_rows": ns, "seconds_sample": t_batch * 1e-3,
                                 "seconds_projected_all_rows": t_batch * 1e-3 * n_local / ns,
                                 "cg_iterations_last_batch": batch_steps,
                                 "flops_per_iteration": 2.0 * B * M * M,
                                 "note": "the reference's W = CG(Kmm+Lambda, Kmn) per batch (models.py:340), B = 4096 "
                                         "columns at threshold 1e-6; timed on the sample, projected linearly"},
                "max_abs_variance_difference_on_sample": dv}
        except Exception as e:  # the headline number must not depend on this leg
            if world > 1:
                raise  # collectives inside: see above
            cdgp = {"error": repr(e)}

    cpu = None
    if extra and rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline
        ns = min(N, args.cpu_sample_rows)
        sec, threads = cpu_baseline.time_cg_iteration(syn.X[:ns], syn.Z, syn.variance, syn.lengthscales,
                                                      syn.noise_variance, kname,
                                                      torch.float64 if esize == 8 else torch.float32)
        cpu = {"value": 1.0 / (sec * (N / ns)), "unit": "CG iters/s", "cores": threads, "kind": "port",
               "sample": f"one CG iteration of the same operator on the first {ns} of {N} rows (dense "
                         f"chunked K build + GEMV, torch-CPU fp{esize * 8}), {sec:.2f} s, scaled by N/sample"}

    if rank == 0:
        headline = args.config == "C3" and not args.rows and not emu and (args.scaling == "strong" or world == 1)
        out = {
            "metric": "CG iters/sec (matrix-free SGPR-CG, N=2^20 M=4096 D=8 fp64)" if headline
                      else (f"CG iters/sec ({args.config}: ONE rank's share of a {emu}-GPU run, {N} of {N_config} rows)"
                            if emu else f"CG iters/sec ({args.config}, N={N} total)"),
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
                       "kmm_rows_of_this_rank": [int(kmm_rows[0]), int(kmm_rows[1])],
                       "emulated_world": emu or None,
                       "parallelism": f"rows of X sharded over {emu or world} GPU(s), one all-reduce of [1,M]+1 per step",
                       "collective": collective,
                       "rccl_ranks": rccl_ranks,
                       "rccl_ranks_note": "mgp_comm_size of libmgp's own communicator when the collective is native, "
                                          "torch.distributed's group size when it goes through the callback hook",
                       "native_comm_error": getattr(allreduce, "native_error", None)},
            "roofline": roof,
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
