"""In-run A/B of the one-RHS dense CG forms (csrc/cg_dense1.hip): one child process per variant (the switches are read
at handle creation), `rounds` alternations, per n: us per iteration with no poll inside the timed region
(check_every = steps) and with the facade's polls every 25 steps; and a digest of the k-step iterate, which must be
the same for every variant (same sums in the same order).
    python tools/ab_dense1.py [rounds] [n ...]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [("two-launch, polls drain (r03)", {"MGP_CG_DENSE1": "1", "MGP_CG_PIPELINE_POLLS": "0"}),
            ("two-launch, one batch in flight", {"MGP_CG_DENSE1": "1", "MGP_CG_PIPELINE_POLLS": "1"}),
            ("one launch, polls drain", {"MGP_CG_DENSE1": "2", "MGP_CG_PIPELINE_POLLS": "0"}),
            ("one launch, one batch in flight", {"MGP_CG_DENSE1": "2", "MGP_CG_PIPELINE_POLLS": "1"})]

if os.environ.get("AB_VARIANTS"):
    VARIANTS = [(n, e) for n, e in json.loads(os.environ["AB_VARIANTS"])]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
    import torch
    from cggp import kernels
    from cggp.conjugate_gradient import conjugate_gradient
    dev = torch.device("cuda:0")

    def timeit(fn, reps, warm=2):
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

    out = {}
    for n in (int(a) for a in sys.argv[2:]):
        g = torch.Generator(device="cpu").manual_seed(n)
        Z = torch.randn(n, 8, generator=g, dtype=torch.float64).to(dev)
        A = kernels.SquaredExponential(1.0, [1.0] * 8).K(Z) + 0.1 * torch.eye(n, dtype=torch.float64, device=dev)
        B = torch.randn(1, n, generator=g, dtype=torch.float64).to(dev)
        k = 400
        t_pure = timeit(lambda: conjugate_gradient(A, B, None, 0.0, max_iterations=k, max_steps_cycle=k + 1, check_every=k), 5)
        t_poll = timeit(lambda: conjugate_gradient(A, B, None, 0.0, max_iterations=k, max_steps_cycle=k + 1, check_every=25), 5)
        sol, (steps, err) = conjugate_gradient(A, B, None, 0.0, max_iterations=37, max_steps_cycle=38, check_every=5)
        # a converging solve through the stopping rule, wall time (host view: includes the tail of gated launches)
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s2, (st2, e2) = conjugate_gradient(A, B, None, 1e-6, max_iterations=n, max_steps_cycle=n + 1, check_every=25)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        out[str(n)] = {"us_per_iteration_no_poll": 1e3 * t_pure / k, "us_per_iteration_poll_25": 1e3 * t_poll / k,
                       "digest_37_steps": hashlib.sha1(sol.cpu().numpy().tobytes()).hexdigest()[:16],
                       "err_37": float(err[0]), "converging_solve_steps": int(st2),
                       "converging_solve_us_per_step_wall": 1e6 * wall / max(int(st2), 1)}
    print(json.dumps(out))
    sys.exit(0)

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ns = sys.argv[2:] or ["2048", "4096", "4001", "1024"]
res = {name: [] for name, _ in VARIANTS}
for r in range(rounds):
    for name, env in VARIANTS:
        p = subprocess.run([sys.executable, __file__, "--child"] + ns, env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=600)
        if p.returncode != 0:
            print(f"[{name}] FAILED rc={p.returncode}\n{p.stderr[-2000:]}", flush=True)
            continue
        d = json.loads(p.stdout.strip().splitlines()[-1])
        res[name].append(d)
        print(f"round {r} [{name}] " + "  ".join(
            f"n={n}: {v['us_per_iteration_no_poll']:.2f} / {v['us_per_iteration_poll_25']:.2f} us (no poll / poll 25), "
            f"solve {v['converging_solve_steps']} steps {v['converging_solve_us_per_step_wall']:.2f} us/step wall, {v['digest_37_steps']}"
            for n, v in d.items()), flush=True)
digests = {n: {d[n]["digest_37_steps"] for runs in res.values() for d in runs if n in d} for n in ns}
print("digests identical across variants:", {n: len(v) == 1 for n, v in digests.items()})
