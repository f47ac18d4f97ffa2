"""The driver's contract for bench.py: one JSON line on stdout carrying the required keys, the
roofline and cpu_baseline objects; run here at a reduced row count so it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup",
                          "1", "--rows", "65536", "--convergence-cap", "8", "--cpu-sample-rows", "2048"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "CG iters/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6 * 1e3
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma", "valu") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["hbm"]["bound"] == "hbm" and r["launches_timed"] == 6
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert d["convergence_preconditioned"]["converged"] is True


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    """The N > 1 launch of the contract (torch.distributed.run, one rank per process) rehearsed with gloo
    on the one GPU of the box: row sharding, the all-reduce hook, max-over-ranks timing, rank 0 prints."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--rows", "65536",
                          "--convergence-cap", "8", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rows_per_gpu"] == 32768 and d["scaling"] == "strong"
    assert d["cpu_baseline"] is None  # rank 0 at N = 1 only
    assert d["convergence_preconditioned"]["converged"] is True
    c = d["cdgp_same_size"]
    assert c["cg_iterations"] > 0 and c["cg_us_per_iteration"] > 0 and "register-resident" in c["cg_form"]
    # the reference's default five probes (models.py:286) and the 64 Hutchinson probes of C3, per iteration
    assert c["probe5_cg"]["columns"] == 5 and c["probe5_cg"]["us_per_iteration"] > 0
    assert c["cg_c2_size"]["probe5_us_per_iteration"] > 0
    assert c["probe_cg"]["columns"] == 64 and 0 < c["probe_cg"]["frac_of_fp64_mfma_peak"] < 1


@pytest.mark.gpu
def test_bench_rank_share_emulation():
    """`--emulate-world W`: one rank's share of a W-GPU step on the one GPU of the box (N/W rows, the 1/W slab of
    Kmm.p, the exchange on a one-rank libmgp communicator).  Timing only -- the other legs are skipped."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulate-world", "4", "--steps", "4",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert c["emulated_world"] == 4 and c["rows_per_gpu"] == (1 << 20) // 4 and c["kmm_rows_of_this_rank"] == [0, 1024]
    assert "mgp_operator.comm" in c["collective"] and d["n_gpus"] == 1
    assert d["cpu_baseline"] is None and d["convergence"] is None and d["cdgp_same_size"] is None
    r = d["roofline"]
    assert r["launches_timed"] == 8 and 0.3 < r["frac"] <= 1.0 and r["frac"] <= r["frac_at_sustained_clock"] <= 1.0
    clk = r["sustained_clock"]
    assert 1000 < clk["min_mhz"] <= clk["mean_mhz"] <= clk["max_mhz"] <= 2500 and clk["workgroups_sampled"] >= 8
    assert r["traffic"] is None  # the PMC passes ran at the full size only
