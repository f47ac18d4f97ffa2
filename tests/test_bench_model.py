"""bench.py's `roofline` is recomputable from its own line and ONE committed file, profiles/valu_issue_model.json
(PMC, per BASELINE config; tools/make_valu_model.py).  CPU-side checks of that file and of the constants bench.py
divides by -- the GPU-side contract is tests/test_gpu_bench_contract.py."""

import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # defines main(); does not run it
    return mod


def test_model_file_covers_every_single_gpu_bench_config():
    m = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_model.json")))
    for cfg in ("C3", "C4", "C5"):
        e = m["configs"][cfg]
        q, ins = e["active_valu_quadcycles_per_wave_pair"], e["valu_instructions_per_pair"]
        assert q > 0 and ins > 0 and q >= ins * 0.999  # an instruction holds its SIMD for at least one quad-cycle
        assert os.path.exists(os.path.join(ROOT, e["source"])), e["source"]
        raw = json.load(open(os.path.join(ROOT, e["source"])))
        # the entry is the mean of the raw per-launch counters it cites
        qs = [v["SQ_ACTIVE_INST_VALU"] / (e["pairs_per_launch"] / 64.0) for k, v in raw["launches"].items()
              if "[knm]" in k or "[kmn]" in k]
        assert len(qs) == 2 and abs(sum(qs) / 2 - q) < 1e-9 * q
    # fp64 classes are one quad-cycle per instruction; fp32 transcendentals hold the SIMD longer
    c3, c4 = m["configs"]["C3"], m["configs"]["C4"]
    assert abs(c3["active_valu_quadcycles_per_wave_pair"] / c3["valu_instructions_per_pair"] - 1.0) < 1e-3
    assert c4["active_valu_quadcycles_per_wave_pair"] > 1.2 * c4["valu_instructions_per_pair"]


def test_model_belongs_to_the_kernel_sources_of_this_tree():
    """The PMC counters describe ONE version of the sweep kernels: the model file names the git blob hashes of the
    sources they were taken with, bench.py reports a mismatch as `roofline.model.stale`, and this test fails on it
     -- after editing csrc/sweep.hip or csrc/mgp_math.h re-run tools/pmc_sweep.sh + tools/make_valu_model.py."""
    import hashlib
    m = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_model.json")))
    assert m["measured_sources"], "the model file must name the sources it was measured with"
    for path, sha in m["measured_sources"].items():
        data = open(os.path.join(ROOT, path), "rb").read()
        assert hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest() == sha, f"{path} edited since the PMC passes"


def test_bench_constants_and_recomputation():
    b = _bench()
    assert b.NUM_SIMDS == 1024 and b.MAX_CLOCK_HZ == 2.4e9
    assert abs(b.PEAK_GQUAD_PER_S - 614.4) < 1e-9
    # 1024 SIMDs x 16 fp64 lanes x 2 flop x 2.4 GHz
    assert abs(b.FP64_VECTOR_PEAK_TFLOPS - 1024 * 16 * 2 * 2.4e9 / 1e12) < 0.1
    assert b.EXECUTED_FLOPS_PER_PAIR[(8, "se")](8, 1) == 28 and b.SURVEY_FLOPS_PER_PAIR(8, 1) == 61
    # the committed line of this round recomputes from the model file
    line_path = os.path.join(ROOT, "profiles", "r04_final_bench.json")
    if os.path.exists(line_path):
        d = json.load(open(line_path))
        r = d["roofline"]
        m = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_model.json")))["configs"]["C3"]
        q = r["model"]["active_valu_quadcycles_per_wave_pair"]
        frac = r["pairs_per_launch"] / 64 * q * 4 / (1024 * 2.4e9) / (r["avg_launch_ms"] * 1e-3)
        assert abs(frac - r["frac"]) < 1e-9 and r["frac"] <= 1.0
        assert abs(q - m["active_valu_quadcycles_per_wave_pair"]) < 1e-3 * q
        f2 = r["pairs_per_launch"] / 64 * q * 4 / (1024 * r["sustained_clock"]["mean_mhz"] * 1e6) / (r["avg_launch_ms"] * 1e-3)
        assert abs(f2 - r["frac_at_sustained_clock"]) < 1e-9 and f2 <= 1.0
