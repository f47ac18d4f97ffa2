"""Builds profiles/valu_issue_model.json -- the ONE file bench.py's `roofline` needs besides its own line -- from the
rocprofv3 --pmc passes of tools/pmc_sweep.sh (one directory per config under gpurun_out/).

    python tools/make_valu_model.py <round tag> C3=gpurun_out/r03_pmc_c3 C4=gpurun_out/r03_pmc_c4 C5=...

Per config it keeps, as means over the K_nm.v and K_mn.u launches (tools/run_sweep.py alternates them):
  active_valu_quadcycles_per_wave_pair = SQ_ACTIVE_INST_VALU / (pairs / 64)   (VALU-busy time, in units of 4 cycles)
  valu_instructions_per_pair           = SQ_INSTS_VALU * 64 / pairs
  hbm_bytes_per_launch                 = (FETCH_SIZE * 2 + WRITE_SIZE) * 1024   (MI355X_MICROARCH.md: gfx950 correction)
and the raw per-launch counter means (copied to profiles/<tag>_pmc_<config>.json by this script).
"""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))
from cggp import synthetic  # noqa: E402

MODEL_SOURCES = ("conjugate-gradient-sparse-gp_amd/csrc/sweep.hip", "conjugate-gradient-sparse-gp_amd/csrc/mgp_math.h")


def git_blob_sha1(path):
    """`git hash-object` of a file: the identifier of the kernel source the counters were taken with."""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


tag = sys.argv[1]
# the instantiation launch_sweep picks for each config (csrc/sweep.hip; rocprofv3 truncates the templated names)
KERNEL_NAMES = {"C3": "sweep_fast_kernel<DP=8, SE, RC=1, RPT=4, NT=512, TBITS=13, DBUF=true>",
                "C4": "sweep_kernel<float, DP=2, SE, RC=1>  (LDS tile, v_exp_f32, v_pk_fma_f32)",
                "C5": "sweep_fast_kernel<DP=32, Matern32, RC=1, RPT=2, NT=256, TBITS=11, DBUF=false>"}
out = {"what": "PMC model of the fused sweep per BASELINE config: VALU-busy quad-cycles per wave-pair (64 pair "
               "evaluations), from rocprofv3 --pmc passes (tools/pmc_sweep.sh; one counter set per pass) over "
               "tools/run_sweep.py.  bench.py: frac = pairs_per_launch / 64 * active_valu_quadcycles_per_wave_pair * 4 "
               "/ (1024 SIMDs * 2.4e9 Hz) / avg_launch_seconds.",
       "round": tag,
       "measured_sources": {p: git_blob_sha1(os.path.join(ROOT, p)) for p in MODEL_SOURCES},
       "measured_sources_note": "git blob hashes of the files that define the sweep kernels' instruction mix when the PMC "
                                "passes ran; bench.py reports a mismatch as roofline.model.stale and "
                                "tests/test_bench_model.py fails on it: re-run tools/pmc_sweep.sh + this script after "
                                "editing them",
       "configs": {}}
for arg in sys.argv[2:]:
    cfg, d = arg.split("=")
    N, D, M, dt, kname = synthetic.CONFIGS[cfg]
    pairs = float(N) * M
    summ = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), d, "sweep",
                                      "--alternate"], capture_output=True, text=True, check=True).stdout)
    # the summary is already filtered to kernels whose name contains "sweep"; the sweep launches are the ones the
    # dispatch order labelled (the partial-reduce kernels carry no such label)
    launches = {k: v for k, v in summ.items() if ("[knm]" in k or "[kmn]" in k) and "SQ_ACTIVE_INST_VALU" in v}
    assert launches, (cfg, list(summ))
    q = [v["SQ_ACTIVE_INST_VALU"] / (pairs / 64.0) for v in launches.values()]
    ins = [v["SQ_INSTS_VALU"] * 64.0 / pairs for v in launches.values()]
    hbm = [(v["FETCH_SIZE"] * 2.0 + v["WRITE_SIZE"]) * 1024.0 for v in launches.values() if "FETCH_SIZE" in v and "WRITE_SIZE" in v]
    util = [v["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0) for v in launches.values()
            if "GRBM_GUI_ACTIVE" in v]
    raw = os.path.join("profiles", f"{tag}_pmc_{cfg.lower()}_sweep.json")
    json.dump({"command": f"tools/pmc_sweep.sh {os.path.basename(d)} {cfg} (rocprofv3 --pmc, one set per pass, over "
                          f"tools/run_sweep.py: 3 x K_nm.v + 3 x K_mn.u, {cfg}: N={N} D={D} M={M} {dt} {kname}, R=1; "
                          "means per dispatch)", "launches": summ}, open(os.path.join(ROOT, raw), "w"), indent=1)
    out["configs"][cfg] = {
        "kernel": KERNEL_NAMES.get(cfg, ""),
        "N": N, "D": D, "M": M, "dtype": dt, "kernel_kind": kname, "pairs_per_launch": pairs,
        "active_valu_quadcycles_per_wave_pair": sum(q) / len(q),
        "active_valu_quadcycles_per_wave_pair_by_launch": dict(zip(sorted(launches), [q[i] for i in sorted(range(len(q)), key=lambda i: sorted(launches).index(list(launches)[i]))])),
        "valu_instructions_per_pair": sum(ins) / len(ins),
        "hbm_bytes_per_launch": (sum(hbm) / len(hbm)) if hbm else None,
        "algorithmic_bytes_per_launch": float(8 if dt == "float64" else 4) * (N * D + M * D + M + N),
        "valu_issue_utilisation_in_the_pmc_run (busy x 4 / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs)": (sum(util) / len(util)) if util else None,
        "source": raw}
json.dump(out, open(os.path.join(ROOT, "profiles", "valu_issue_model.json"), "w"), indent=1)
print(json.dumps({k: {a: v[a] for a in ("active_valu_quadcycles_per_wave_pair", "valu_instructions_per_pair", "hbm_bytes_per_launch")} for k, v in out["configs"].items()}, indent=1))
