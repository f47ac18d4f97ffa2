#!/bin/bash
# round 4, final evidence: default bench bare + under rocprofv3 --kernel-trace --stats; one rank's share of 8; C2-size
set -u
R=$GRAFT_REPO_ROOT
mkdir -p "$R/gpurun_out"
bash "$R/tools/final_evidence.sh" r04_final
cd "$R"
python3 bench.py --emulate-world 8 --steps 100 > gpurun_out/r04_rank_share_w8.json 2> gpurun_out/r04_rank_share_w8.err; echo "w8 rc=$?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r04_final_bench.json")); c=d["cdgp_same_size"]; r=d["roofline"]
print("C3: %.1f it/s %.4f ms/step sweep %.4f frac %.3f / %.3f stale=%s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["frac_at_sustained_clock"], r["model"].get("stale")))
print("cdgp: cg %d it %.3f ms %.2f us/it | c2 %s | probe5 %s | probe %.1f us %.3f" % (c["cg_iterations"], c["cg_ms"], c["cg_us_per_iteration"], c["cg_c2_size"], {k: c["probe5_cg"][k] for k in ("iterations", "ms", "us_per_iteration")}, c["probe_cg"]["us_per_iteration"], c["probe_cg"]["frac_of_fp64_mfma_peak"]))
print("step - 2 sweeps = %.1f us" % (1e3 * (d["ms_per_step"] - 2 * r["avg_launch_ms"])))
e=json.load(open("gpurun_out/r04_rank_share_w8.json")); print("w8: %.4f ms/step sweep %.4f" % (e["ms_per_step"], e["roofline"]["avg_launch_ms"]))
PY
