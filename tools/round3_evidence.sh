#!/bin/bash
# Round-3 evidence beyond tools/final_evidence.sh: rank shares (one rank of a 2/4/8-GPU run), the C4 / C5 lines, and
# the itemised kernel stats of the 8-rank share.  Usage (GPU box, repo root): bash tools/round3_evidence.sh
set -u
R=$GRAFT_REPO_ROOT
mkdir -p "$R/gpurun_out"
for w in 8 4 2; do
  python3 "$R/bench.py" --emulate-world $w --steps 100 > "$R/gpurun_out/r03_rank_share_w$w.json" 2> "$R/gpurun_out/r03_rank_share_w$w.err"
done
python3 "$R/bench.py" --config C4 --emulate-world 8 --steps 50 > "$R/gpurun_out/r03_rank_share_c4_w8.json" 2>/dev/null
python3 "$R/bench.py" --config C5 --no-cpu-baseline --convergence-cap 64 > "$R/gpurun_out/r03_bench_c5.json" 2>/dev/null
python3 "$R/bench.py" --config C4 --no-cpu-baseline --convergence-cap 64 > "$R/gpurun_out/r03_bench_c4.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r03_rank8_prof" -o rank8 -- python3 "$R/bench.py" --emulate-world 8 --steps 200 > "$R/gpurun_out/r03_rank_share_w8_under_rocprof.json" 2>/dev/null
cd "$R"
python3 tools/show_stats.py "gpurun_out/r03_rank8_prof/rank8_kernel_stats.csv" > gpurun_out/r03_rank_share_w8_kernel_stats_summary.txt
python3 - <<'PY'
import json
for f in ("r03_rank_share_w8", "r03_rank_share_w4", "r03_rank_share_w2", "r03_rank_share_c4_w8", "r03_bench_c5", "r03_bench_c4"):
    d = json.load(open(f"gpurun_out/{f}.json")); r = d["roofline"]; c = r.get("sustained_clock") or {}
    print(f, "it/s %.1f step_ms %.4f sweep_ms %.4f frac %.3f frac_sust %.3f clock %.0f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("frac_at_sustained_clock") or 0, c.get("mean_mhz") or 0))
PY
head -8 gpurun_out/r03_rank_share_w8_kernel_stats_summary.txt
