#!/bin/bash
# Round-end evidence: the default bench command bare, then the same command under rocprofv3 --kernel-trace --stats.
# Usage (on the GPU box, from the repo root): bash tools/final_evidence.sh <tag>   -> gpurun_out/<tag>_*
set -u
tag=${1:-final}
R=$GRAFT_REPO_ROOT
python3 "$R/bench.py" > "$R/gpurun_out/${tag}_bench.json" 2> "$R/gpurun_out/${tag}_bench.err"
echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${tag}_prof" -- python3 "$R/bench.py" > "$R/gpurun_out/${tag}_bench_under_rocprof.json" 2> "$R/gpurun_out/${tag}_rocprof.err"
echo "rocprof rc=$?"
cp "$R"/gpurun_out/${tag}_prof/*/*kernel_stats.csv "$R/gpurun_out/${tag}_kernel_stats.csv"
cp "$R"/gpurun_out/${tag}_prof/*/*kernel_trace.csv "$R/gpurun_out/${tag}_kernel_trace.csv"
python3 "$R/tools/show_stats.py" "$R/gpurun_out/${tag}_kernel_stats.csv" "$R/gpurun_out/${tag}_kernel_trace.csv" > "$R/gpurun_out/${tag}_kernel_stats_summary.txt"
head -5 "$R/gpurun_out/${tag}_kernel_stats_summary.txt"
