set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rank -o rank -- python3 $R/bench.py --rows 131072 --force-collective --no-cpu-baseline --steps 50 --convergence-cap 200 > $R/gpurun_out/rank_share_bench.json 2> $R/gpurun_out/rank_share_bench.err
cd $R
python3 tools/show_stats.py gpurun_out/prof_rank/rank_kernel_stats.csv | head -30
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/rank_share_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])
PY
