set -e
mkdir -p gpurun_out
show() { python -c "
import json,sys; d=json.load(open('$1')); r=d['roofline']; c=r['sustained_clock'] or {}; print('$2', 'it/s %.1f step_ms %.4f sweep_ms %.4f clock %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], c.get('mean_mhz')))"; }
for rep in 1 2; do
for v in "MGP_SWEEP_FAST=2" "MGP_SWEEP_FAST=1" "MGP_SWEEP_FAST=1 MGP_SWEEP_RPT=2" "MGP_SWEEP_FAST=1 MGP_SWEEP_RPT=3" "MGP_SWEEP_FAST=2 MGP_PF_TRIPS=8"; do
  env $v python bench.py --emulate-world 8 --steps 100 > gpurun_out/ab.json 2>/dev/null; show gpurun_out/ab.json "[$v]"
done; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rank8 -o rank8 -- python3 $R/bench.py --emulate-world 8 --steps 200 > $R/gpurun_out/r03_rank_share_w8_under_rocprof.json 2>/dev/null
cd $R
python3 tools/show_stats.py gpurun_out/prof_rank8/rank8_kernel_stats.csv > gpurun_out/r03_rank_share_w8_kernel_stats_summary.txt; cat gpurun_out/r03_rank_share_w8_kernel_stats_summary.txt | cut -c1-150
