set -e
mkdir -p gpurun_out
show() { python -c "
import json,sys; d=json.load(open('$1')); r=d['roofline']; c=r['sustained_clock'] or {}; print('$2', 'it/s %.1f step_ms %.4f sweep_ms %.4f frac %.3f frac_sust %s clock mean %s p10 %s p90 %s n %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('frac_at_sustained_clock'), c.get('mean_mhz'), c.get('p10_mhz'), c.get('p90_mhz'), c.get('workgroups_sampled')))"; }
python bench.py > gpurun_out/r03_bench_c3.json 2> gpurun_out/r03_bench_c3.err; show gpurun_out/r03_bench_c3.json "C3 default"
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_c3.json')); print(json.dumps(d['cdgp_same_size'], indent=1)[:3000])"
for w in 8 4 2; do python bench.py --emulate-world $w --steps 100 > gpurun_out/r03_rank_share_w$w.json 2> gpurun_out/r03_rank_share_w$w.err; show gpurun_out/r03_rank_share_w$w.json "emulate-world $w"; done
python bench.py --config C5 --no-cpu-baseline --convergence-cap 64 > gpurun_out/r03_bench_c5.json 2> gpurun_out/r03_bench_c5.err; show gpurun_out/r03_bench_c5.json "C5"
python bench.py --config C4 --no-cpu-baseline --convergence-cap 64 > gpurun_out/r03_bench_c4.json 2> gpurun_out/r03_bench_c4.err; show gpurun_out/r03_bench_c4.json "C4"
