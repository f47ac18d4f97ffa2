set -e
mkdir -p gpurun_out
show() { python -c "
import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', 'step_ms %.4f sweep_ms %.4f frac %.3f frac_sust %s clock %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('frac_at_sustained_clock'), r['sustained_clock'] and round(r['sustained_clock']['mean_mhz'])))"; }
for rep in 1 2 3; do
  MGP_FUSE_AGREE=1 python bench.py --emulate-world 8 --steps 100 > gpurun_out/ab_f1.json 2>/dev/null; show gpurun_out/ab_f1.json "fuse=1 sampler=on "
  MGP_FUSE_AGREE=0 python bench.py --emulate-world 8 --steps 100 > gpurun_out/ab_f0.json 2>/dev/null; show gpurun_out/ab_f0.json "fuse=0 sampler=on "
  MGP_FUSE_AGREE=1 python bench.py --emulate-world 8 --steps 100 --no-clock-sampler > gpurun_out/ab_f1n.json 2>/dev/null; show gpurun_out/ab_f1n.json "fuse=1 sampler=off"
done
python bench.py --rows 131072 --force-collective --no-extra-legs --steps 100 > gpurun_out/ab_full_slab.json 2>/dev/null; show gpurun_out/ab_full_slab.json "rows=131072, full Kmm slab (round-2 command)"
