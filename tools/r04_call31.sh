#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["default",{}],["two-launch",{"MGP_CG_DENSE1":"1"}]]'
timeout -k 10 300 python tools/ab_dense1.py 2 2048 4096 > gpurun_out/r04_ab_dense1_check.txt 2>&1; cut -c1-400 gpurun_out/r04_ab_dense1_check.txt | sed 's/, [0-9a-f]\{16\}//g; s/ (no poll \/ poll 25)//g'
python bench.py --no-extra-legs --steps 20 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%.1f it/s sweep %.4f frac %.3f/%.3f clock %s' % (d['value'], r['avg_launch_ms'], r['frac'], r['frac_at_sustained_clock'], r.get('sustained_clock')))"
