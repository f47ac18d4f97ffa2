#!/bin/bash
# A/B: the s2*Kmm.p slab product beside the K_nm sweep (MGP_SGPR_KMM_ASIDE=1, new) against behind both sweeps (0)
mkdir -p gpurun_out; out=gpurun_out/r04_ab_kmm_aside.txt; : > $out
for rnd in 0 1; do for v in 1 0; do
  MGP_SGPR_KMM_ASIDE=$v timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-extra-legs > gpurun_out/_ab.json 2> gpurun_out/_ab.err || { echo "bench failed"; tail -5 gpurun_out/_ab.err; exit 1; }
  python - >> $out <<PY
import json
d=json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("round $rnd aside=$v: %.2f it/s  %.4f ms/step  sweep %.4f ms  step - 2 sweeps = %.1f us" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], 1e3*(d["ms_per_step"]-2*r["avg_launch_ms"])))
PY
done; done
for v in 1 0; do MGP_SGPR_KMM_ASIDE=$v timeout -k 10 300 python bench.py --emulate-world 8 --steps 100 --no-extra-legs > gpurun_out/_ab.json 2> gpurun_out/_ab.err || exit 1
python - >> $out <<PY
import json
d=json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("one rank's share of 8, aside=$v: %.4f ms/step  sweep %.4f ms  step - 2 sweeps = %.1f us" % (d["ms_per_step"], r["avg_launch_ms"], 1e3*(d["ms_per_step"]-2*r["avg_launch_ms"])))
PY
done
cat $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py tests/test_gpu_rccl.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r04_call27_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_call27_pytest.log
