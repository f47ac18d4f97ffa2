#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_covertree.py -m gpu -x -q -s > gpurun_out/r04_call8_ct.log 2>&1
echo "covertree pytest rc=$?"; tail -4 gpurun_out/r04_call8_ct.log; grep "cover tree," gpurun_out/r04_call8_ct.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r04_call8_sweeps.log 2>&1
echo "sweep pytest rc=$?"; tail -3 gpurun_out/r04_call8_sweeps.log
for rep in 1 2; do
  for v in 0 1; do
    MGP_SWEEP_TWO_SIZES=$v timeout -k 10 300 python bench.py --no-extra-legs --steps 20 --warmup 3 > gpurun_out/r04_two_sizes${v}_$rep.json 2> gpurun_out/r04_two_sizes${v}_$rep.err || exit 1
    MGP_SWEEP_TWO_SIZES=$v timeout -k 10 300 python bench.py --no-extra-legs --emulate-world 8 --steps 50 --warmup 5 > gpurun_out/r04_two_sizes${v}_w8_$rep.json 2> gpurun_out/r04_two_sizes${v}_w8_$rep.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/r04_two_sizes${v}_$rep.json")); e=json.load(open("gpurun_out/r04_two_sizes${v}_w8_$rep.json"))
print("two_sizes=$v rep=$rep: C3 %.2f it/s %.4f ms/step sweep %.4f ms | w8 share %.4f ms/step sweep %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], e["ms_per_step"], e["roofline"]["avg_launch_ms"]))
PY
  done
done
