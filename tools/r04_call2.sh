#!/bin/bash
# round 4, GPU call 2: generic-D F1/F2 + chunk granularity A/B at C5 (D = 32)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_rccl.py tests/test_distributed.py -m gpu -x -q -k "any_dimension or generic_dimension or rccl or nccl or sharded or bootstrap or vjp or nearest" > gpurun_out/r04_call2_pytest.log 2>&1
echo "pytest rc=$?"
tail -3 gpurun_out/r04_call2_pytest.log
for rep in 1 2; do
  for g in 0 256; do
    MGP_SWEEP_GRAN=$g timeout -k 10 300 python bench.py --config C5 --no-extra-legs --steps 10 --warmup 2 > gpurun_out/r04_c5_gran${g}_$rep.json 2> gpurun_out/r04_c5_gran${g}_$rep.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/r04_c5_gran${g}_$rep.json"))
print("C5 gran=$g rep=$rep", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
PY
  done
done
