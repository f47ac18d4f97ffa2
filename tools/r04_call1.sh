#!/bin/bash
# round 4, first GPU call: N > 1 hardening
mkdir -p gpurun_out && rm -f gpurun_out/rehearse_summary.txt
timeout -k 10 900 python -m pytest tests/test_distributed.py tests/test_gpu_bench_contract.py tests/test_gpu_rccl.py -m gpu -x -q > gpurun_out/r04_call1_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/rehearse_summary.txt
tools/rehearse_world.sh 3 && tools/rehearse_world.sh 5 && tools/rehearse_world.sh 6
echo "done" | tee -a gpurun_out/rehearse_summary.txt
