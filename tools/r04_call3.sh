#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dense1.py tests/test_gpu_training.py -m gpu -x -q > gpurun_out/r04_call3_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r04_call3_pytest.log
timeout -k 10 900 python tools/ab_dense1.py 2 > gpurun_out/r04_ab_dense1.txt 2>&1
echo "ab rc=$?"; cat gpurun_out/r04_ab_dense1.txt
