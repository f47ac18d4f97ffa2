#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q > gpurun_out/r04_call11_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04_call11_pytest.log
timeout -k 10 600 python tools/ab_dense_cols.py > gpurun_out/r04_ab_dense_cols.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r04_ab_dense_cols.txt
