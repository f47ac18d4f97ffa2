#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q > gpurun_out/r04_call17_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04_call17_pytest.log
timeout -k 10 600 python tools/ab_dense_cols.py 2048x2 2048x3 2048x5 2048x8 1024x5 1500x5 > gpurun_out/r04_ab_dense_cols5.txt 2>&1; echo "ab rc=$?"; grep round gpurun_out/r04_ab_dense_cols5.txt | sed 's/dense CG //; s/ per iteration (300 steps)//'
