#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q > gpurun_out/r04_call30_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04_call30_pytest.log
export AB_VARIANTS='[["columns of a chunk at different owners",{}],["at one owner",{"MGP_D1_OWNER_SPREAD":"0"}]]'
timeout -k 10 900 python tools/ab_dense_cols.py 2048x2 2048x3 2048x5 2048x8 1024x5 1536x4 > gpurun_out/r04_ab_dense_cols_spread_full.txt 2>&1; sed 's/dense CG //; s/ per iteration (300 steps)//; s/; checksum.*//' gpurun_out/r04_ab_dense_cols_spread_full.txt
timeout -k 10 300 python tools/run_training.py 2>&1 | tail -2
timeout -k 10 300 python tools/stress_dense1.py 200 21 2>&1 | tail -1
