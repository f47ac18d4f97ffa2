#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_dense1.py tests/test_gpu_training.py tests/test_gpu_models.py -m gpu -x -q > gpurun_out/r04_call23_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04_call23_pytest.log
export AB_VARIANTS='[["default",{}],["first poll 0",{"MGP_D1_FIRST_POLL":"0"}],["first poll 8",{"MGP_D1_FIRST_POLL":"8"}],["first poll 32",{"MGP_D1_FIRST_POLL":"32"}]]'
timeout -k 10 300 python tools/ab_dense1.py 1 2048 4096 > gpurun_out/r04_ab_dense1_firstpoll.txt 2>&1; cut -c1-400 gpurun_out/r04_ab_dense1_firstpoll.txt | sed 's/, [0-9a-f]\{16\}//g; s/ (no poll \/ poll 25)//g'
timeout -k 10 600 python tools/ab_dense_cols.py 4096x5 4096x6 2048x5 > gpurun_out/r04_ab_dense_cols_fp.txt 2>&1; sed 's/dense CG //; s/ per iteration (300 steps)//; s/; checksum.*//' gpurun_out/r04_ab_dense_cols_fp.txt
timeout -k 10 300 python tools/stress_dense1.py 300 11 2>&1 | tail -2
timeout -k 10 300 python tools/run_training.py 2>&1 | tail -3
