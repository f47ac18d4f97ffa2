#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q > gpurun_out/r04_call18_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04_call18_pytest.log
export AB_VARIANTS='[["register-resident",{}],["two-launch",{"MGP_CG_DENSE1":"1"}]]'
timeout -k 10 400 python tools/ab_dense1.py 1 1024 2048 3000 4096 > gpurun_out/r04_ab_dense1_b128.txt 2>&1; cut -c1-520 gpurun_out/r04_ab_dense1_b128.txt | sed 's/ (no poll \/ poll 25)//g; s/, [0-9a-f]\{16\}//g'
MGP_ONLY=1 timeout -k 10 600 python tools/ab_dense_cols.py 2048x2 2048x3 2048x5 2048x8 1024x5 > gpurun_out/r04_ab_dense_cols6.txt 2>&1; grep "tile scheme" gpurun_out/r04_ab_dense_cols6.txt | sed 's/dense CG //; s/ per iteration (300 steps)//'
