#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["default",{}],["two-launch",{"MGP_CG_DENSE1":"1"}]]'
timeout -k 10 300 python tools/ab_dense1.py 1 1024 2048 3000 4096 > gpurun_out/r04_ab_dense1_check2.txt 2>&1; cut -c1-600 gpurun_out/r04_ab_dense1_check2.txt | sed 's/, [0-9a-f]\{16\}//g; s/ (no poll \/ poll 25)//g'
export AB_VARIANTS='[["default",{}]]'
MGP_ONLY=1 timeout -k 10 600 python tools/ab_dense_cols.py 4096x2 4096x3 4096x5 4096x6 4096x7 4096x8 2048x2 2048x5 2048x8 3000x5 > gpurun_out/r04_ab_dense_cols_idle.txt 2>&1; grep "round 0" gpurun_out/r04_ab_dense_cols_idle.txt | sed 's/dense CG //; s/ per iteration (300 steps)//; s/; checksum.*//'
timeout -k 10 1000 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q  > gpurun_out/r04_call32_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r04_call32_pytest.log
