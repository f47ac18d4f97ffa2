#!/bin/bash
# super-block form of the register-resident dense CG: tests, then A/B against the round-robin triangle form / the routes of before
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q > gpurun_out/r04_call20_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r04_call20_pytest.log
export AB_VARIANTS='[["super-blocks",{}],["round-robin triangle",{"MGP_CG_DENSE1":"5"}]]'
timeout -k 10 300 python tools/ab_dense1.py 1 2112 3000 4096 > gpurun_out/r04_ab_dense1_blk.txt 2>&1; cut -c1-600 gpurun_out/r04_ab_dense1_blk.txt | sed 's/, [0-9a-f]\{16\}//g'
MGP_ONLY=1 timeout -k 10 600 python tools/ab_dense_cols.py 4096x2 4096x3 4096x4 4096x5 4096x6 3000x5 > gpurun_out/r04_ab_dense_cols_blk.txt 2>&1; sed 's/dense CG //; s/ per iteration (300 steps)//' gpurun_out/r04_ab_dense_cols_blk.txt
