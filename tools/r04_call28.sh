#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q > gpurun_out/r04_call28_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04_call28_pytest.log
export AB_VARIANTS='[["columns of a chunk at different owners",{}],["at one owner",{"MGP_D1_OWNER_SPREAD":"0"}]]'
timeout -k 10 900 python tools/ab_dense_cols.py 4096x2 4096x3 4096x4 4096x5 4096x6 3000x5 2112x5 > gpurun_out/r04_ab_dense_cols_spread.txt 2>&1; sed 's/dense CG //; s/ per iteration (300 steps)//; s/; checksum.*//' gpurun_out/r04_ab_dense_cols_spread.txt
MGP_D1_OWNER_SPREAD=0 timeout -k 10 600 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q -k "super_block or missing_workgroup" 2>&1 | tail -2
