#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q -k "super_block or several_columns or every_form or missing_workgroup" > gpurun_out/r04_call29_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04_call29_pytest.log
export AB_VARIANTS='[["super-blocks",{}],["skinny product + fused update",{"MGP_CG_DENSE1_COLS":"1"}]]'
timeout -k 10 900 python tools/ab_dense_cols.py 4096x6 4096x7 4096x8 3000x8 > gpurun_out/r04_ab_dense_cols_78.txt 2>&1; sed 's/dense CG //; s/ per iteration (300 steps)//; s/; checksum.*//' gpurun_out/r04_ab_dense_cols_78.txt
