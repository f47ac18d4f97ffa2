#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["default",{}]]'
timeout -k 10 300 python tools/ab_dense1.py 2 1024 2048 > gpurun_out/r04_ab_dense1_full_u.txt 2>&1; cut -c1-300 gpurun_out/r04_ab_dense1_full_u.txt | sed 's/, [0-9a-f]\{16\}//g; s/ (no poll \/ poll 25)//g'
MGP_ONLY=1 timeout -k 10 600 python tools/ab_dense_cols.py 2048x2 2048x5 2048x8 1024x5 > gpurun_out/r04_ab_dense_cols_full_u.txt 2>&1; grep "round 0" gpurun_out/r04_ab_dense_cols_full_u.txt | sed 's/dense CG //; s/ per iteration (300 steps)//; s/; checksum.*//'
timeout -k 10 600 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q -k "fixed_steps or several_columns or missing_workgroup or every_form" 2>&1 | tail -2
