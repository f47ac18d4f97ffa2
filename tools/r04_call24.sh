#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["one round in flight",{}],["two, stagger 8",{"MGP_D1_POLL2":"8"}],["two, stagger 16",{"MGP_D1_POLL2":"16"}],["two, stagger 24",{"MGP_D1_POLL2":"24"}],["two, stagger 40",{"MGP_D1_POLL2":"40"}]]'
timeout -k 10 400 python tools/ab_dense1.py 2 2048 4096 > gpurun_out/r04_ab_dense1_poll2.txt 2>&1; cut -c1-400 gpurun_out/r04_ab_dense1_poll2.txt | sed 's/, [0-9a-f]\{16\}//g; s/ (no poll \/ poll 25)//g'
timeout -k 10 600 python tools/ab_dense_cols.py 4096x5 2048x5 > gpurun_out/r04_ab_dense_cols_poll2.txt 2>&1; sed 's/dense CG //; s/ per iteration (300 steps)//; s/; checksum.*//' gpurun_out/r04_ab_dense_cols_poll2.txt
MGP_D1_POLL2=16 timeout -k 10 600 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q -k "super_block or fixed_steps or several_columns" 2>&1 | tail -3
