#!/bin/bash
# polls with all loads in flight (poll_units): tests, then A/B of the register-resident forms
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q > gpurun_out/r04_call21_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04_call21_pytest.log
export AB_VARIANTS='[["default (full matrix <= 2048, super-blocks above)",{}],["super-blocks everywhere",{"MGP_CG_DENSE1":"4"}],["two-launch",{"MGP_CG_DENSE1":"1"}]]'
timeout -k 10 300 python tools/ab_dense1.py 1 1024 2048 3000 4096 > gpurun_out/r04_ab_dense1_polls.txt 2>&1; cut -c1-700 gpurun_out/r04_ab_dense1_polls.txt | sed 's/, [0-9a-f]\{16\}//g; s/ (no poll \/ poll 25)//g'
export AB_VARIANTS='[["default",{}],["super-blocks everywhere",{"MGP_CG_DENSE1":"4"}]]'
timeout -k 10 600 python tools/ab_dense_cols.py 4096x2 4096x3 4096x4 4096x5 4096x6 3000x5 2048x2 2048x5 2048x6 2048x8 1024x5 > gpurun_out/r04_ab_dense_cols_polls.txt 2>&1; sed 's/dense CG //; s/ per iteration (300 steps)//; s/; checksum.*//' gpurun_out/r04_ab_dense_cols_polls.txt
