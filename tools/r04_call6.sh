#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["two-launch",{"MGP_CG_DENSE1":"1"}],["register-resident",{"MGP_CG_DENSE1":"3"}]]'
timeout -k 10 300 python tools/ab_dense1.py 1 2048 4096 4001 1024 > gpurun_out/r04_ab_dense1_persist.txt 2>&1
echo "ab rc=$?"; cut -c1-420 gpurun_out/r04_ab_dense1_persist.txt
timeout -k 10 600 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q > gpurun_out/r04_call6_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04_call6_pytest.log
