#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dense1.py -m gpu -x -q -k "missing_workgroup or fixed_steps" > gpurun_out/r04_call26_pytest.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r04_call26_pytest.log
