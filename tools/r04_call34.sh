#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py tests/test_gpu_configs.py tests/test_gpu_bench_contract.py -m gpu -x -q > gpurun_out/r04_call34_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_call34_pytest.log
