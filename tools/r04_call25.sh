#!/bin/bash
# full GPU suite
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04_call25_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r04_call25_pytest.log
[ $rc -eq 0 ] || exit 1
