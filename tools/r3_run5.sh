set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense1.py -x -q -s > gpurun_out/r3_dense1_tests.log 2>&1 || { tail -40 gpurun_out/r3_dense1_tests.log; exit 1; }
grep "steps HIP" gpurun_out/r3_dense1_tests.log; tail -3 gpurun_out/r3_dense1_tests.log
MGP_CG_DENSE1=0 python -m pytest tests/test_gpu_dense1.py -q -s -k "converged" > gpurun_out/r3_dense1_tests_old.log 2>&1 || true
grep "steps HIP" gpurun_out/r3_dense1_tests_old.log; tail -3 gpurun_out/r3_dense1_tests_old.log
python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_all.log 2>&1 || { tail -40 gpurun_out/r3_gpu_all.log; exit 1; }
tail -3 gpurun_out/r3_gpu_all.log
