set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_predict_configs.py -x -q -s > gpurun_out/r3_predict_tests.log 2>&1 || { tail -30 gpurun_out/r3_predict_tests.log; exit 1; }
tail -15 gpurun_out/r3_predict_tests.log
python -m pytest tests/test_gpu_parity.py -x -q -k "cg or symm or empty" > gpurun_out/r3_cg_tests.log 2>&1 || { tail -30 gpurun_out/r3_cg_tests.log; exit 1; }
tail -3 gpurun_out/r3_cg_tests.log
for f in 0 1 2; do for c in 0 1; do MGP_TRI_FORM=$f MGP_CG_TRI_FUSED=$c python tools/run_dense_cg.py >> gpurun_out/r3_dense_cg.log 2>&1; done; done
cat gpurun_out/r3_dense_cg.log
