set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "cg or symm or empty" > gpurun_out/r3_cg_tests.log 2>&1 || { tail -30 gpurun_out/r3_cg_tests.log; exit 1; }
tail -3 gpurun_out/r3_cg_tests.log
rm -f gpurun_out/r3_dense_cg.log
for f in 0 1 2; do for c in 0 1; do MGP_TRI_FORM=$f MGP_CG_TRI_FUSED=$c python tools/run_dense_cg.py >> gpurun_out/r3_dense_cg.log 2>&1; done; done
grep -v amdgpu.ids gpurun_out/r3_dense_cg.log
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  MGP_TRI_FORM=$f MGP_CG_TRI_FUSED=$f rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dense$f -o dense$f -- python3 $GRAFT_REPO_ROOT/tools/run_dense_cg.py 4096 > $GRAFT_REPO_ROOT/gpurun_out/prof_dense$f.log 2>&1
done
cd $GRAFT_REPO_ROOT
for f in 0 1; do echo "== form/fused $f"; find gpurun_out/prof_dense$f -name "*kernel_stats.csv" | head -1 | xargs head -12; done
