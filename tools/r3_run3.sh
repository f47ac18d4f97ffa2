set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "cg or symm or empty" > gpurun_out/r3_cg_tests.log 2>&1 || { tail -40 gpurun_out/r3_cg_tests.log; exit 1; }
tail -3 gpurun_out/r3_cg_tests.log
rm -f gpurun_out/r3_dense_cg.log
for c in 0 1; do MGP_TRI_FORM=1 MGP_CG_DENSE1=$c python tools/run_dense_cg.py >> gpurun_out/r3_dense_cg.log 2>&1; done
grep -v amdgpu.ids gpurun_out/r3_dense_cg.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_d1 -o d1 -- python3 $GRAFT_REPO_ROOT/tools/run_dense_cg.py 4096 > $GRAFT_REPO_ROOT/gpurun_out/prof_d1.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_d1 -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200
