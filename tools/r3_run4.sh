set -e
mkdir -p gpurun_out
rm -f gpurun_out/r3_dense_cg.log
for rep in 1 2; do
python tools/run_dense_cg.py 4096 2048 >> gpurun_out/r3_dense_cg.log 2>&1
MGP_LIBRARY=$PWD/conjugate-gradient-sparse-gp_amd/cggp/libmgp_ab.so python tools/run_dense_cg.py 4096 2048 2>&1 | sed 's/^/[AB lib] /' >> gpurun_out/r3_dense_cg.log
done
grep -v amdgpu.ids gpurun_out/r3_dense_cg.log
