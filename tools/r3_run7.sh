set -e
mkdir -p gpurun_out
rm -f gpurun_out/r3_rank_ab.log
for t in 16 8 4 32; do MGP_SWEEP_TARGET=$t python tools/run_rank_share.py 131072 262144 >> gpurun_out/r3_rank_ab.log 2>&1; done
for ns in 2 4 16; do MGP_NOSPLIT_PER_CU=$ns python tools/run_rank_share.py 131072 262144 >> gpurun_out/r3_rank_ab.log 2>&1; done
grep -v amdgpu.ids gpurun_out/r3_rank_ab.log
