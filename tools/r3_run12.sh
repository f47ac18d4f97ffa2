set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_all.log 2>&1 || { tail -40 gpurun_out/r3_gpu_all.log; exit 1; }
tail -3 gpurun_out/r3_gpu_all.log
R=$GRAFT_REPO_ROOT
python tools/microbench.py contract > gpurun_out/r03_contract_microbench.txt 2>&1; grep -v amdgpu gpurun_out/r03_contract_microbench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_contract_c5 -o c5 -- python3 $R/tools/run_contract.py C5 > $R/gpurun_out/prof_contract_c5.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_contract_c5 -o c5 -- python3 $R/tools/run_contract.py C5 > $R/gpurun_out/pmc_contract_c5.log 2>&1
cd $R
python3 tools/show_stats.py gpurun_out/prof_contract_c5/c5_kernel_stats.csv | head -8 | cut -c1-160
python3 tools/pmc_summary.py gpurun_out/pmc_contract_c5_parent gemm 2>/dev/null | head -5 || true
ls gpurun_out/pmc_contract_c5
