#!/bin/bash
# PMC passes over the skinny symmetric product (tools/run_skinny.py).  Usage: tools/pmc_skinny.sh <outdir> [n] [Bt]
set -u
out=gpurun_out/${1:-pmc_skinny}; n=${2:-4096}; bt=${3:-64}
mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$out/$name" -- python3 "$R/tools/run_skinny.py" $n $bt > "$R/$out/$name.log" 2>&1
  echo "pass $name rc=$? $(tail -1 $R/$out/$name.log)"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run b SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
run c TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run d FETCH_SIZE
run e SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
