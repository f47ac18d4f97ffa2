#!/bin/bash
# PMC passes over the fused sweep at a BASELINE config (default C3); one --pmc set per run, as
# MI355X_MICROARCH.md prescribes (SQ: 8 slots per pass; FETCH_SIZE and WRITE_SIZE in passes of their own).
# Usage: tools/pmc_sweep.sh <outdir under gpurun_out> [config] [right-hand sides]
set -u
out=gpurun_out/${1:-pmc}
cfg=${2:-C3}
rhs=${3:-1}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$out/$name" -- python3 "$R/tools/run_sweep.py" --config "$cfg" --rhs "$rhs" --launches 3 > "$R/$out/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
run b SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run c FETCH_SIZE
run d WRITE_SIZE
