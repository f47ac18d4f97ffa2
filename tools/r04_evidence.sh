#!/bin/bash
# round 4 evidence: default bench bare + under rocprofv3 --kernel-trace --stats, PMC passes of the sweep at C3 / C4 / C5
set -u
R=$GRAFT_REPO_ROOT
mkdir -p "$R/gpurun_out"
bash "$R/tools/final_evidence.sh" r04_final
cd "$R"
for c in C3 C4 C5; do
  lc=$(echo $c | tr A-Z a-z)
  bash tools/pmc_sweep.sh r04_pmc_$lc $c > gpurun_out/r04_pmc_$lc.log 2>&1
  echo "pmc $c rc=$?"
done
python3 tools/make_valu_model.py r04 C3=gpurun_out/r04_pmc_c3 C4=gpurun_out/r04_pmc_c4 C5=gpurun_out/r04_pmc_c5 > gpurun_out/r04_model.log 2>&1
echo "model rc=$?"; tail -20 gpurun_out/r04_model.log
