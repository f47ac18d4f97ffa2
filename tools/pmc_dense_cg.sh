#!/bin/bash
# PMC passes over the register-resident dense CG (one launch = a 300-step solve at n = 4096, one and five columns):
# HBM bytes fetched / written per launch, in passes of their own as MI355X_MICROARCH.md prescribes.
# Usage: tools/pmc_dense_cg.sh <outdir under gpurun_out>
set -u
out=gpurun_out/${1:-pmc_dense_cg}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for bt in 1 5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/$out/bt${bt}_$c" -- python3 "$R/tools/run_probe_cg.py" 4096 $bt 300 > "$R/$out/bt${bt}_$c.log" 2>&1
    echo "pass bt=$bt $c rc=$? $(tail -1 $R/$out/bt${bt}_$c.log | cut -c1-100)"
  done
done
cd "$R"
for bt in 1 5; do echo "== $bt column(s)"; python3 tools/pmc_summary.py "$out" d1_persist 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k[-70:], {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items()})
"; break; done
