#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r04_d1_trace_*.txt
for bt in 1 5; do D1_TRACE_BT=$bt timeout -k 10 200 python tools/d1_trace.py gpurun_out/r04_d1_trace_blk_bt$bt.txt 4096 || exit 1; done
for bt in 1 5; do D1_TRACE_BT=$bt timeout -k 10 200 python tools/d1_trace.py gpurun_out/r04_d1_trace_full_bt$bt.txt 2048 || exit 1; done
for f in gpurun_out/r04_d1_trace_*.txt; do echo "== $f"; grep -E "it (1[0-9]|2[0-3]):" $f | tail -28; done
