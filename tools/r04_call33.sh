#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["now",{}],["mid-round build (ea9d79a)",{"MGP_LIBRARY":"'$GRAFT_REPO_ROOT'/tools/dbg/libmgp_r04mid.so"}]]'
timeout -k 10 400 python tools/ab_dense1.py 2 2048 3000 4096 > gpurun_out/r04_ab_dense1_vs_mid.txt 2>&1; cut -c1-500 gpurun_out/r04_ab_dense1_vs_mid.txt | sed 's/, [0-9a-f]\{16\}//g; s/ (no poll \/ poll 25)//g'
for bt in 1; do D1_TRACE_BT=$bt timeout -k 10 200 python tools/d1_trace.py gpurun_out/r04_d1_trace_now_bt$bt.txt 4096 || exit 1; done
grep -E "it (1[0-9]):" gpurun_out/r04_d1_trace_now_bt1.txt | tail -20
