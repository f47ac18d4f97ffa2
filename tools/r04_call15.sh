#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["fp=0",{"MGP_D1_FIRST_POLL":"0"}],["fp=8",{"MGP_D1_FIRST_POLL":"8"}],["fp=12",{"MGP_D1_FIRST_POLL":"12"}],["fp=16",{"MGP_D1_FIRST_POLL":"16"}],["fp=24",{"MGP_D1_FIRST_POLL":"24"}]]'
timeout -k 10 900 python tools/ab_dense1.py 2 4096 3000 2048 > gpurun_out/r04_ab_dense1_fp.txt 2>&1
echo "ab rc=$?"; cut -c1-420 gpurun_out/r04_ab_dense1_fp.txt
