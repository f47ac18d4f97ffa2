#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["triangle (MGP_CG_DENSE1=4)",{"MGP_CG_DENSE1":"4"}],["full fp=0",{"MGP_D1_FIRST_POLL":"0"}],["full fp=8",{"MGP_D1_FIRST_POLL":"8"}],["full fp=16",{"MGP_D1_FIRST_POLL":"16"}],["full fp=24",{"MGP_D1_FIRST_POLL":"24"}],["full fp=40",{"MGP_D1_FIRST_POLL":"40"}],["two-launch",{"MGP_CG_DENSE1":"1"}]]'
timeout -k 10 900 python tools/ab_dense1.py 1 1024 1280 1536 2048 > gpurun_out/r04_ab_dense1_full.txt 2>&1
echo "ab rc=$?"; cut -c1-520 gpurun_out/r04_ab_dense1_full.txt
