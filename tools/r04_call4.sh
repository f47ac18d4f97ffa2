#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["two-launch",{"MGP_CG_DENSE1":"1"}],["fused hs=0",{"MGP_CG_DENSE1":"2","MGP_D1_HEADSTART":"0"}],["fused hs=8",{"MGP_CG_DENSE1":"2","MGP_D1_HEADSTART":"8"}],["fused hs=16",{"MGP_CG_DENSE1":"2","MGP_D1_HEADSTART":"16"}],["fused hs=32",{"MGP_CG_DENSE1":"2","MGP_D1_HEADSTART":"32"}],["fused hs=64",{"MGP_CG_DENSE1":"2","MGP_D1_HEADSTART":"64"}],["fused hs=128",{"MGP_CG_DENSE1":"2","MGP_D1_HEADSTART":"128"}]]'
timeout -k 10 900 python tools/ab_dense1.py 1 2048 4096 > gpurun_out/r04_ab_dense1_hs.txt 2>&1
echo "ab rc=$?"; cat gpurun_out/r04_ab_dense1_hs.txt
