#!/bin/bash
mkdir -p gpurun_out
export AB_VARIANTS='[["two-launch",{"MGP_CG_DENSE1":"1"}],["fused base",{"MGP_CG_DENSE1":"2"}],["fused sleep=32",{"MGP_CG_DENSE1":"2","MGP_D1_POLL_SLEEP":"32"}],["fused sleep=128",{"MGP_CG_DENSE1":"2","MGP_D1_POLL_SLEEP":"128"}],["fused delay=64 sleep=16",{"MGP_CG_DENSE1":"2","MGP_D1_POLL_DELAY":"64","MGP_D1_POLL_SLEEP":"16"}],["fused delay=128 sleep=32",{"MGP_CG_DENSE1":"2","MGP_D1_POLL_DELAY":"128","MGP_D1_POLL_SLEEP":"32"}],["fused occ=1",{"MGP_CG_DENSE1":"2","MGP_D1_OCC":"1"}],["fused occ=1 delay=64 sleep=32",{"MGP_CG_DENSE1":"2","MGP_D1_OCC":"1","MGP_D1_POLL_DELAY":"64","MGP_D1_POLL_SLEEP":"32"}]]'
timeout -k 10 900 python tools/ab_dense1.py 1 2048 4096 > gpurun_out/r04_ab_dense1_poll.txt 2>&1
echo "ab rc=$?"; cut -c1-330 gpurun_out/r04_ab_dense1_poll.txt
