#!/bin/bash
# Rehearse bench.py's N > 1 launch with W gloo ranks on the ONE GPU of a gpurun box (RCCL refuses two ranks on
# one device, so the exchange goes through the callback hook; rows, Kmm slabs, agreement word and the enqueue
# loop are the real ones).  usage: tools/rehearse_world.sh W [rows]   ->  gpurun_out/rehearse_wW.{json,err}
# The box allows at most 6 processes on the card at once: W <= 5 leaves room for the launcher's children.
set -u
W=${1:?world size}
ROWS=${2:-98304}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=$((29600 + W))
timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$W" --master-addr 127.0.0.1 \
  --master-port "$PORT" bench.py --gpus "$W" --backend gloo --steps 5 --warmup 2 --rows "$ROWS" \
  --convergence-cap 16 --no-cpu-baseline --bootstrap-timeout 120 --run-timeout 300 \
  > "gpurun_out/rehearse_w$W.json" 2> "gpurun_out/rehearse_w$W.err"
rc=$?
echo "rehearse W=$W rc=$rc json_bytes=$(stat -c %s gpurun_out/rehearse_w$W.json)" | tee -a gpurun_out/rehearse_summary.txt
exit $rc
