#!/bin/bash
# bash scripts/probe/mt_loop.sh <rounds> [ENV=VAL ...]: three concurrent copies of multi_trainer_det.py per round; counts process-runs with a mismatch
rounds=$1; shift
bad=0; tot=0
for i in $(seq 1 $rounds); do
  for p in a b c; do env "$@" python scripts/probe/multi_trainer_det.py > gpurun_out/mt_$p.log 2>&1 & done
  wait
  for p in a b c; do tot=$((tot+1)); grep -q "MISMATCH" gpurun_out/mt_$p.log && bad=$((bad+1)); grep -q "done" gpurun_out/mt_$p.log || { echo "run died:"; tail -3 gpurun_out/mt_$p.log; }; done
done
echo "[$*] process-runs with a mismatch: $bad of $tot"
