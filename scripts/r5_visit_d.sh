#!/bin/bash
# the GPU suite, then same-box A/B of the one-launch BatchNorm for small tensors on the three workloads
tag=${1:-r5m}
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -4 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
for wl in center vid16 wholeim; do
  AB_ARGS="--workload $wl" bash scripts/ab_env.sh gpurun_out/${tag}_ab_bn_small_$wl.txt 2 "VF_BN_SMALL=0" "VF_BN_SMALL=1" > /dev/null || exit 1
  echo "== $wl"; cat gpurun_out/${tag}_ab_bn_small_$wl.txt
done
