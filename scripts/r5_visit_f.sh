#!/bin/bash
# fused bottleneck update on the bf16 pipe (VF_ADAM_PLANES=1, default) against the fp32-pipe form (0): the kernel alone, its tests, the iteration
tag=${1:-r5y}
for pl in 0 1 0 1; do echo "VF_ADAM_PLANES=$pl"; VF_ADAM_PLANES=$pl timeout -k 10 120 python scripts/bench_fused_adam.py 2>/dev/null; done > gpurun_out/${tag}_fused_adam.txt
cat gpurun_out/${tag}_fused_adam.txt
timeout -k 10 600 python -m pytest tests/test_gpu_fused_adam.py tests/test_gpu_dp_rehearsal.py -m gpu -q -x > gpurun_out/${tag}_adam_tests.log 2>&1; rc=$?; tail -4 gpurun_out/${tag}_adam_tests.log
[ $rc -ne 0 ] && exit $rc
bash scripts/ab_env.sh gpurun_out/${tag}_ab_adam_planes.txt 4 "VF_ADAM_PLANES=0" "VF_ADAM_PLANES=1"
