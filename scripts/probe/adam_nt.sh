#!/bin/bash
# non-temporal stores (1) / loads + stores (3) of x, m, v in the fused bottleneck update — VF_ADAM_NT lived in a timing-only build of
# vf_wgrad_small.hip (not in the tree); results: profiles/r05_i_fused_update_bf16_pipe.txt
for nt in 0 1 3 0 1 3; do echo "VF_ADAM_NT=$nt"; VF_ADAM_NT=$nt timeout -k 10 120 python scripts/bench_fused_adam.py 2>/dev/null | head -3; done
bash scripts/ab_env.sh gpurun_out/r5x_ab_adam_nt.txt 4 "VF_ADAM_NT=0" "VF_ADAM_NT=1" "VF_ADAM_NT=3"
