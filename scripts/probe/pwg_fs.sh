#!/bin/bash
# one fragment set (73 registers: three blocks per CU) against two (124: two blocks per CU) in k_pwgrad_group
for fs in 2 1 2 1; do echo "VF_PWG_FS=$fs"; VF_PWG_FS=$fs timeout -k 10 120 python scripts/bench_pwgrad.py 64 2>/dev/null | grep " dW "; done
bash scripts/ab_env.sh gpurun_out/r5u_ab_pwg_fs.txt 3 "VF_PWG_FS=2" "VF_PWG_FS=1"
