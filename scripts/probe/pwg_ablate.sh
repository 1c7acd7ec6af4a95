#!/bin/bash
# VF_PWG_DBG ablations of k_pwgrad_group (1 = no DMAs after a block's first stage, 4 = no MFMAs, 8 = no stores) — the switch lived in a
# timing-only build of vf_pgemm.hip (a __device__ flag read at kernel entry; not in the tree); results: profiles/r05_i_pwgrad_three_blocks_per_cu.txt
for dbg in 0 1 4 5 8 13; do echo "VF_PWG_DBG=$dbg"; VF_PWG_DBG=$dbg timeout -k 10 120 python scripts/bench_pwgrad.py 64 2>/dev/null | grep " dW " ; done
