#!/bin/bash
# VF_PG_DBG ablations of k_pconv_patch_h beside k_pconv_patch_g (timing only; wrong results): 1 = no DMAs after the prologue, 4 = no MFMAs,
# 8 = no output stores, 32 = no first stage either
out=gpurun_out/${1:-r5h2}_patch_h_ablate.txt
: > $out
for dbg in 0 1 4 5 8 33 37 45; do
  echo "VF_PG_DBG=$dbg" >> $out
  VF_PG_DBG=$dbg ONLY=${ONLY:-E2,C1@2B} timeout -k 10 120 python scripts/probe/patch_h_check.py 64 2>/dev/null | grep " us " >> $out || exit 1
done
cat $out
