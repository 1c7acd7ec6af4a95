#!/bin/bash
# ONE round (three concurrent processes) of scripts/probe/multi_trainer_det.py per library variant.
# bash scripts/probe/mt_variants.sh <iters> <variant...>      ("cur" = the in-tree build)
iters=$1; shift
out=gpurun_out/mt_variants; mkdir -p $out
for v in "$@"; do
  if [ $v = cur ]; then unset VF_HIP_LIB; else export VF_HIP_LIB=$PWD/video-filler_amd/lib/alt/libvf_hip_$v.so; fi
  pids=""
  for p in a b c; do VF_PROBE_ITERS=$iters timeout -k 10 400 python scripts/probe/multi_trainer_det.py > $out/${v}_$p.log 2>&1 & pids="$pids $!"; done
  rc=0; for q in $pids; do wait $q || rc=$?; done
  echo "== $v (rc $rc): $(grep -c MISMATCH $out/${v}_?.log | tr '\n' ' ')"
  if [ $rc -ge 124 ]; then echo "killed: no further GPU step"; exit $rc; fi
done
