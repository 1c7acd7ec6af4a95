#!/bin/bash
# ONE bounded run per variant of the row-dot kernel (scripts/probe/rowdot_variants.hip): three concurrent copies of the in-place
# probe share the GPU, as in the round-3 failure.  Usage on the GPU box: bash scripts/probe/rowdot_variants.sh [N]
N=${1:-20000}
out=gpurun_out/rowdot_variants; mkdir -p $out
for v in prefix prefixdiag uncond lds depth1 dpp shipped; do
  if [ $v = shipped ]; then unset VF_HIP_LIB; else export VF_HIP_LIB=$PWD/video-filler_amd/lib/alt/libvf_hip_rd$v.so; fi
  pids=""
  for p in a b c; do N=$N timeout -k 10 300 python scripts/probe/smallm_det.py > $out/${v}_$p.log 2>&1 & pids="$pids $!"; done
  rc=0; for q in $pids; do wait $q || rc=$?; done
  echo "== $v (rc $rc): $(grep -h 'mismatching runs' $out/${v}_?.log | tr '\n' ' ')"
  if [ $rc -ge 124 ]; then echo "killed: no further GPU step"; exit $rc; fi
done
