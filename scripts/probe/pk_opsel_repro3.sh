#!/bin/bash
# Which unit's traffic on the CU does the packed-FMA fault need?  The reproducer beside a neighbour kernel of its OWN process on a
# second stream: matrix cores / LDS / global memory / plain VALU.  bash scripts/probe/pk_opsel_repro3.sh [seconds]
secs=${1:-10}
out=gpurun_out/pk_repro; mkdir -p $out
(cd scripts/probe && for v in 0 1; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DVARIANT=$v -fno-slp-vectorize pk_opsel_repro.hip -o pk_opsel_repro$v -ldl 2>/dev/null || exit 1; done) || exit 1
for nb in 0 1 2 3 4; do timeout -k 10 $((secs + 60)) scripts/probe/pk_opsel_repro0 $secs $nb > $out/nb$nb.log 2>&1; head -1 $out/nb$nb.log; tail -1 $out/nb$nb.log | grep "of the first"; done
echo "== control (variant 1) beside the matrix-core neighbour"
timeout -k 10 $((secs + 60)) scripts/probe/pk_opsel_repro1 $secs 1 | head -1
