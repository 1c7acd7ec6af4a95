#!/bin/bash
# Bisecting the trigger: the reproducer beside synthetic neighbour kernels of its own process that each do ONE of the things the
# library's implicit GEMM does.  bash scripts/probe/pk_opsel_repro7.sh [seconds]
secs=${1:-8}
(cd scripts/probe && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DVARIANT=0 -fno-slp-vectorize pk_opsel_repro.hip -o pk_opsel_repro0 -ldl 2>/dev/null) || exit 1
for nb in 6 7 8 9 1 5; do timeout -k 10 $((secs + 60)) scripts/probe/pk_opsel_repro0 $secs $nb | head -1 | sed 's/(0 none.*VALU) //'; done
