#!/bin/bash
# Does the fault need another PROCESS, or just the library's GEMM on the same CUs?  The reproducer with E3's forward pass (the
# library's implicit GEMM through the C-ABI) on a second stream of its OWN process.  bash scripts/probe/pk_opsel_repro6.sh [seconds]
secs=${1:-10}
(cd scripts/probe && for v in 0 1; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DVARIANT=$v -fno-slp-vectorize pk_opsel_repro.hip -o pk_opsel_repro$v -ldl 2>/dev/null || exit 1; done) || exit 1
for v in 0 1; do timeout -k 10 $((secs + 60)) scripts/probe/pk_opsel_repro$v $secs 5 | head -2; done
